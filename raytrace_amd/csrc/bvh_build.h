// Host builder of the device BVH.
//
// Replaces BVHAccel construction (Enigma/BVH.cpp:148-635).  The reference builds an HLBVH
// with up to 127 primitives per leaf over pointer-chasing primitives and flattens it to
// 32-byte nodes; results of ray queries do not depend on the tree (only exact-distance ties
// do), so the device tree is built for the GPU instead: a binned-SAH binary tree with small
// leaves, collapsed to a 4-wide tree whose node carries the boxes of all four children.  A
// ray then decides the descent among four subtrees from one 128-byte fetch (one cache line
// per lane), which halves the chain of dependent fetches of a binary tree.
#pragma once
#include <cstdint>
#include <vector>

namespace enigma {

// 128-byte node, eight float4 loads (seven used); child boxes in structure-of-arrays form.
// child >= 0: inner node index.  child < 0: leaf, ~child = (first << 3) | (count - 1),
// count in 1..8, first = index into the reordered triangle array.  kEmptyChild: unused slot
// (its box is inverted, lo = +FLT_MAX, hi = -FLT_MAX).
constexpr int32_t kEmptyChild = INT32_MIN;
struct Bvh4Node {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "node must be 128 bytes");

struct BvhResult {
    std::vector<Bvh4Node> nodes;     // node 0 is the root (always an inner node)
    std::vector<uint32_t> triOrder;  // reordered position -> input triangle index
    uint32_t maxDepth = 0;           // inner-node levels of the 4-wide tree
    uint32_t stackNeed = 0;          // deepest stack a traversal can need: 3 entries per level
    float rootLo[3], rootHi[3];
};

// vertices: [nV*3], indices: [nT*3].  A scene with a single leaf gets a root with one used
// slot, an empty scene a root with none, so traversal code needs no special case.  maxLeaf <= 8.
void build_bvh(const float* vertices, const uint32_t* indices, uint32_t nTris, uint32_t maxLeaf, BvhResult* out);

} // namespace enigma
