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

// ---- the 8-wide compressed tree ----
// 80-byte node, five 16-byte loads.  The boxes of up to eight children are stored as 8-bit offsets from the node's
// own corner `lo` in units of 2^(e - 127) per axis (e = the biased exponent byte of that axis), rounded outwards, so
// a child's stored box always contains its true box: boxes only cull, a wider one costs a visit, never a hit.
// Children sit in slots chosen at build time so that a ray can visit them near to far without sorting: for a ray
// whose direction signs are (sx, sy, sz), oct = (sx < 0) | (sy < 0) << 1 | (sz < 0) << 2, the slots are visited in
// falling order of (slot ^ oct); slot 7 holds the child nearest the (-,-,-) corner, slot 0 the one nearest (+,+,+).
// meta[slot]: 0 = empty.  Inner node: 0x20 | (24 + slot); its node index is childBase + the number of inner slots
// below it.  Leaf of n triangles (1..3): the low n of the top three bits set (0x20, 0x60, 0xe0), low five bits = the
// offset of its first triangle from triBase (0..23; a node's leaf triangles are consecutive in the reordered
// triangle array).  imask has bit `slot` set for inner nodes.  (Layout after Ylitie, Karras, Laine: "Efficient
// Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017.)
struct Bvh8Node {
    float lo[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t childBase;
    uint32_t triBase;
    uint8_t meta[8];
    uint8_t qlo[3][8]; // [axis][slot]
    uint8_t qhi[3][8];
};
static_assert(sizeof(Bvh8Node) == 80, "node must be 80 bytes");

struct Bvh8Result {
    std::vector<Bvh8Node> nodes;     // node 0 is the root
    std::vector<uint32_t> triOrder;  // reordered position -> input triangle index
    uint32_t maxDepth = 0;           // levels of inner nodes; a traversal stacks at most one entry per level
    float rootLo[3], rootHi[3];
};
void build_bvh8(const float* vertices, const uint32_t* indices, uint32_t nTris, Bvh8Result* out);

// vertices: [nV*3], indices: [nT*3].  A scene with a single leaf gets a root with one used
// slot, an empty scene a root with none, so traversal code needs no special case.  maxLeaf <= 8.
void build_bvh(const float* vertices, const uint32_t* indices, uint32_t nTris, uint32_t maxLeaf, BvhResult* out);

} // namespace enigma
