// Host builder of the device BVH.
//
// Replaces BVHAccel construction (Enigma/BVH.cpp:148-635).  The reference builds an HLBVH
// with up to 127 primitives per leaf over pointer-chasing primitives and flattens it to
// 32-byte nodes; results of ray queries do not depend on the tree (only exact-distance ties
// do), so the device tree is built for the GPU instead: binned-SAH binary tree, small
// leaves, and a node that carries BOTH children's boxes so one 64-byte fetch decides the
// descent.
#pragma once
#include <cstdint>
#include <vector>

namespace enigma {

// 64-byte node, four float4 loads:
//   q0 = {lo0.x, lo0.y, lo0.z, hi0.x}  q1 = {hi0.y, hi0.z, lo1.x, lo1.y}
//   q2 = {lo1.z, hi1.x, hi1.y, hi1.z}  q3 = {child0, child1, 0, 0} (bit patterns)
// child >= 0: inner node index.  child < 0: leaf, ~child = (first << 3) | (count - 1),
// count in 1..8, first = index into the reordered triangle array.
struct BvhNode {
    float lo0[3], hi0x;
    float hi0yz[2], lo1xy[2];
    float lo1z, hi1[3];
    int32_t child0, child1;
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "node must be 64 bytes");

struct BvhResult {
    std::vector<BvhNode> nodes;      // node 0 is the root (always an inner node; see below)
    std::vector<uint32_t> triOrder;  // reordered position -> input triangle index
    uint32_t maxDepth = 0;           // deepest stack a traversal can need
    float rootLo[3], rootHi[3];
};

// vertices: [nV*3], indices: [nT*3].  A scene with a single leaf gets a root whose second
// child is an empty box, so traversal code needs no special case.  maxLeaf <= 8.
void build_bvh(const float* vertices, const uint32_t* indices, uint32_t nTris, uint32_t maxLeaf, BvhResult* out);

} // namespace enigma
