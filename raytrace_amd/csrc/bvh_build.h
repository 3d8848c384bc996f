// Host builder of the device BVH.
//
// Replaces BVHAccel construction (Enigma/BVH.cpp:148-635).  The reference builds an HLBVH
// with up to 127 primitives per leaf over pointer-chasing primitives and flattens it to
// 32-byte nodes; results of ray queries do not depend on the tree (only exact-distance ties
// do), so the device tree is built for the GPU instead: a binned-SAH binary tree down to
// single primitives, collapsed to an 8-wide tree with quantised child boxes whose node a ray
// tests with one 80-byte fetch.
#pragma once
#include <cstdint>
#include <vector>

namespace enigma {

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
    std::vector<uint32_t> triOrder;  // reordered position -> input primitive index (triangles first, then the boxes)
    uint32_t maxDepth = 0;           // levels of inner nodes; a traversal stacks at most one entry per level
    float rootLo[3], rootHi[3];
};
// vertices: [nV*3], indices: [nTriangles*3]; boxes: [nBoxes*6] (lo.xyz, hi.xyz) of primitives that are not triangles
// (spheres), which become leaf entries like triangles: input index nTriangles + i in triOrder.  An empty scene gets a
// root without children, a single leaf a root with one used slot, so traversal code needs no special case.  The tree
// is at most 64 levels deep.
void build_bvh8(const float* vertices, const uint32_t* indices, uint32_t nTriangles, const float* boxes, uint32_t nBoxes, Bvh8Result* out);

} // namespace enigma
