// Traversal of the 8-wide compressed tree (bvh_build.h, Bvh8Node).
//
// Replaces BVHAccel::intersect / intersectP (Enigma/BVH.cpp:639-722) like dev_trace.h does, with the same triangle
// test (Triangle::intersects, operation for operation), so hits, t and barycentrics stay bit-identical; only the
// culling differs, and culling here is conservative by construction:
//  - child boxes are stored rounded outwards (8-bit offsets from the node's corner);
//  - the slab distances are one fused multiply-add per plane, t = q * (2^e / d) + (corner - o) / d, evaluated with a
//    slack that bounds its rounding error: entry distances are lowered and exit distances raised by
//    2^-21 * (|(corner - o) / d| + 256 * |2^e / d|), more than twice the worst case of the roundings involved
//    (1 / d from the hardware reciprocal, one ulp; the difference, the product and the fused multiply-add, half an ulp each);
//  - 1 / d is clamped to +-2^80, which keeps every product finite (a ray parallel to a slab gets entry / exit
//    distances of the right sign and astronomic size, i.e. the slab does not constrain it while the origin is inside).
// A box that the exact ray meets within [0, tMax] is therefore never culled; the reference's own slab test gives the
// same guarantee through its 1 + 2 gamma(3) factor (Enigma/BBox.h:267-295).
//
// Children are visited near to far without sorting: slots are laid out by octant at build time, the hit children of
// a node form a bit mask whose bit order is the visiting order for this ray (slot ^ octant), and what is left of a
// node's children waits on the stack as ONE entry (first child index + mask).  The stack is as deep as the tree, not
// three entries per level.  Triangles of the hit leaves of a node form a 24-bit mask over the node's triangle range.
//
// Spheres are leaf entries like triangles (template flag SPH, the build for scenes that have any): a sphere owns a
// slot of the triangle array whose first word pair says so (sign bit of v0.w set, sphere index below it), and the
// triangle step of such a slot runs Sphere::intersect / intersectP (dev_sphere.h) instead.  A sphere hit is reported
// as primitive sphereBase + sphere index, sphereBase = the number of slots of the triangle array, with the object-
// space hit point in place of the barycentrics.
#pragma once
#include "dev_trace.h"
#include "dev_sphere.h"

namespace edev {

constexpr int kNode8Vec4 = 5;      // 16-byte pieces per node in global memory
// Triangle steps dealt out over the wave (dealTriangles), for scenes in global memory: on the 1.18 M-triangle scene the
// streaming shadow kernel gains 4 % by it (2.44 -> 2.34 ms per launch) and the closest-hit kernel 6 % (4.95 -> 4.67).  The
// closest-hit walker pays for it in registers (the gathered ray, the pairs, the returned hit records): in a build whose
// register allocation was already tight (device code at -O3: 59 registers spilled instead of 16) it LOST 15 %.  Both paths
// give the same bits (ENIGMA_SHARED_TRI: bit 0 = any-hit walker, bit 1 = closest-hit walker, for experiments).
#ifndef ENIGMA_SHARED_TRI
#define ENIGMA_SHARED_TRI 3
#endif
constexpr bool kSharedTriangleStepsAny = (ENIGMA_SHARED_TRI & 1) != 0, kSharedTriangleStepsClosest = (ENIGMA_SHARED_TRI & 2) != 0;
// Wave priorities (ENIGMA_WAVE_PRIO: the raised level, 0 = off).  A traversal wave alternates between long stretches of
// arithmetic (a node test, a triangle test, a ray's set-up) and the housekeeping in between that ends in the next fetch
// (stack, refill, vote, address).  Housekeeping runs at the raised priority, arithmetic at priority 0: a wave that is on
// its way to a fetch overtakes the waves of its SIMD that are computing, its loads are out sooner, and their latency is
// spent under the others' arithmetic.  Measured on the 1.18 M-triangle scene: closest-hit kernel 5.35 -> 5.06 ms per
// launch, streaming shadow kernel 2.45 -> 2.34 (5.83 -> 5.52 and 2.52 -> 2.40 in the -O3 build; raising the priority
// only from the start of a step to its loads gave a fifth of that).  The same switch around the fetches of the shade
// kernels and around a ray's set-up changed nothing.
#ifndef ENIGMA_WAVE_PRIO
#define ENIGMA_WAVE_PRIO 2
#endif
// after: the value whose computation must be complete first (keeps the compiler from moving the switch)
DEV void arithmeticBegins() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ENIGMA_WAVE_PRIO != 0) __builtin_amdgcn_s_setprio(0);
#endif
}
DEV void arithmeticEnds() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ENIGMA_WAVE_PRIO != 0) __builtin_amdgcn_s_setprio(ENIGMA_WAVE_PRIO);
#endif
}
template <class T> DEV void pinHere(const T* p) { asm volatile("" :: "v"(p)); }
DEV void pinHere(uint32_t a, uint32_t b) { asm volatile("" :: "v"(a), "v"(b)); }
DEV void pinHere(float a) { asm volatile("" :: "v"(a)); }
constexpr uint32_t kNodeHits = 0xff000000u;
// (A lower bound of the children's entry distances kept with a stacked group, to drop the group unfetched once the ray
// has a nearer hit, was measured and bought nothing: the bound includes the child taken first, whose subtree is what
// found the hit.)

struct Ray8 { float ox, oy, oz, ix, iy, iz; uint32_t oct4; bool nx, ny, nz; };

DEV float clampInv(float v) { // keeps the sign of +-inf and of +-0's reciprocal
    const float k = 0x1p80f;
    return v > k ? k : (v < -k ? -k : v);
}
DEV Ray8 makeRay8(const RaySetup& r) {
    Ray8 q;
    q.ox = r.o.x; q.oy = r.o.y; q.oz = r.o.z;
    q.ix = clampInv(r.invDir.x); q.iy = clampInv(r.invDir.y); q.iz = clampInv(r.invDir.z);
    q.nx = r.negX; q.ny = r.negY; q.nz = r.negZ;
    q.oct4 = ((r.negX ? 1u : 0u) | (r.negY ? 2u : 0u) | (r.negZ ? 4u : 0u)) * 0x01010101u;
    return q;
}

// a + b for counters that are the same in every lane of the wave: says so to the compiler, which then keeps them in
// scalar registers
DEV uint32_t uniformAdd(uint32_t a, uint32_t b) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(a + b)); }
DEV float ubyteToFloat(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); } // v_cvt_f32_ubyte<b>
typedef float Pair __attribute__((ext_vector_type(2)));
#ifndef ENIGMA_PACKED_NODE_TEST
#define ENIGMA_PACKED_NODE_TEST 0
#endif
// v_pk_fma_f32 saves an instruction per plane pair and still loses: 2 % on the closest-hit and 3 % on the shadow kernel
// of the 1.18 M-triangle scene (the packed form wants its operands in register pairs, and the moves cost more)
DEV Pair pairFma(Pair a, Pair b, Pair c) { // each half rounded once, like fmaf
    if (ENIGMA_PACKED_NODE_TEST) return __builtin_elementwise_fma(a, b, c);
    return Pair{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}

// Cross-lane moves of a wave (all 64 lanes must be active): gatherLane reads v from lane `src`, scatterLane sends v to
// lane `dst` (the destinations of the wave must be a permutation of the lanes).
DEV uint32_t gatherLane(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
DEV float gatherLaneF(float v, uint32_t src) { return __uint_as_float(gatherLane(__float_as_uint(v), src)); }
DEV uint32_t scatterLane(uint32_t v, uint32_t dst) { return (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)v); }

// A triangle step shared by the wave.  Lanes that hold triangles ("owners") hand up to two of them to the wave; the
// (owner, triangle) pairs -- at most 64 -- are dealt out one per lane, whoever the lane is, so that the test runs with
// as many lanes busy as there are pairs instead of once per owner for its first triangle and once more for its second.
// Pairs are numbered: the owners' first triangles in lane order, then their second ones in lane order (as many as still
// fit 64).  posA / posB: the lanes that work on this lane's pairs; worker lanes learn their owner through two
// permutations of the lane numbers.
struct TrianglePairs {
    uint32_t triA, triB;   // owner: the triangles handed over
    uint32_t posA, posB;   // owner: the lanes that test them
    bool hasA, hasB;       // owner: pairs exist
    uint32_t owner;        // worker: the lane whose ray this lane tests (valid when `work`)
    uint32_t tri;          // worker: the triangle
    bool work;
};
template <class WALKER>
DEV TrianglePairs dealTriangles(WALKER& w) {
    TrianglePairs P;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    P.hasA = w.ty != 0;
    P.triA = P.hasA ? w.takeTriangle() : 0u;
    const bool wantB = w.ty != 0;
    const unsigned long long mA = __ballot(P.hasA), mWant = __ballot(wantB);
    const uint32_t cA = __popcll(mA);
    P.posA = __popcll(mA & below);
    P.posB = cA + __popcll(mWant & below);
    P.hasB = wantB && P.posB < 64u;        // in lane order: the lanes that still fit are the first ones that want to
    P.triB = P.hasB ? w.takeTriangle() : 0u;
    const unsigned long long mB = __ballot(P.hasB);
    const uint32_t cB = __popcll(mB);
    // two permutations of the lanes: owners send their number to the lanes of their pairs, the others fill the rest
    const uint32_t dstA = P.hasA ? P.posA : cA + __popcll(~mA & below);
    const uint32_t dstB = (P.hasB ? P.posB : cA + cB + __popcll(~mB & below)) & 63u;
    const uint32_t fromA = scatterLane(lane, dstA), fromB = scatterLane(lane, dstB);
    P.work = lane < cA + cB;
    const bool second = lane >= cA;
    P.owner = second ? fromB : fromA;
    const uint32_t ta = gatherLane(P.triA, P.owner), tb = gatherLane(P.triB, P.owner);
    P.tri = second ? tb : ta;
    return P;
}
// the owner's ray as the triangle test needs it, gathered by the worker
template <class WALKER>
DEV TriRay gatherTriRay(const WALKER& w, uint32_t owner) {
    TriRay r;
    r.o = F3(gatherLaneF(w.r.o.x, owner), gatherLaneF(w.r.o.y, owner), gatherLaneF(w.r.o.z, owner));
    r.Sx = gatherLaneF(w.r.Sx, owner); r.Sy = gatherLaneF(w.r.Sy, owner); r.Sz = gatherLaneF(w.r.Sz, owner);
    const uint32_t kz = gatherLane((w.r.kz0 ? 1u : 0u) | (w.r.kz1 ? 2u : 0u), owner);
    r.kz0 = (kz & 1u) != 0; r.kz1 = (kz & 2u) != 0;
    return r;
}

// The five 16-byte pieces of a node
struct Node8Words { uint4 n0, n1, n2, n3, n4; };
template <int NS>
DEV Node8Words fetchNode8(const float4* nodes, uint32_t index) {
    const uint4* np = reinterpret_cast<const uint4*>(nodes + (size_t)index * NS);
    Node8Words w;
    w.n0 = np[0]; w.n1 = np[1]; w.n2 = np[2]; w.n3 = np[3]; w.n4 = np[4];
    pinHere(np);
    return w;
}

// One node against the ray: the node group (first inner child, hit bits in visiting order | inner mask) and the
// triangle group (first triangle, one bit per triangle of the hit leaves).
DEV void testNode8(const Node8Words& w, const Ray8& r, float tMax, uint32_t* gx, uint32_t* gy, uint32_t* tx, uint32_t* ty) {
    const uint4 n0 = w.n0, n1 = w.n1, n2 = w.n2, n3 = w.n3, n4 = w.n4;
    arithmeticBegins(); // the loads are out
    padVectorIssue<ENIGMA_EXP_PAD_NODE>();
    const uint32_t ew = n0.w;
    const float Ax = __uint_as_float((ew & 0xffu) << 23) * r.ix;
    const float Ay = __uint_as_float(((ew >> 8) & 0xffu) << 23) * r.iy;
    const float Az = __uint_as_float(((ew >> 16) & 0xffu) << 23) * r.iz;
    const float Bx = (__uint_as_float(n0.x) - r.ox) * r.ix;
    const float By = (__uint_as_float(n0.y) - r.oy) * r.iy;
    const float Bz = (__uint_as_float(n0.z) - r.oz) * r.iz;
    const float kSlack = 0x1p-21f;
    const float sx = __builtin_fmaf(256.f, fabsf(Ax), fabsf(Bx)), sy = __builtin_fmaf(256.f, fabsf(Ay), fabsf(By)), sz = __builtin_fmaf(256.f, fabsf(Az), fabsf(Bz));
    // entry and exit plane of an axis side by side (pairFma)
    const Pair Bx2 = {__builtin_fmaf(-kSlack, sx, Bx), __builtin_fmaf(kSlack, sx, Bx)};
    const Pair By2 = {__builtin_fmaf(-kSlack, sy, By), __builtin_fmaf(kSlack, sy, By)};
    const Pair Bz2 = {__builtin_fmaf(-kSlack, sz, Bz), __builtin_fmaf(kSlack, sz, Bz)};
    const Pair Ax2 = {Ax, Ax}, Ay2 = {Ay, Ay}, Az2 = {Az, Az};
    // the plane a ray enters a slab through is the low one unless it travels in the negative direction
    const uint32_t nxw[2] = {r.nx ? n3.z : n2.x, r.nx ? n3.w : n2.y}, fxw[2] = {r.nx ? n2.x : n3.z, r.nx ? n2.y : n3.w};
    const uint32_t nyw[2] = {r.ny ? n4.x : n2.z, r.ny ? n4.y : n2.w}, fyw[2] = {r.ny ? n2.z : n4.x, r.ny ? n2.w : n4.y};
    const uint32_t nzw[2] = {r.nz ? n4.z : n3.x, r.nz ? n4.w : n3.y}, fzw[2] = {r.nz ? n3.x : n4.z, r.nz ? n3.y : n4.w};
    const uint32_t metaw[2] = {n1.z, n1.w};
    uint32_t hitmask = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint32_t meta4 = metaw[half];
        const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;   // low five bits >= 24: an inner node
        const uint32_t innerMask4 = (isInner4 >> 4) * 0xffu;
        const uint32_t bitIndex4 = (meta4 ^ (r.oct4 & innerMask4)) & 0x1f1f1f1fu; // inner: 24 + (slot ^ octant); leaf: its triangle offset
        const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;                 // inner: 1; leaf: one bit per triangle
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const Pair tx = pairFma(Pair{ubyteToFloat(nxw[half], b), ubyteToFloat(fxw[half], b)}, Ax2, Bx2);
            const Pair ty = pairFma(Pair{ubyteToFloat(nyw[half], b), ubyteToFloat(fyw[half], b)}, Ay2, By2);
            const Pair tz = pairFma(Pair{ubyteToFloat(nzw[half], b), ubyteToFloat(fzw[half], b)}, Az2, Bz2);
            const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.f));
            const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tMax));
            const uint32_t bits = ((childBits4 >> (8 * b)) & 0xffu) << ((bitIndex4 >> (8 * b)) & 0xffu);
            hitmask |= tn <= tf ? bits : 0u; // a select, not a branch: the wave would skip it only when no lane hits
        }
    }
    *gx = n1.x; *gy = (hitmask & kNodeHits) | (ew >> 24);
    *tx = n1.y; *ty = hitmask & ~kNodeHits;
    pinHere(*gy, *ty); arithmeticEnds();
}

template <int NS>
DEV void intersectNode8(const float4* nodes, uint32_t index, const Ray8& r, float tMax, uint32_t* gx, uint32_t* gy, uint32_t* tx, uint32_t* ty) {
    testNode8(fetchNode8<NS>(nodes, index), r, tMax, gx, gy, tx, ty);
}

// Per-lane stack of node groups, two words an entry; the first LDSN levels in LDS (entry l of a lane at words
// (2l) * STRIDE and (2l + 1) * STRIDE of its column: a lane always uses bank lane % 32), deeper levels in a private
// array.  Behind the stack rows sit the lane's two visit tallies
// (nodes fetched, triangles tested since the block zeroed them: ds_add without return, no register lives across steps).
// LANE_TALLY = false: no tally rows; the walkers then count per wave (their voted steps know how many lanes take part),
// and a full stack is reported through `full`, a counter in global memory.
template <int LDSN, int OVF, int STRIDE, bool LANE_TALLY = true>
struct GroupStack {
    static constexpr int kRows = 2 * LDSN;
    static constexpr int kWords = kRows + (LANE_TALLY ? 2 : 0);
    uint32_t* lds;
    uint32_t* ov; // [2 * OVF], caller's frame
    unsigned long long* full;
    int sp;
    DEV GroupStack(uint32_t* column, uint32_t* spill, unsigned long long* fullCounter = nullptr) : lds(column), ov(spill), full(fullCounter), sp(0) {}
    DEV void countNode() { if (LANE_TALLY) atomicAdd(&lds[kRows * STRIDE], 1u); }
    DEV void countTriangle() { if (LANE_TALLY) atomicAdd(&lds[(kRows + 1) * STRIDE], 1u); }
    DEV void push(uint32_t x, uint32_t y) {
        if (sp < LDSN) { lds[(2 * sp) * STRIDE] = x; lds[(2 * sp + 1) * STRIDE] = y; ++sp; }
        else if (OVF > 0 && sp < LDSN + OVF) { ov[2 * (sp - LDSN)] = x; ov[2 * (sp - LDSN) + 1] = y; ++sp; }
        // cannot happen (the host builds no tree this deep): reported, not silent
        else if (LANE_TALLY) atomicOr(&lds[(kRows + 1) * STRIDE], kStackFullFlag);
        else if (full) atomicAdd(full, 1ull);
    }
    DEV bool pop(uint32_t* x, uint32_t* y) {
        if (sp == 0) return false;
        --sp;
        if (OVF > 0 && sp >= LDSN) { *x = ov[2 * (sp - LDSN)]; *y = ov[2 * (sp - LDSN) + 1]; }
        else { *x = lds[(2 * sp) * STRIDE]; *y = lds[(2 * sp + 1) * STRIDE]; }
        return true;
    }
};

// Common state machine of the two walkers.  A lane holds a node group G (children still to visit) and a triangle
// group T (triangles still to test); T is emptied before the lane takes its next node.  done: both empty, stack empty.
template <int LDSN, int OVF, int STRIDE, int NS, bool LANE_TALLY = true>
struct Walker8Base {
    RaySetup r;
    Ray8 q;
    uint32_t gx, gy, tx, ty;
    uint32_t nodeSteps, triSteps;   // wave steps of either kind this walker has made (the same in every lane of the wave)
    uint32_t nodeVisits, triVisits; // LANE_TALLY = false: nodes fetched / triangles tested by the lanes of the wave in those steps
    GroupStack<LDSN, OVF, STRIDE, LANE_TALLY> st;
    DEV Walker8Base(uint32_t* column, uint32_t* spill, unsigned long long* fullCounter = nullptr)
        : gx(0), gy(0), tx(0), ty(0), nodeSteps(0), triSteps(0), nodeVisits(0), triVisits(0), st(column, spill, fullCounter) {}
    DEV void begin(F3 o, F3 d) {
        // the hardware reciprocal (one ulp) is enough for the box distances: their slack covers it; the shear constants
        // of the triangle test stay correctly rounded quotients
        r = setupRayWith(o, d, F3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)));
        q = makeRay8(r);
        st.sp = 0;
        gx = 0; gy = 0x80000000u; // the root as the only child of a group without siblings (inner mask 0)
        tx = 0; ty = 0;
    }
    DEV void finish() { gy = 0; ty = 0; st.sp = 0; }
    DEV bool done() const { return ty == 0 && (gy & kNodeHits) == 0; }
    DEV bool atTriangles() const { return ty != 0; }
    DEV bool atNode() const { return ty == 0 && (gy & kNodeHits) != 0; }
    // takes the nearest unvisited child of G, leaves the rest of G on the stack, and intersects the child's node
    DEV void nodeStep(const float4* nodes, float tMax) {
        st.countNode();
        const uint32_t hits = gy & kNodeHits;
        const uint32_t bit = 31u - (uint32_t)__clz((int)hits);
        const uint32_t rest = gy & ~(1u << bit);
        const uint32_t slot = (bit - 24u) ^ (q.oct4 & 7u);
        const uint32_t child = gx + __popc(rest & 0xffu & ((1u << slot) - 1u)); // inner slots below it
        // (the push behind the node's loads instead of in front of them: the twenty registers of a node then live across
        // the push's branches, more is spilled, and the closest-hit kernel runs 3 % slower)
        if (rest & kNodeHits) st.push(gx, rest);
        intersectNode8<NS>(nodes, child, q, tMax, &gx, &gy, &tx, &ty);
    }
    // when G and T are both empty: the next group from the stack, if any
    DEV void refill() {
        if (ty == 0 && (gy & kNodeHits) == 0 && !st.pop(&gx, &gy)) gy = 0;
    }
    DEV uint32_t takeTriangle() { // lowest set bit of T
        const uint32_t b = (uint32_t)__ffs((int)ty) - 1u;
        ty &= ty - 1u;
        st.countTriangle();
        return tx + b;
    }
};

// Degenerate triangles report no hit (Enigma/TriangleMesh.cpp:238-256): Triangle::intersect gives up when its dpdu x dpdv
// vanishes (or its uvs are degenerate) AND the geometric normal Cross(p2 - p0, p1 - p0) vanishes too.  A triangle with a
// normal is therefore never refused, and only one without needs dpdu and dpdv -- from the mesh's own uvs where it has
// them (shade: four float4 per triangle, the uvs in the .w of the first three and in the fourth; mw: the triangle's
// material word, bit 28 = the mesh has uvs), else from the default uvs (0,0), (1,0), (1,1).  A collinear sliver can pass
// the hit test; whether it is reported then hangs on the last bits of these products, so they are the reference's.
DEV bool triangleAccepted(F3 p0, F3 p1, F3 p2, uint32_t mw, const float4* shade, uint32_t tri) {
    if (lengthSqr(cross(p2 - p0, p1 - p0)) != 0) return true;
    F3 dp02 = p0 - p2, dp12 = p1 - p2;
    F3 dpdu = dp12 - dp02, dpdv = -dp12;
    if (shade && ((mw >> 28) & 1u)) {
        const float4* sp = shade + (size_t)tri * 4;
        const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], vv = sp[3];
        float du02 = s0.w - s2.w, dv02 = vv.x - vv.z, du12 = s1.w - s2.w, dv12 = vv.y - vv.z;
        float determinant = du02 * dv12 - dv02 * du12;
        if (fabsf(determinant) < 1e-8) return false;
        float invdet = 1 / determinant;
        dpdu = (dv12 * dp02 - dv02 * dp12) * invdet;
        dpdv = (-du12 * dp02 + du02 * dp12) * invdet;
    }
    return lengthSqr(cross(dpdu, dpdv)) != 0;
}

// what the walkers of scenes with spheres know of them (uniform values)
struct SphereRefs {
    const DevSphere* table;
    uint32_t base; // primitive number of sphere 0
    const float4* shade; // per-triangle normals and uvs of meshes that have them (triangleAccepted), nullptr if none has
    DEV SphereRefs() : table(nullptr), base(0), shade(nullptr) {}
    DEV SphereRefs(const DevScene& sc) : table(sc.spheres), base(sc.nTris), shade(sc.triShade) {}
};
DEV bool slotIsSphere(float4 v0) { return (__float_as_uint(v0.w) >> 31) != 0; }
DEV uint32_t slotSphere(float4 v0) { return __float_as_uint(v0.w) & 0x7fffffffu; }

// SHARED_TRI: triangle steps dealt out over the wave (scenes in global memory; for a scene in LDS, whose triangle
// fetches cost next to nothing, the dealing costs more than the second test it saves: 0.64 -> 0.71 ms per Cornell launch)
template <int LDSN, int OVF, int STRIDE, int NS = kNode8Vec4, bool SPH = false, bool LANE_TALLY = true, bool SHARED_TRI = true>
struct ClosestWalker8 : Walker8Base<LDSN, OVF, STRIDE, NS, LANE_TALLY> {
    using Base = Walker8Base<LDSN, OVF, STRIDE, NS, LANE_TALLY>;
    static constexpr int kWords = GroupStack<LDSN, OVF, STRIDE, LANE_TALLY>::kWords, kRows = GroupStack<LDSN, OVF, STRIDE, LANE_TALLY>::kRows;
    Hit h;
    SphereRefs spheres;
    DEV ClosestWalker8(uint32_t* column, uint32_t* spill, SphereRefs sr = SphereRefs(), unsigned long long* fullCounter = nullptr)
        : Base(column, spill, fullCounter), spheres(sr) { h.tri = -1; h.t = 0.f; h.b0 = h.b1 = h.b2 = 0.f; }
    DEV void start(F3 o, F3 d, float tMax) {
        Base::begin(o, d);
        h.tri = -1; h.t = tMax; h.b0 = h.b1 = h.b2 = 0.f;
    }
    DEV void testOne(const float4* tris) {
        const uint32_t tri = Base::takeTriangle();
        const float4* tp = tris + (size_t)tri * 3;
        float4 v0 = tp[0];
        if (SPH && slotIsSphere(v0)) { // Sphere::intersect
            F3 pHit, dObj;
            float th;
            const uint32_t i = slotSphere(v0);
            if (sphereHit(spheres.table[i], this->r.o, this->r.d, h.t, &pHit, &dObj, &th)) { h.t = th; h.tri = (int)(spheres.base + i); h.b0 = pHit.x; h.b1 = pHit.y; h.b2 = pHit.z; }
            return;
        }
        float4 v1 = tp[1], v2 = tp[2];
        pinHere(tp); arithmeticBegins();
        float t, b0, b1, b2;
        if (triangleTest(this->r, h.t, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2) && triangleAccepted(xyz(v0), xyz(v1), xyz(v2), __float_as_uint(v1.w), spheres.shade, tri)) {
            h.t = t; h.tri = (int)tri; h.b0 = b0; h.b1 = b1; h.b2 = b2;
        }
        pinHere(h.t); arithmeticEnds();
    }
    // The wave's triangle step (scenes without spheres): up to 64 (owner, triangle) pairs, one per lane.  A pair is tested
    // with the distance its owner had when the step began; an owner applies its first pair's result, then its second
    // pair's -- in the order the reference tests them -- repeating the one comparison that involves the ray's tMax if
    // the first has shortened the ray meanwhile (triangleStillInRange).  Same operations on the same operands as one
    // test after the other: hits, distances and barycentrics are bit for bit those of testOne.
    DEV void sharedTriangleStep(const float4* tris) {
        TrianglePairs P = dealTriangles(*this);
        const TriRay ray = gatherTriRay(*this, P.owner);
        const float tMaxOwner = gatherLaneF(h.t, P.owner);
        float t = -1.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, tScaled = 0.f, det = 0.f;
        if (P.work) {
            const float4* tp = tris + (size_t)P.tri * 3;
            const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            pinHere(tp); arithmeticBegins();
            float tt, c0, c1, c2, ts, dd;
            if (triangleTestEx(ray, tMaxOwner, xyz(v0), xyz(v1), xyz(v2), &tt, &c0, &c1, &c2, &ts, &dd) &&
                triangleAccepted(xyz(v0), xyz(v1), xyz(v2), __float_as_uint(v1.w), spheres.shade, P.tri)) {
                t = tt; b0 = c0; b1 = c1; b2 = c2; tScaled = ts; det = dd;
            }
        }
        pinHere(t); arithmeticEnds();
        // back to the owners (a hit has t > 0, a miss carries -1)
        const float tA = gatherLaneF(t, P.posA), tB = gatherLaneF(t, P.posB & 63u);
        const bool hitA = P.hasA && tA > 0.f;
        bool hitB = P.hasB && tB > 0.f;
        if (__ballot(hitA || hitB) == 0) return;
        const float a0 = gatherLaneF(b0, P.posA), a1 = gatherLaneF(b1, P.posA), a2 = gatherLaneF(b2, P.posA);
        if (hitA) { h.t = tA; h.tri = (int)P.triA; h.b0 = a0; h.b1 = a1; h.b2 = a2; }
        if (__ballot(hitB) == 0) return;
        const uint32_t pb = P.posB & 63u;
        const float sB = gatherLaneF(tScaled, pb), dB = gatherLaneF(det, pb);
        const float e0 = gatherLaneF(b0, pb), e1 = gatherLaneF(b1, pb), e2 = gatherLaneF(b2, pb);
        if (hitB && hitA) hitB = triangleStillInRange(sB, dB, h.t);
        if (hitB) { h.t = tB; h.tri = (int)P.triB; h.b0 = e0; h.b1 = e1; h.b2 = e2; }
    }
    // one kind of work per wave step, whichever has more lanes waiting.  FULL_WAVE: the caller keeps all 64 lanes of the
    // wave in its loop (the pooled kernels do); otherwise the shared triangle step, which moves data between lanes, is
    // taken only when the wave happens to be complete.
    template <bool FULL_WAVE = true>
    DEV void step(const float4* nodes, const float4* tris) {
        const bool n = Base::atNode(), t = Base::atTriangles();
        const uint32_t cn = __popcll(__ballot(n)), ct = __popcll(__ballot(t));
        if (cn >= ct) {
            this->nodeSteps = uniformAdd(this->nodeSteps, 1u);
            this->nodeVisits = uniformAdd(this->nodeVisits, cn);
            if (!n) return;
            Base::nodeStep(nodes, h.t);
        } else if (!SPH && SHARED_TRI && kSharedTriangleStepsClosest && (FULL_WAVE || __ballot(true) == ~0ull)) {
            this->triSteps = uniformAdd(this->triSteps, 1u);
            const uint32_t before = LANE_TALLY ? 0u : (uint32_t)__popcll(__ballot(this->ty != 0)) + (uint32_t)__popcll(__ballot((this->ty & (this->ty - 1u)) != 0));
            sharedTriangleStep(tris);
            this->triVisits = uniformAdd(this->triVisits, ct + (LANE_TALLY ? 0u : min(before, 64u) - ct));
        } else {
            this->triSteps = uniformAdd(this->triSteps, 1u);
            if (t) testOne(tris);
            const bool second = t && this->ty != 0;
            this->triVisits = uniformAdd(this->triVisits, ct + (LANE_TALLY ? 0u : (uint32_t)__popcll(__ballot(second)))); // that vote is for the count only
            if (!second) { if (t) Base::refill(); return; }
            testOne(tris);
        }
        Base::refill();
    }
    // descend until the lane holds triangles (or is through), then test them all
    DEV void round(const float4* nodes, const float4* tris) {
        while (Base::atNode()) { Base::nodeStep(nodes, h.t); Base::refill(); }
        while (this->ty != 0) testOne(tris);
        Base::refill();
    }
    template <bool ROUNDS> DEV void advance(const float4* nodes, const float4* tris) { if (ROUNDS) round(nodes, tris); else step(nodes, tris); }
};

template <int LDSN, int OVF, int STRIDE, int NS = kNode8Vec4, bool SPH = false, bool LANE_TALLY = true>
struct AnyWalker8 : Walker8Base<LDSN, OVF, STRIDE, NS, LANE_TALLY> {
    using Base = Walker8Base<LDSN, OVF, STRIDE, NS, LANE_TALLY>;
    static constexpr int kWords = GroupStack<LDSN, OVF, STRIDE, LANE_TALLY>::kWords, kRows = GroupStack<LDSN, OVF, STRIDE, LANE_TALLY>::kRows;
    float tMax;
    bool occluded;
    SphereRefs spheres;
    DEV AnyWalker8(uint32_t* column, uint32_t* spill, SphereRefs sr = SphereRefs(), unsigned long long* fullCounter = nullptr)
        : Base(column, spill, fullCounter), tMax(0.f), occluded(false), spheres(sr) {}
    DEV void start(F3 o, F3 d, float tMaxRay) {
        Base::begin(o, d);
        tMax = tMaxRay;
        occluded = false;
    }
    DEV void testOne(const float4* tris) { // Triangle::intersectP
        const uint32_t tri = Base::takeTriangle();
        const float4* tp = tris + (size_t)tri * 3;
        float4 v0 = tp[0];
        if (SPH && slotIsSphere(v0)) { // Sphere::intersectP
            F3 pHit, dObj;
            float th;
            if (sphereHit(spheres.table[slotSphere(v0)], this->r.o, this->r.d, tMax, &pHit, &dObj, &th)) { occluded = true; Base::finish(); }
            return;
        }
        float4 v1 = tp[1], v2 = tp[2];
        pinHere(tp); arithmeticBegins();
        float t, b0, b1, b2;
        if (triangleTest(this->r, tMax, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2)) { occluded = true; Base::finish(); }
        pinHere(this->gy, this->ty); arithmeticEnds();
    }
    // the wave's triangle step, as ClosestWalker8::sharedTriangleStep: any pair that hits ends its owner's ray
    DEV void sharedTriangleStep(const float4* tris) {
        TrianglePairs P = dealTriangles(*this);
        const TriRay ray = gatherTriRay(*this, P.owner);
        const float tMaxOwner = gatherLaneF(tMax, P.owner);
        uint32_t hit = 0u;
        if (P.work) {
            const float4* tp = tris + (size_t)P.tri * 3;
            const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            pinHere(tp); arithmeticBegins();
            float t, b0, b1, b2, ts, dd;
            hit = triangleTestEx(ray, tMaxOwner, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2, &ts, &dd) ? 1u : 0u;
        }
        pinHere(hit, hit); arithmeticEnds();
        if (__ballot(hit != 0u) == 0) return;
        const uint32_t hA = gatherLane(hit, P.posA), hB = gatherLane(hit, P.posB & 63u);
        if ((P.hasA && hA) || (P.hasB && hB)) { occluded = true; Base::finish(); }
    }
    template <bool FULL_WAVE = true>
    DEV void step(const float4* nodes, const float4* tris) {
        const bool n = Base::atNode(), t = Base::atTriangles();
        const uint32_t cn = __popcll(__ballot(n)), ct = __popcll(__ballot(t));
        if (cn >= ct) {
            this->nodeSteps = uniformAdd(this->nodeSteps, 1u);
            this->nodeVisits = uniformAdd(this->nodeVisits, cn);
            if (!n) return;
            Base::nodeStep(nodes, tMax);
        } else if (!SPH && kSharedTriangleStepsAny && (FULL_WAVE || __ballot(true) == ~0ull)) {
            this->triSteps = uniformAdd(this->triSteps, 1u);
            const uint32_t before = LANE_TALLY ? 0u : (uint32_t)__popcll(__ballot(this->ty != 0)) + (uint32_t)__popcll(__ballot((this->ty & (this->ty - 1u)) != 0));
            sharedTriangleStep(tris);
            this->triVisits = uniformAdd(this->triVisits, ct + (LANE_TALLY ? 0u : min(before, 64u) - ct));
        } else {
            this->triSteps = uniformAdd(this->triSteps, 1u);
            if (t) testOne(tris);
            const bool second = t && this->ty != 0;
            this->triVisits = uniformAdd(this->triVisits, ct + (LANE_TALLY ? 0u : (uint32_t)__popcll(__ballot(second))));
            if (!second) { if (t) Base::refill(); return; }
            testOne(tris);
        }
        Base::refill();
    }
    DEV void round(const float4* nodes, const float4* tris) {
        while (Base::atNode()) { Base::nodeStep(nodes, tMax); Base::refill(); }
        while (this->ty != 0) testOne(tris);
        Base::refill();
    }
    template <bool ROUNDS> DEV void advance(const float4* nodes, const float4* tris) { if (ROUNDS) round(nodes, tris); else step(nodes, tris); }
};

// ---- the walkers of a kernel ----
// NS: 16-byte pieces between consecutive nodes (the same in HBM and in a block's LDS copy of a small scene: five
// pieces put consecutive nodes on different banks already).
constexpr int kLdsNodePitch = kNode8Vec4;
template <int LDSN, int OVF, int STRIDE, bool LDS_SCENE, bool SPH = false, bool LANE_TALLY = true> struct WalkerSel {
    static constexpr int NS = LDS_SCENE ? kLdsNodePitch : kNode8Vec4;
    using Closest = ClosestWalker8<LDSN, OVF, STRIDE, NS, SPH, LANE_TALLY, !LDS_SCENE>;
    using Any = AnyWalker8<LDSN, OVF, STRIDE, NS, SPH, LANE_TALLY>;
};
constexpr int spillWords(int ovf) { return 2 * (ovf > 0 ? ovf : 1); }

// one ray per lane, start to finish (kernels that do not pool rays); column: the lane's LDS column of Closest::kWords rows
template <int LDSN, int OVF, int STRIDE, bool LDS_SCENE = false, bool SPH = false>
DEV Hit traceClosest(const float4* nodes, const float4* tris, F3 o, F3 d, float tMax, uint32_t* column, SphereRefs sr = SphereRefs()) {
    uint32_t spill[spillWords(OVF)];
    typename WalkerSel<LDSN, OVF, STRIDE, LDS_SCENE, SPH>::Closest w(column, spill, sr);
    w.start(o, d, tMax);
    while (__ballot(!w.done()) != 0) w.template step<false>(nodes, tris); // step() votes wave-wide: the caller's lanes stay in the loop together
    return w.h;
}
template <int LDSN, int OVF, int STRIDE, bool SPH = false>
DEV bool traceAny(const float4* nodes, const float4* tris, F3 o, F3 d, float tMax, uint32_t* column, SphereRefs sr = SphereRefs()) {
    uint32_t spill[spillWords(OVF)];
    typename WalkerSel<LDSN, OVF, STRIDE, false, SPH>::Any w(column, spill, sr);
    w.start(o, d, tMax);
    while (__ballot(!w.done()) != 0) w.template step<false>(nodes, tris);
    return w.occluded;
}

} // namespace edev
