// Ray / box / triangle tests and per-lane stack traversal of the device BVH.
//
// Replaces BVHAccel::intersect / intersectP (Enigma/BVH.cpp:639-722), the slab test
// BBox3::intersectP (Enigma/BBox.h:267-295) and the watertight Triangle::intersects
// (Enigma/TriangleMesh.cpp:491-602).  The arithmetic of the two tests is the reference's,
// operation for operation, so hit/miss decisions, t and barycentrics are bit-identical;
// only the tree being walked differs.
#pragma once
#include "dev_types.h"

namespace edev {

struct RaySetup {
    F3 o, d, invDir;
    bool negX, negY, negZ;   // dirIsNeg
    bool kz0, kz1;           // largest |d| component is x (kz0) / y (kz1) / z (neither)
    float Sx, Sy, Sz;        // shear constants (depend on the ray only)
};

DEV F3 permuteK(F3 v, bool kz0, bool kz1) {
    // (kx,ky,kz) = (1,2,0) | (2,0,1) | (0,1,2)
    return F3(kz0 ? v.y : (kz1 ? v.z : v.x), kz0 ? v.z : (kz1 ? v.x : v.y), kz0 ? v.x : (kz1 ? v.y : v.z));
}

// invDir: 1 / d as the slab test in use needs it (the reference's own test takes the correctly rounded quotient)
DEV RaySetup setupRayWith(F3 o, F3 d, F3 invDir) {
    RaySetup r;
    r.o = o; r.d = d;
    r.invDir = invDir;
    r.negX = r.invDir.x < 0; r.negY = r.invDir.y < 0; r.negZ = r.invDir.z < 0;
    // MaxDimension(Abs(d)), MathCommon.h:342-353
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int kz;
    if (ax > ay) kz = (ax > az) ? 0 : 2;
    else kz = (ay > az) ? 1 : 2;
    r.kz0 = kz == 0; r.kz1 = kz == 1;
    F3 dp = permuteK(d, r.kz0, r.kz1);
    r.Sx = -dp.x / dp.z; r.Sy = -dp.y / dp.z; r.Sz = 1.f / dp.z;
    return r;
}
DEV RaySetup setupRay(F3 o, F3 d) { return setupRayWith(o, d, F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z)); } // Inverted(), MathCommon.h:225-230

// BBox3::intersectP(ray, invDir, dirIsNeg), Enigma/BBox.h:267-295.  tEntry = the final tMin.
DEV bool slabTest(const RaySetup& r, float tMaxRay, F3 lo, F3 hi, float* tEntry) {
    constexpr float kRobust = 1 + 2 * gammaf(3);
    float tMin = ((r.negX ? hi.x : lo.x) - r.o.x) * r.invDir.x;
    float tMax = ((r.negX ? lo.x : hi.x) - r.o.x) * r.invDir.x;
    float tyMin = ((r.negY ? hi.y : lo.y) - r.o.y) * r.invDir.y;
    float tyMax = ((r.negY ? lo.y : hi.y) - r.o.y) * r.invDir.y;
    tMax *= kRobust;
    tyMax *= kRobust;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((r.negZ ? hi.z : lo.z) - r.o.z) * r.invDir.z;
    float tzMax = ((r.negZ ? lo.z : hi.z) - r.o.z) * r.invDir.z;
    tzMax *= kRobust;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    *tEntry = tMin;
    return (tMin < tMaxRay) && (tMax > 0);
}

// The same test with the near / far plane of each axis already picked (the picks depend on the ray only,
// so the traversal fetches "near x of four children" straight from the node instead of selecting per child).
DEV bool slabTestNF(const RaySetup& r, float tMaxRay, float nx, float ny, float nz, float fx, float fy, float fz, float* tEntry) {
    constexpr float kRobust = 1 + 2 * gammaf(3);
    float tMin = (nx - r.o.x) * r.invDir.x;
    float tMax = (fx - r.o.x) * r.invDir.x;
    float tyMin = (ny - r.o.y) * r.invDir.y;
    float tyMax = (fy - r.o.y) * r.invDir.y;
    tMax *= kRobust;
    tyMax *= kRobust;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = (nz - r.o.z) * r.invDir.z;
    float tzMax = (fz - r.o.z) * r.invDir.z;
    tzMax *= kRobust;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    *tEntry = tMin;
    return (tMin < tMaxRay) && (tMax > 0);
}

// Triangle::intersects, Enigma/TriangleMesh.cpp:491-602.
DEV bool triangleTest(const RaySetup& r, float tMaxRay, F3 p0, F3 p1, F3 p2, float* tOut, float* b0o, float* b1o, float* b2o) {
    F3 p0t = permuteK(p0 - r.o, r.kz0, r.kz1);
    F3 p1t = permuteK(p1 - r.o, r.kz0, r.kz1);
    F3 p2t = permuteK(p2 - r.o, r.kz0, r.kz1);
    p0t.x += r.Sx * p0t.z; p0t.y += r.Sy * p0t.z;
    p1t.x += r.Sx * p1t.z; p1t.y += r.Sy * p1t.z;
    p2t.x += r.Sx * p2t.z; p2t.y += r.Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= r.Sz; p1t.z *= r.Sz; p2t.z *= r.Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMaxRay * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMaxRay * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = maxComponent(fabs3(F3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = gammaf(3) * maxZt;
    float maxXt = maxComponent(fabs3(F3(p0t.x, p1t.x, p2t.x)));
    float maxYt = maxComponent(fabs3(F3(p0t.y, p1t.y, p2t.y)));
    float deltaX = gammaf(5) * (maxXt + maxZt);
    float deltaY = gammaf(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammaf(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = maxComponent(fabs3(F3(e0, e1, e2)));
    float deltaT = 3 * (gammaf(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return false;
    *tOut = t; *b0o = b0; *b1o = b1; *b2o = b2;
    return true;
}

struct Hit { int tri; float t, b0, b1, b2; };

// Per-launch tallies of the traversal work: for each kind of ray query (closest hit of path rays, any hit of shadow
// rays, closest hit of MIS rays) the nodes fetched and the triangles tested, summed per block into one of kVisitSlots
// slots (block b adds to slot b mod kVisitSlots; the adds of different blocks go to different words as long as a
// grid has at most kVisitSlots blocks, which the launch code keeps).  Layout: [kind][0 = nodes, 1 = triangles][slot].
constexpr uint32_t kVisitSlots = 8192;
enum VisitKind : uint32_t { kVisitClosest = 0, kVisitShadow = 1, kVisitMis = 2, kVisitKinds = 3 };
DEV void blockZeroVisits(uint32_t* column, int rows, int stride) { column[rows * stride] = 0u; column[(rows + 1) * stride] = 0u; }
// every thread of the block calls it once, after its last traversal step; red: 2 * (block size / 64) words of LDS
constexpr uint32_t kVisitTallies = 4;
// a wave's step counts (uniform values: one lane adds them)
DEV void waveFlushSteps(unsigned long long* visits, uint32_t kind, uint32_t nodeSteps, uint32_t triSteps) {
    if (!visits || (threadIdx.x & 63u) != 0) return;
    const uint32_t slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kVisitSlots;
    atomicAdd(&visits[((size_t)kind * kVisitTallies + 2) * kVisitSlots + slot], (unsigned long long)nodeSteps);
    atomicAdd(&visits[((size_t)kind * kVisitTallies + 3) * kVisitSlots + slot], (unsigned long long)triSteps);
}
DEV void blockFlushVisits(const uint32_t* column, int rows, int stride, unsigned long long* visits, uint32_t kind, uint32_t* red) {
    if (!visits) return;
    uint32_t n = column[rows * stride], t = column[(rows + 1) * stride];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n += __shfl_xor(n, d); t += __shfl_xor(t, d); }
    const uint32_t wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63u) == 0) { red[wid] = n; red[nw + wid] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sn = 0, stt = 0;
        for (uint32_t w = 0; w < nw; ++w) { sn += red[w]; stt += red[nw + w]; }
        const uint32_t slot = blockIdx.x % kVisitSlots;
        atomicAdd(&visits[((size_t)kind * kVisitTallies + 0) * kVisitSlots + slot], sn);
        atomicAdd(&visits[((size_t)kind * kVisitTallies + 1) * kVisitSlots + slot], stt);
    }
}

// One 4-wide node (see bvh_build.h): child boxes in structure-of-arrays form, seven float4 loads.
// Planes come out as near / far along the ray: the float4 index of "near x" is 0 or 3 by the sign of d.x, etc.
struct Node4 { float4 nx, ny, nz, fx, fy, fz; int c0, c1, c2, c3; };
DEV Node4 loadNode4(const float4* n, const RaySetup& r) {
    Node4 q;
    const int ox = r.negX ? 3 : 0, oy = r.negY ? 3 : 0, oz = r.negZ ? 3 : 0;
    q.nx = n[ox]; q.ny = n[1 + oy]; q.nz = n[2 + oz];
    q.fx = n[3 - ox]; q.fy = n[4 - oy]; q.fz = n[5 - oz];
    float4 ch = n[6];
    q.c0 = __float_as_int(ch.x); q.c1 = __float_as_int(ch.y); q.c2 = __float_as_int(ch.z); q.c3 = __float_as_int(ch.w);
    return q;
}
// compare-exchange of (entry distance, child) pairs: afterwards ta <= tb
DEV void orderPair(float& ta, int& ca, float& tb, int& cb) {
    bool s = tb < ta;
    float tl = s ? tb : ta, th = s ? ta : tb;
    int cl = s ? cb : ca, chh = s ? ca : cb;
    ta = tl; tb = th; ca = cl; cb = chh;
}

// Per-lane traversal stack.  The first LDSN levels live in LDS: entry (level, lane) at
// stack[level * STRIDE + lane], so a lane always uses LDS bank (lane % 32) whatever its depth --
// conflict free for the two 32-lane groups of ds_read_b32 / ds_write_b32.  Levels LDSN .. LDSN+OVF-1
// spill to a per-lane private array (scratch memory); with near-child-first descent almost no ray
// gets that deep, so LDS per block -- and with it occupancy -- no longer grows with the tree depth.
// The closest-hit stack keeps the far child's entry distance beside its reference (second plane at
// +LDSN*STRIDE words): a popped node whose entry distance is no longer below the current hit distance
// is dropped without being fetched -- the same decision the reference takes when it visits the node
// (tMin < ray.m_maxT, Enigma/BBox.h:294).
constexpr int kTraverseDone = (int)0x80000000; // never a valid leaf reference (nTris < 2^28); also the unused-slot marker of a node

// Behind the stack planes every lane owns two more LDS words (rows kRows and kRows + 1 of its column): the number of
// tree nodes it has fetched and of triangles it has tested since the block zeroed them.  ds_add without return, no
// register lives across steps for them; the block sums them up once, when it ends (blockFlushVisits).
template <int LDSN, int OVF, int STRIDE, bool DIST>
struct LaneStack {
    static constexpr int kRows = DIST ? 2 * LDSN : LDSN; // words of a column used by the stack itself
    static constexpr int kWords = kRows + 2;             // ... and with the two visit counters
    uint32_t* lds;
    uint32_t* ovN;   // spill arrays live in the caller's frame as objects of their own: inside this struct
    float* ovT;      // their run-time indexing would drag every other member into scratch memory with them
    int sp;
    DEV LaneStack(uint32_t* column, uint32_t* spillNodes, float* spillDist) : lds(column), ovN(spillNodes), ovT(spillDist), sp(0) {}
    DEV void countNode() { atomicAdd(&lds[kRows * STRIDE], 1u); }
    DEV void countTriangle() { atomicAdd(&lds[(kRows + 1) * STRIDE], 1u); }
    DEV void push(int node, float t) {
        if (sp < LDSN) {
            lds[sp * STRIDE] = (uint32_t)node;
            if (DIST) lds[(LDSN + sp) * STRIDE] = __float_as_uint(t);
            ++sp;
        } else if (OVF > 0 && sp < LDSN + OVF) {
            ovN[sp - LDSN] = (uint32_t)node;
            if (DIST) ovT[sp - LDSN] = t;
            ++sp;
        }
    }
    // next node whose entry distance is still below tMax (any node when !DIST), or kTraverseDone
    DEV int pop(float tMax) {
        while (sp > 0) {
            --sp;
            if (OVF > 0 && sp >= LDSN) {
                if (!DIST || ovT[sp - LDSN] < tMax) return (int)ovN[sp - LDSN];
            } else {
                if (!DIST || __uint_as_float(lds[(LDSN + sp) * STRIDE]) < tMax) return (int)lds[sp * STRIDE];
            }
        }
        return kTraverseDone;
    }
};

// Traversal in wave-wide steps.  A lane is either at an inner node, at a leaf triangle, or done.  Each
// step the wave does ONE kind of work -- a node step for the lanes at inner nodes, or one triangle test
// for the lanes at leaves -- whichever has more lanes waiting; the others sit the step out.  With 64
// incoherent rays per wave this keeps at least half of the unfinished lanes busy in every step, where
// "descend until every lane has a leaf, then test leaves" leaves most lanes waiting for the slowest.
// Each ray still sees its own nodes and triangles in the same depth-first order.  The walkers keep a
// ray's whole traversal state, which lets a kernel hand a finished lane a new ray between steps.
// A leaf in progress is held in cur as ~((next triangle << 3) | (triangles left - 1)).  A triangle step tests up to
// two triangles of the lane's leaf: fewer votes and fetch round trips per leaf (BigMesh: 8 % faster than one per step;
// all four at once is no better, lanes with short leaves then wait for the long ones).
constexpr int kLeafTrisPerStep = 2;
// NS: float4s between consecutive nodes (8 in HBM; 9 in the LDS copy, which staggers the nodes over the banks)
template <int LDSN, int OVF, int STRIDE, int NS = 8>
struct ClosestWalker {
    RaySetup r;
    Hit h;
    int cur;
    uint32_t nodeSteps = 0, triSteps = 0; // wave steps of either kind (the same in every lane of the wave)
    LaneStack<LDSN, OVF, STRIDE, true> st;
    static constexpr int kWords = LaneStack<LDSN, OVF, STRIDE, true>::kWords, kRows = LaneStack<LDSN, OVF, STRIDE, true>::kRows;
    DEV ClosestWalker(uint32_t* column, uint32_t* spillNodes, float* spillDist) : cur(kTraverseDone), st(column, spillNodes, spillDist) { h.tri = -1; h.t = 0.f; h.b0 = h.b1 = h.b2 = 0.f; }
    // one spill array of 2 * max(OVF, 1) words in the caller's frame: node references, then entry distances
    DEV ClosestWalker(uint32_t* column, uint32_t* spill) : ClosestWalker(column, spill, reinterpret_cast<float*>(spill + (OVF > 0 ? OVF : 1))) {}
    DEV void start(F3 o, F3 d, float tMax) {
        r = setupRay(o, d);
        h.tri = -1; h.t = tMax; h.b0 = h.b1 = h.b2 = 0.f;
        st.sp = 0;
        cur = 0;
    }
    DEV bool done() const { return cur == kTraverseDone; }
    DEV void step(const float4* nodes, const float4* tris) {
        const bool atNode = cur >= 0;
        const bool atLeaf = cur < 0 && cur != kTraverseDone;
        if (__popcll(__ballot(atNode)) >= __popcll(__ballot(atLeaf))) {
            ++nodeSteps;
            if (!atNode) return;
            st.countNode();
            Node4 q = loadNode4(nodes + (size_t)cur * NS, r);
            // children the ray misses (or unused slots) get entry distance +inf and sort to the back
            float t0, t1, t2, t3;
            if (!(q.c0 != kTraverseDone && slabTestNF(r, h.t, q.nx.x, q.ny.x, q.nz.x, q.fx.x, q.fy.x, q.fz.x, &t0))) t0 = EDEV_INF;
            if (!(q.c1 != kTraverseDone && slabTestNF(r, h.t, q.nx.y, q.ny.y, q.nz.y, q.fx.y, q.fy.y, q.fz.y, &t1))) t1 = EDEV_INF;
            if (!(q.c2 != kTraverseDone && slabTestNF(r, h.t, q.nx.z, q.ny.z, q.nz.z, q.fx.z, q.fy.z, q.fz.z, &t2))) t2 = EDEV_INF;
            if (!(q.c3 != kTraverseDone && slabTestNF(r, h.t, q.nx.w, q.ny.w, q.nz.w, q.fx.w, q.fy.w, q.fz.w, &t3))) t3 = EDEV_INF;
            orderPair(t0, q.c0, t1, q.c1); orderPair(t2, q.c2, t3, q.c3);
            orderPair(t0, q.c0, t2, q.c2); orderPair(t1, q.c1, t3, q.c3);
            orderPair(t1, q.c1, t2, q.c2);
            if (t0 == EDEV_INF) { cur = st.pop(h.t); return; }
            if (t3 != EDEV_INF) st.push(q.c3, t3); // far to near, so the nearest pending child pops first
            if (t2 != EDEV_INF) st.push(q.c2, t2);
            if (t1 != EDEV_INF) st.push(q.c1, t1);
            cur = q.c0;
        } else {
            ++triSteps;
            if (!atLeaf) return;
            uint32_t leaf = ~(uint32_t)cur;
            uint32_t tri = leaf >> 3, left = leaf & 7u;
            for (int rep = 0;; ++rep) { // up to kLeafTrisPerStep triangles of the leaf
                st.countTriangle();
                const float4* tp = tris + (size_t)tri * 3;
                float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                float t, b0, b1, b2;
                if (triangleTest(r, h.t, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2)) {
                    // degenerate triangles report no hit (Enigma/TriangleMesh.cpp:249-256)
                    F3 p0 = xyz(v0), p1 = xyz(v1), p2 = xyz(v2);
                    F3 dpdu = (p1 - p2) - (p0 - p2);
                    F3 dpdv = -(p1 - p2);
                    bool ok = true;
                    if (lengthSqr(cross(dpdu, dpdv)) == 0) ok = lengthSqr(cross(p2 - p0, p1 - p0)) != 0;
                    if (ok) { h.t = t; h.tri = (int)tri; h.b0 = b0; h.b1 = b1; h.b2 = b2; }
                }
                if (!left || rep == kLeafTrisPerStep - 1) break;
                ++tri; --left;
            }
            cur = left ? (int)~(((tri + 1u) << 3) | (left - 1u)) : st.pop(h.t);
        }
    }
    // The other schedule ("while-while"): descend until every lane of the wave holds a leaf (or is done),
    // then test the leaves.  Cheaper per step (no votes); better when rays of a wave stay together, as they
    // do in the shallow tree of a scene small enough to be traversed out of LDS.
    DEV void round(const float4* nodes, const float4* tris) {
        while (cur >= 0) {
            st.countNode();
            Node4 q = loadNode4(nodes + (size_t)cur * NS, r);
            float t0, t1, t2, t3;
            if (!(q.c0 != kTraverseDone && slabTestNF(r, h.t, q.nx.x, q.ny.x, q.nz.x, q.fx.x, q.fy.x, q.fz.x, &t0))) t0 = EDEV_INF;
            if (!(q.c1 != kTraverseDone && slabTestNF(r, h.t, q.nx.y, q.ny.y, q.nz.y, q.fx.y, q.fy.y, q.fz.y, &t1))) t1 = EDEV_INF;
            if (!(q.c2 != kTraverseDone && slabTestNF(r, h.t, q.nx.z, q.ny.z, q.nz.z, q.fx.z, q.fy.z, q.fz.z, &t2))) t2 = EDEV_INF;
            if (!(q.c3 != kTraverseDone && slabTestNF(r, h.t, q.nx.w, q.ny.w, q.nz.w, q.fx.w, q.fy.w, q.fz.w, &t3))) t3 = EDEV_INF;
            orderPair(t0, q.c0, t1, q.c1); orderPair(t2, q.c2, t3, q.c3);
            orderPair(t0, q.c0, t2, q.c2); orderPair(t1, q.c1, t3, q.c3);
            orderPair(t1, q.c1, t2, q.c2);
            if (t0 == EDEV_INF) { cur = st.pop(h.t); continue; }
            if (t3 != EDEV_INF) st.push(q.c3, t3);
            if (t2 != EDEV_INF) st.push(q.c2, t2);
            if (t1 != EDEV_INF) st.push(q.c1, t1);
            cur = q.c0;
        }
        if (cur == kTraverseDone) return;
        uint32_t leaf = ~(uint32_t)cur;
        uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
        for (uint32_t i = 0; i < cnt; ++i) {
            st.countTriangle();
            const float4* tp = tris + (size_t)(first + i) * 3;
            float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            float t, b0, b1, b2;
            if (triangleTest(r, h.t, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2)) {
                F3 p0 = xyz(v0), p1 = xyz(v1), p2 = xyz(v2);
                F3 dpdu = (p1 - p2) - (p0 - p2);
                F3 dpdv = -(p1 - p2);
                bool ok = true;
                if (lengthSqr(cross(dpdu, dpdv)) == 0) ok = lengthSqr(cross(p2 - p0, p1 - p0)) != 0;
                if (ok) { h.t = t; h.tri = (int)(first + i); h.b0 = b0; h.b1 = b1; h.b2 = b2; }
            }
        }
        cur = st.pop(h.t);
    }
    template <bool ROUNDS> DEV void advance(const float4* nodes, const float4* tris) { if (ROUNDS) round(nodes, tris); else step(nodes, tris); }
};

// Any-hit (Scene::intersectP): no ordering, first accepted triangle ends the walk.
template <int LDSN, int OVF, int STRIDE, int NS = 8>
struct AnyWalker {
    RaySetup r;
    float tMax;
    int cur;
    bool occluded;
    uint32_t nodeSteps = 0, triSteps = 0;
    LaneStack<LDSN, OVF, STRIDE, false> st;
    static constexpr int kWords = LaneStack<LDSN, OVF, STRIDE, false>::kWords, kRows = LaneStack<LDSN, OVF, STRIDE, false>::kRows;
    DEV AnyWalker(uint32_t* column, uint32_t* spillNodes) : tMax(0.f), cur(kTraverseDone), occluded(false), st(column, spillNodes, nullptr) {}
    DEV void start(F3 o, F3 d, float tMaxRay) {
        r = setupRay(o, d);
        tMax = tMaxRay;
        occluded = false;
        st.sp = 0;
        cur = 0;
    }
    DEV bool done() const { return cur == kTraverseDone; }
    DEV void step(const float4* nodes, const float4* tris) {
        const bool atNode = cur >= 0;
        const bool atLeaf = cur < 0 && cur != kTraverseDone;
        if (__popcll(__ballot(atNode)) >= __popcll(__ballot(atLeaf))) {
            ++nodeSteps;
            if (!atNode) return;
            st.countNode();
            Node4 q = loadNode4(nodes + (size_t)cur * NS, r);
            float te;
            bool h0 = q.c0 != kTraverseDone && slabTestNF(r, tMax, q.nx.x, q.ny.x, q.nz.x, q.fx.x, q.fy.x, q.fz.x, &te);
            bool h1 = q.c1 != kTraverseDone && slabTestNF(r, tMax, q.nx.y, q.ny.y, q.nz.y, q.fx.y, q.fy.y, q.fz.y, &te);
            bool h2 = q.c2 != kTraverseDone && slabTestNF(r, tMax, q.nx.z, q.ny.z, q.nz.z, q.fx.z, q.fy.z, q.fz.z, &te);
            bool h3 = q.c3 != kTraverseDone && slabTestNF(r, tMax, q.nx.w, q.ny.w, q.nz.w, q.fx.w, q.fy.w, q.fz.w, &te);
            int next = kTraverseDone;
            if (h3) next = q.c3;
            if (h2) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c2; }
            if (h1) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c1; }
            if (h0) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c0; }
            cur = (next != kTraverseDone) ? next : st.pop(0.f);
        } else {
            ++triSteps;
            if (!atLeaf) return;
            uint32_t leaf = ~(uint32_t)cur;
            uint32_t tri = leaf >> 3, left = leaf & 7u;
            for (int rep = 0;; ++rep) { // up to kLeafTrisPerStep triangles of the leaf
                st.countTriangle();
                const float4* tp = tris + (size_t)tri * 3;
                float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                float t, b0, b1, b2;
                if (triangleTest(r, tMax, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2)) { occluded = true; break; } // Triangle::intersectP
                if (!left || rep == kLeafTrisPerStep - 1) break;
                ++tri; --left;
            }
            if (occluded) cur = kTraverseDone;
            else cur = left ? (int)~(((tri + 1u) << 3) | (left - 1u)) : st.pop(0.f);
        }
    }
    DEV void round(const float4* nodes, const float4* tris) { // see ClosestWalker::round
        while (cur >= 0) {
            st.countNode();
            Node4 q = loadNode4(nodes + (size_t)cur * NS, r);
            float te;
            bool h0 = q.c0 != kTraverseDone && slabTestNF(r, tMax, q.nx.x, q.ny.x, q.nz.x, q.fx.x, q.fy.x, q.fz.x, &te);
            bool h1 = q.c1 != kTraverseDone && slabTestNF(r, tMax, q.nx.y, q.ny.y, q.nz.y, q.fx.y, q.fy.y, q.fz.y, &te);
            bool h2 = q.c2 != kTraverseDone && slabTestNF(r, tMax, q.nx.z, q.ny.z, q.nz.z, q.fx.z, q.fy.z, q.fz.z, &te);
            bool h3 = q.c3 != kTraverseDone && slabTestNF(r, tMax, q.nx.w, q.ny.w, q.nz.w, q.fx.w, q.fy.w, q.fz.w, &te);
            int next = kTraverseDone;
            if (h3) next = q.c3;
            if (h2) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c2; }
            if (h1) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c1; }
            if (h0) { if (next != kTraverseDone) st.push(next, 0.f); next = q.c0; }
            cur = (next != kTraverseDone) ? next : st.pop(0.f);
        }
        if (cur == kTraverseDone) return;
        uint32_t leaf = ~(uint32_t)cur;
        uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
        for (uint32_t i = 0; i < cnt; ++i) {
            st.countTriangle();
            const float4* tp = tris + (size_t)(first + i) * 3;
            float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            float t, b0, b1, b2;
            if (triangleTest(r, tMax, xyz(v0), xyz(v1), xyz(v2), &t, &b0, &b1, &b2)) occluded = true;
        }
        cur = occluded ? kTraverseDone : st.pop(0.f);
    }
    template <bool ROUNDS> DEV void advance(const float4* nodes, const float4* tris) { if (ROUNDS) round(nodes, tris); else step(nodes, tris); }
};

} // namespace edev
