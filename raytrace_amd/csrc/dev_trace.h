// Ray set-up, the triangle test and the work tallies of the BVH traversal (the walk itself: dev_trace8.h).
//
// Replaces the watertight Triangle::intersects (Enigma/TriangleMesh.cpp:491-602).  Its arithmetic is the reference's,
// operation for operation, so hit/miss decisions, t and barycentrics are bit-identical; only the tree being walked
// differs from BVHAccel::intersect / intersectP (Enigma/BVH.cpp:639-722).
#pragma once
#include "dev_types.h"

namespace edev {

// Sensitivity knobs for experiment builds: N extra vector instructions without any effect in every node test
// (ENIGMA_EXP_PAD_NODE) or triangle test (ENIGMA_EXP_PAD_TRI).  How the run time follows N says whether a traversal
// kernel is bound by instruction issue (time grows with the instruction count) or by memory latency (it does not).
#ifndef ENIGMA_EXP_PAD_NODE
#define ENIGMA_EXP_PAD_NODE 0
#endif
#ifndef ENIGMA_EXP_PAD_TRI
#define ENIGMA_EXP_PAD_TRI 0
#endif
template <int N> DEV void padVectorIssue() {
    if (N > 0) {
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(x));
    }
}

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

// What the triangle test reads of a ray (RaySetup has it all; a lane that tests another lane's ray gathers just this)
struct TriRay {
    F3 o;
    bool kz0, kz1;
    float Sx, Sy, Sz;
};

// Triangle::intersects, Enigma/TriangleMesh.cpp:491-602.  tScaledOut / detOut: the two values that the test compares
// the ray's tMax with -- the only place tMax enters -- so that a caller whose tMax has shrunk since can repeat exactly
// that comparison (triangleStillInRange) instead of the whole test.
template <class RAY>
DEV bool triangleTestEx(const RAY& r, float tMaxRay, F3 p0, F3 p1, F3 p2, float* tOut, float* b0o, float* b1o, float* b2o, float* tScaledOut, float* detOut) {
    padVectorIssue<ENIGMA_EXP_PAD_TRI>();
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
    *tScaledOut = tScaled; *detOut = det;
    return true;
}
DEV bool triangleTest(const RaySetup& r, float tMaxRay, F3 p0, F3 p1, F3 p2, float* tOut, float* b0o, float* b1o, float* b2o) {
    float tScaled, det;
    return triangleTestEx(r, tMaxRay, p0, p1, p2, tOut, b0o, b1o, b2o, &tScaled, &det);
}
// The range comparison of the test above for a hit found under a larger tMax (det is never 0 for a hit)
DEV bool triangleStillInRange(float tScaled, float det, float tMaxRay) {
    return det < 0 ? !(tScaled < tMaxRay * det) : !(tScaled > tMaxRay * det);
}

struct Hit { int tri; float t, b0, b1, b2; };

// Per-launch tallies of the traversal work: for each kind of ray query (closest hit of path rays, any hit of shadow
// rays, closest hit of MIS rays) the nodes fetched and the triangles tested, summed per block into one of kVisitSlots
// slots (block b adds to slot b mod kVisitSlots; the adds of different blocks go to different words as long as a
// grid has at most kVisitSlots blocks, which the launch code keeps).  Layout: [kind][tally][slot], tally 0 = nodes,
// 1 = triangles, 2 / 3 = wave steps of either kind, 4 = pushes that found a lane's stack full (always 0: the host
// builds no tree deeper than the stacks; the top bit of a lane's triangle tally carries the flag).
constexpr uint32_t kVisitSlots = 8192;
constexpr uint32_t kStackFullFlag = 0x80000000u;
enum VisitKind : uint32_t { kVisitClosest = 0, kVisitShadow = 1, kVisitMis = 2, kVisitKinds = 3 };
DEV void blockZeroVisits(uint32_t* column, int rows, int stride) { column[rows * stride] = 0u; column[(rows + 1) * stride] = 0u; }
// every thread of the block calls it once, after its last traversal step; red: 2 * (block size / 64) words of LDS
constexpr uint32_t kVisitTallies = 5;
// where a walker without per-lane tallies reports a push that found its stack full
DEV unsigned long long* stackFullCounter(unsigned long long* visits, uint32_t kind) {
    return visits ? &visits[((size_t)kind * kVisitTallies + 4) * kVisitSlots + blockIdx.x % kVisitSlots] : nullptr;
}
// a wave's own counts of nodes fetched and triangles tested (walkers without per-lane tallies; uniform values: one lane adds them)
DEV void waveFlushVisits(unsigned long long* visits, uint32_t kind, uint32_t nodeVisits, uint32_t triVisits) {
    if (!visits || (threadIdx.x & 63u) != 0) return;
    const uint32_t slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kVisitSlots;
    atomicAdd(&visits[((size_t)kind * kVisitTallies + 0) * kVisitSlots + slot], (unsigned long long)nodeVisits);
    atomicAdd(&visits[((size_t)kind * kVisitTallies + 1) * kVisitSlots + slot], (unsigned long long)triVisits);
}
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
    const unsigned long long full = __ballot((t & kStackFullFlag) != 0);
    t &= ~kStackFullFlag;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n += __shfl_xor(n, d); t += __shfl_xor(t, d); }
    if (full != 0 && (threadIdx.x & 63u) == 0) atomicAdd(&visits[((size_t)kind * kVisitTallies + 4) * kVisitSlots + blockIdx.x % kVisitSlots], (unsigned long long)__popcll(full));
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

} // namespace edev
