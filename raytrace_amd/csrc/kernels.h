// The wavefront PathIntegrator kernels (gfx950, wave64).
//
// One render pass = N camera samples in flight (every pixel of the render region times a
// few sample numbers).  Stages per pass:
//   k_generate              SobolSampler::GetCameraSample + PerspectiveCamera ray
//                           (Enigma/Sampler.cpp:30-37,224-240; Enigma/Camera.cpp:144-192)
//   per bounce:
//     k_intersect_closest   Scene::intersect for the path rays + the head of the
//                           PathIntegrator::Li loop body (emission, termination,
//                           Enigma/PathIntegrator.cpp:67-86); bins survivors by material kind
//     k_shade<kind>         BSDF construction, UniformSampleOneLight/EstimateDirect set-up,
//                           BSDF sampling, Russian roulette
//                           (Enigma/PathIntegrator.cpp:88-175, Enigma/Integrator.cpp:71-204)
//     k_shadow              VisibilityTester::Unoccluded (Enigma/Lights.cpp:713-716)
//     k_mis                 the BSDF-sampled MIS ray of EstimateDirect (Enigma/Integrator.cpp:180-197)
//   k_film_splat            radiance scrub + FilmTile::AddSample
//                           (Enigma/Integrator.cpp:284-295, Enigma/Film.cpp:263-300)
// Queues hold path indices; they are compacted with 64-lane ballots + prefix popcounts, the
// per-wave totals scanned through LDS so a block issues one atomic per queue.
#pragma once
#include "dev_bsdf.h"
#include "dev_trace8.h"
#include "dev_sphere.h"

namespace edev {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

// ---------------------------------------------------------------- sampler
DEV uint32_t sobolBits(const DevSampler& S, uint32_t relx, uint32_t rely, uint32_t slot, uint32_t dim) {
    return S.tx[(size_t)dim * S.res + relx] ^ S.ty[(size_t)dim * S.res + rely] ^ S.sampleConst[(size_t)slot * S.nDims + dim];
}
// HaltonSampler::GetIndexForSample + SampleDimension (Enigma/Sampler.cpp:336-367) with RadicalInverseSpecialized /
// ScrambledRadicalInverseSpecialized (Enigma/SamplerInclude.h:49-84), the base a run-time value instead of one of the
// reference's 1000 template instances.  The index of (pixel, sample number) is below 2^32 (checked when the sampler is
// configured), so the digits come off in 32-bit arithmetic; the reversed digits need 64 bits as in the reference.
// DevSampler::bitConst holds, for this sampler: {stride, exponent of 2, scale of 3, sample-at-centre flag}, then per
// dimension a record of six words (halton_host.h), then the permutations, one word per digit value.  A digit comes off
// with a multiplication (the record's magic and shift: exact for every 32-bit numerator), 1 / base and the tail term
// invBase * perm[0] / (1 - invBase) are the host's, rounded as the reference rounds them.
// Kernels that draw samples are compiled once per sampler (template argument HAL): with the choice made at run time
// the shade kernels of the Sobol' sampler, which never run this, were 3.5 % slower, inlined or not.  Inlined into the
// Halton builds: as one function per module, called from the dozen places where a shade kernel draws samples, those
// kernels were 4 % slower.
#ifdef ENIGMA_EXP_HALTON_NOINLINE
#define ENIGMA_HALTON_LINKAGE __noinline__
#else
#define ENIGMA_HALTON_LINKAGE __forceinline__
#endif
static __device__ ENIGMA_HALTON_LINKAGE float haltonValue(const uint32_t* H, uint32_t off, uint32_t sampleNum, uint32_t dim) {
    if (dim < 2u && H[3] != 0u) return 0.5f;
    const uint32_t stride = H[0];
    if (off >= stride) off -= stride;                // the sum of two terms, each already reduced modulo the stride
    uint32_t a = off + sampleNum * stride;
    if (dim == 0u) return (float)__brev(a >> H[1]) * 0x1p-32f; // ReverseBits64(a) * 2^-64, one rounding to float in both forms
    // the dimension's record: base, permutation offset, magic and shift of the division by the base, 1 / base, the tail term
    const uint32_t* rec = H + 4u + 6u * dim;
    const uint32_t base = rec[0], magic = rec[2], shift = rec[3];
    const float invBase = __uint_as_float(rec[4]);
    unsigned long long reversedDigits = 0;
    float invBaseN = 1.f;
    if (dim == 1u) {
        a /= H[2];
        while (a) {
            const uint32_t t = __umulhi(magic, a), next = (t + ((a - t) >> 1)) >> shift, digit = a - next * base;
            reversedDigits = reversedDigits * base + digit;
            invBaseN *= invBase;
            a = next;
        }
        return smin((float)reversedDigits * invBaseN, kOneMinusEpsilon);
    }
    const uint32_t* perm = H + rec[1];
    while (a) {
        const uint32_t t = __umulhi(magic, a), next = (t + ((a - t) >> 1)) >> shift, digit = a - next * base;
        reversedDigits = reversedDigits * base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    return smin(invBaseN * ((float)reversedDigits + __uint_as_float(rec[5])), kOneMinusEpsilon);
}
DEV float haltonDimension(const DevSampler& S, uint32_t relx, uint32_t rely, uint32_t sampleNum, uint32_t dim) {
    return haltonValue(S.bitConst, S.tx[relx] + S.ty[rely], sampleNum, dim);
}

// SobolSampleFloat + SobolSampler::SampleDimension (Enigma/SamplerInclude.h:2627-2637, Enigma/Sampler.cpp:229-240)
template <bool HAL>
DEV float sampleDimension(const DevSampler& S, uint32_t relx, uint32_t rely, uint32_t slot, uint32_t dim) {
    // relx, rely: pixel relative to the sample-bounds origin (m_currentPixel - m_sampleBounds.m_min)
    if constexpr (HAL) return haltonDimension(S, relx, rely, S.sampleConst[slot], dim); // HaltonSampler: one sample number per slot
    uint32_t v = sobolBits(S, relx, rely, slot, dim);
    float s = smin((float)v * 0x1p-32f, kOneMinusEpsilon);
    if (dim == 0 || dim == 1) {
        s = s * S.resolutionF + (float)(dim == 0 ? S.sbx0 : S.sby0);
        s = clampf(s - (float)(dim == 0 ? (int)relx + S.sbx0 : (int)rely + S.sby0), 0.f, kOneMinusEpsilon);
    }
    return s;
}

// Sample of an arbitrary sample number of the pixel (the entries of GlobalSampler's sample arrays,
// Enigma/Sampler.cpp:170-180, are samples of OTHER sample numbers of the same pixel): the per-sample
// constant is GF(2)-linear in the bits of the sample number, so it is XORed together from per-bit constants.
template <bool HAL>
DEV float sampleDimensionOf(const DevSampler& S, uint32_t relx, uint32_t rely, unsigned long long sampleNum, uint32_t dim) {
    if constexpr (HAL) return haltonDimension(S, relx, rely, (uint32_t)sampleNum, dim);
    uint32_t v = S.tx[(size_t)dim * S.res + relx] ^ S.ty[(size_t)dim * S.res + rely];
    for (int b = 0; sampleNum != 0 && b < kSampleBits; ++b, sampleNum >>= 1)
        if (sampleNum & 1ull) v ^= S.bitConst[(size_t)dim * kSampleBits + b];
    return smin((float)v * 0x1p-32f, kOneMinusEpsilon); // dims >= 2 only
}

// Pixels of the render region are numbered tile by tile: 8 x 8 tiles in row-major order (64 consecutive numbers =
// one wave = one compact bundle of camera rays), then the pixels right of the last full tile column and below the
// last full tile row in scanline order.  Which number a pixel gets changes nothing in the image; it decides which
// rays travel through the kernels side by side.
DEV void pixelOfIndex(const DevIntegrator& I, uint32_t pix, uint32_t* xo, uint32_t* yo) {
    const uint32_t rw = (uint32_t)I.rw, rh = (uint32_t)I.rh;
    const uint32_t tX = rw >> 3, tY = rh >> 3, tiled = tX * tY * 64u;
    if (pix < tiled) {
        uint32_t tile = pix >> 6, in = pix & 63u;
        uint32_t ty = tile / tX, tx = tile - ty * tX;
        *xo = tx * 8u + (in & 7u); *yo = ty * 8u + (in >> 3);
        return;
    }
    uint32_t r = pix - tiled;
    const uint32_t stripW = rw - tX * 8u, stripH = tY * 8u; // right strip: stripW x stripH
    if (r < stripW * stripH) { uint32_t y = r / stripW; *xo = tX * 8u + (r - y * stripW); *yo = y; return; }
    r -= stripW * stripH;                                    // bottom strip: rw x (rh - stripH)
    uint32_t y = r / rw;
    *xo = r - y * rw; *yo = stripH + y;
}

// Pixel (relative to the sample-bounds origin) and sample slot of a path: paths of a pass are numbered
// slot-major over the pixel numbers of the render region.
DEV void pathPixel(const DevIntegrator& I, const DevSampler& S, uint32_t p, uint32_t* relx, uint32_t* rely, uint32_t* slot) {
    uint32_t s = p / I.nPixels, pix = p - s * I.nPixels;
    uint32_t x, y;
    pixelOfIndex(I, pix, &x, &y);
    *relx = (uint32_t)(I.rx0 - S.sbx0) + x;
    *rely = (uint32_t)(I.ry0 - S.sby0) + y;
    *slot = I.slotBase + s;
}

// ---------------------------------------------------------------- camera
DEV F3 xfPoint(const float* m, F3 p) { // Transform::transformPoint, Enigma/Transform.h:36-45
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.f) return F3(xp, yp, zp);
    return F3(xp, yp, zp) / wp;
}
DEV F3 xfVector(const float* m, F3 v) { // Transform::transformVector, Enigma/Transform.h:47-52
    return F3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// EnvironmentCamera::GenerateRay's direction in camera space (Enigma/Camera.cpp:372-377, Enigma/MathCommon.cpp:13-28)
DEV F3 environmentDirection(const DevCamera& C, float fx, float fy) {
    const float theta = kPi * fy / C.resY;
    const float phi = (kPi * 2.0f) * fx / C.resX;
    const float sinTheta = sinf(theta);
    return F3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosf(theta));
}
DEV void cameraRay(const DevCamera& C, float fx, float fy, float lu, float lv, F3* o, F3* d) {
    if (C.type == 1u) { // ENIGMA_CAMERA_ENVIRONMENT
        *d = xfVector(C.c2w, environmentDirection(C, fx, fy));
        *o = xfPoint(C.c2w, F3(0.f, 0.f, 0.f));
        return;
    }
    F3 Pcamera = xfPoint(C.r2c, F3(fx, fy, 0.f));
    F3 ro(0.f, 0.f, 0.f), rd = normalize(Pcamera);
    if (C.lensRadius > 0.f) {
        float lx, ly;
        concentricSampleDisk(lu, lv, &lx, &ly);
        lx *= C.lensRadius; ly *= C.lensRadius;
        float ft = C.focalDistance / rd.z;
        F3 Pfocus = ro + rd * ft;
        ro = F3(lx, ly, 0.f);
        rd = normalize(Pfocus - ro);
    }
    *d = xfVector(C.c2w, rd);
    *o = xfPoint(C.c2w, ro);
}

// ---------------------------------------------------------------- queue compaction
// Block-wide append: every thread of the block must call this (uniform control flow).
// Returns the queue slot for lanes with pred (undefined otherwise).
DEV uint32_t blockReserve(bool pred, uint32_t* counter, uint32_t* lds /* kWavesPerBlock + 1 words */) {
    unsigned long long mask = __ballot(pred);
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wid] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) { uint32_t c = lds[w]; lds[w] = total; total += c; }
        lds[kWavesPerBlock] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t slot = lds[kWavesPerBlock] + lds[wid] + prefix;
    __syncthreads(); // lds is reused by the next reservation
    return slot;
}
// Three block-wide appends (shadow, MIS, next-bounce queues) behind one pair of barriers.  lds holds
// 2 x 16 words; callers alternate the half per loop iteration so no third barrier is needed.
DEV void blockReserve3(bool a, bool b, bool c, uint32_t* ca, uint32_t* cb, uint32_t* cc, uint32_t* lds, uint32_t* sa, uint32_t* sb, uint32_t* sc) {
    unsigned long long ma = __ballot(a), mb = __ballot(b), mc = __ballot(c);
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { lds[wid] = __popcll(ma); lds[4 + wid] = __popcll(mb); lds[8 + wid] = __popcll(mc); }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t total = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) total += lds[threadIdx.x * 4 + w];
        uint32_t* ctr = threadIdx.x == 0 ? ca : (threadIdx.x == 1 ? cb : cc);
        lds[12 + threadIdx.x] = total ? atomicAdd(ctr, total) : 0u;
    }
    __syncthreads();
    uint32_t oa = lds[12], ob = lds[13], oc = lds[14];
    for (uint32_t w = 0; w < wid; ++w) { oa += lds[w]; ob += lds[4 + w]; oc += lds[8 + w]; }
    *sa = oa + __popcll(ma & below); *sb = ob + __popcll(mb & below); *sc = oc + __popcll(mc & below);
}
static_assert(kWavesPerBlock == 4, "blockReserve3 lays its counts out for four waves");

// ---------------------------------------------------------------- LDS-resident scene
// Scenes whose BVH nodes + triangles fit a few tens of KB (the Cornell-box configurations) are copied
// into LDS by every traversal block before it starts: node and triangle fetches become ds_read_b128
// instead of 64 divergent 16-byte requests per wave-instruction through the texture-address path,
// which is what bounds traversal of a cache-resident scene.  Layout: [nodes: 5 float4 each][tris: 3 float4 each]
// (80-byte nodes start 20 banks apart: lanes at neighbouring nodes read from different banks).
DEV void stageSceneToLds(const DevScene& sc, float4* lds) {
    const uint32_t n = sc.nNodes * kLdsNodePitch + sc.nTris * 3u;
    const uint32_t nN = sc.nNodes * kLdsNodePitch;
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) lds[i] = i < nN ? sc.nodes[i] : sc.tris[i - nN];
    __syncthreads();
}

// ---------------------------------------------------------------- geometry at a hit
// second triangle word: material index [0:24) | material kind [24:27) | per-vertex normals [27] | per-vertex uvs [28] |
// reverseOrientation [29] | flip-normal [31]
DEV uint32_t triMaterial(uint32_t mw) { return mw & 0xffffffu; }
DEV uint32_t triKind(uint32_t mw) { return (mw >> 24) & 7u; }
DEV bool triFlip(uint32_t mw) { return (mw >> 31) != 0; }

DEV bool triHasNormals(uint32_t mw) { return (mw >> 27) & 1u; }
DEV bool triHasUvs(uint32_t mw) { return (mw >> 28) & 1u; }
DEV bool triReverse(uint32_t mw) { return (mw >> 29) & 1u; }

// n: geometric normal (turned towards the shading normal when the mesh has normals); ns, dpdu: shading normal and shading dpdu
// wo: Interaction::m_wo.  normalize(-d) for a triangle; a sphere's comes back from object space un-normalised.
struct HitGeom { F3 p, pError, n, ns, dpdu, wo; int matIdx, lightIdx; };

DEV F3 triNormal(F3 p0, F3 p1, F3 p2, bool flip) { // Enigma/TriangleMesh.cpp:287-289
    F3 n = normalize(cross(p0 - p2, p1 - p2));
    return flip ? -n : n;
}
// Triangle::intersect after the hit test, Enigma/TriangleMesh.cpp:226-371 (no alpha mask, no per-vertex tangents)
DEV HitGeom hitGeometry(const DevScene& sc, const float4* tris, int tri, float b0, float b1, float b2, F3 rayDir, uint32_t path) {
    if (sc.sphereGeom && tri >= (int)sc.nTris) { // a sphere hit: k_sphere_surface has left the interaction in the path's slot
        const float4* sg = sc.sphereGeom + (size_t)path * 4;
        float4 a = sg[0], b = sg[1], c = sg[2], d = sg[3];
        HitGeom g;
        g.p = xyz(a); g.n = xyz(b); g.dpdu = xyz(c); g.wo = xyz(d);
        g.pError = F3(a.w, b.w, c.w);
        g.ns = g.n;
        g.matIdx = __float_as_int(d.w);
        g.lightIdx = -1;
        return g;
    }
    const float4* tp = tris + (size_t)tri * 3;
    float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
    F3 p0 = xyz(v0), p1 = xyz(v1), p2 = xyz(v2);
    uint32_t mw = __float_as_uint(v1.w);
    HitGeom g;
    g.matIdx = (int)triMaterial(mw);
    g.lightIdx = __float_as_int(v2.w);
    F3 dp02 = p0 - p2, dp12 = p1 - p2;
    F3 dpdu = dp12 - dp02; // the default uvs (0,0), (1,0), (1,1): determinant 1
    F3 dpdv = -dp12;
    bool degenerateUV = false;
    const bool shaded = sc.triShade && (triHasNormals(mw) || triHasUvs(mw));
    float4 s0, s1, s2;
    if (shaded) {
        const float4* sp = sc.triShade + (size_t)tri * 4;
        s0 = sp[0]; s1 = sp[1]; s2 = sp[2];
        if (triHasUvs(mw)) {
            float4 vv = sp[3];
            float du02 = s0.w - s2.w, dv02 = vv.x - vv.z, du12 = s1.w - s2.w, dv12 = vv.y - vv.z;
            float determinant = du02 * dv12 - dv02 * du12;
            degenerateUV = fabsf(determinant) < 1e-8;
            if (!degenerateUV) {
                float invdet = 1 / determinant;
                dpdu = (dv12 * dp02 - dv02 * dp12) * invdet;
                dpdv = (-du12 * dp02 + du02 * dp12) * invdet;
            }
        }
    }
    if (degenerateUV || lengthSqr(cross(dpdu, dpdv)) == 0) {
        F3 ng = cross(p2 - p0, p1 - p0);
        coordinateSystem(normalize(ng), &dpdu, &dpdv);
    }
    g.dpdu = dpdu;
    float xAbsSum = (fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x));
    float yAbsSum = (fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y));
    float zAbsSum = (fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z));
    g.pError = gammaf(7) * F3(xAbsSum, yAbsSum, zAbsSum);
    g.p = b0 * p0 + b1 * p1 + b2 * p2;
    g.n = triNormal(p0, p1, p2, triFlip(mw));
    g.ns = g.n;
    g.wo = normalize(-rayDir); // Interaction.cpp:26
    if (shaded && triHasNormals(mw)) { // shading geometry, Enigma/TriangleMesh.cpp:291-370 and SetShadingGeometry, Enigma/Interaction.cpp:132-150
        F3 ns = b0 * xyz(s0) + b1 * xyz(s1) + b2 * xyz(s2);
        ns = lengthSqr(ns) > 0 ? normalize(ns) : g.n;
        F3 ss = normalize(dpdu);
        F3 ts = cross(ss, ns);
        if (lengthSqr(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinateSystem(ns, &ss, &ts);
        if (triReverse(mw)) ts = -ts;
        g.ns = normalize(cross(ss, ts));
        if (dot(g.n, g.ns) < 0.f) g.n = -g.n;
        g.dpdu = ss;
    }
    return g;
}

// The surface interactions of this bounce's sphere hits, worked out before the shade kernels run so that those only
// load them: with the sphere code inlined the shade kernels of every scene would carry its registers.
__global__ void __launch_bounds__(kBlock) k_sphere_surface(DevScene sc, DevPaths P, const uint32_t* matQueue, int kind, DevCounters* ctr) {
    const uint32_t count = ctr->matCount[kind].v;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
        const uint32_t p = matQueue[i];
        const float4 hv = P.hit[p];
        const int tri = __float_as_int(hv.x);
        if (tri < (int)sc.nTris) continue;
        const DevSphere& sp = sc.spheres[tri - (int)sc.nTris];
        F3 pos, pErr, n, dpdu, wo;
        sphereSurface(sp, F3(hv.y, hv.z, hv.w), xyz(P.rayD[p]), &pos, &pErr, &n, &dpdu, &wo);
        float4* sg = sc.sphereGeom + (size_t)p * 4;
        sg[0] = mk4(pos, pErr.x); sg[1] = mk4(n, pErr.y); sg[2] = mk4(dpdu, pErr.z); sg[3] = mk4(wo, __int_as_float((int)triMaterial(sp.mw)));
    }
}

// The traversal kernels exist in two builds (template flag SPH): scenes without spheres run code that knows nothing of
// them -- with the sphere test inlined, its registers cost the BVH walk a third of its speed.  A sphere's primitive
// number is sc.nTris + its index (sc.nTris counts the slots of the triangle array, the spheres' own included).
template <bool SPH>
DEV bool isSpherePrim(const DevScene& sc, int prim) { return SPH && prim >= (int)sc.nTris; }
// the second triangle word of a primitive of either kind
template <bool SPH>
DEV uint32_t primWord(const DevScene& sc, const float4* tris, int prim) {
    return isSpherePrim<SPH>(sc, prim) ? sc.spheres[prim - (int)sc.nTris].mw : __float_as_uint(tris[(size_t)prim * 3 + 1].w);
}

// ---------------------------------------------------------------- kernels
template <bool HAL = false>
__global__ void __launch_bounds__(kBlock) k_generate(DevPaths P, DevSampler S, DevCamera C, DevIntegrator I, uint32_t nPaths, uint32_t nPixels,
                                                     uint32_t slotBase, DevCounters* ctr, uint32_t firstDim, DevDirect D) {
    __shared__ float4 recs[kBlock * 4];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < nPaths;
    F3 o(0.f), d(0.f);
    if (active) {
        uint32_t slot = slotBase + i / nPixels, pix = i % nPixels;
        uint32_t tx, ty;
        pixelOfIndex(I, pix, &tx, &ty);
        int px = I.rx0 + (int)tx, py = I.ry0 + (int)ty;
        uint32_t relx = (uint32_t)(px - S.sbx0), rely = (uint32_t)(py - S.sby0);
        // GetCameraSample: image = pixel + Get2D (dims 0,1), time = Get1D (dim 2), lens = Get2D (dims 3,4)
        float fx = (float)px + sampleDimension<HAL>(S, relx, rely, slot, 0);
        float fy = (float)py + sampleDimension<HAL>(S, relx, rely, slot, 1);
        float lu = 0.f, lv = 0.f;
        if (C.lensRadius > 0.f) { lu = sampleDimension<HAL>(S, relx, rely, slot, 3); lv = sampleDimension<HAL>(S, relx, rely, slot, 4); }
        cameraRay(C, fx, fy, lu, lv, &o, &d);
        P.L[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        P.pfilm[i] = make_float2(fx, fy);
        if (D.arrayOff) { D.arrayOff[i] = 0; D.pendCount[i] = 0; }
        if (i == 0) atomicAdd(&ctr->samples, (unsigned long long)nPaths);
    }
    // the 64-byte records of the block's paths go out through LDS, so that every store instruction of a wave
    // covers 1 KB of consecutive bytes (a lane writing its own record would touch 64 separate lines four times)
    float4* mine = recs + threadIdx.x * 4;
    mine[0] = mk4(o, 0.f);
    mine[1] = mk4(d, __uint_as_float(packState(0u, firstDim, 0u, 0u))); // 5 after GetCameraSample; DirectLightingIntegrator skips its sample arrays
    mine[2] = make_float4(1.f, 1.f, 1.f, 1.f);
    mine[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const size_t blockBase = (size_t)blockIdx.x * kBlock * 4;
    const size_t limit = (size_t)nPaths * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        size_t e = blockBase + (size_t)k * kBlock + threadIdx.x;
        if (e < limit) P.rayO.base[e] = recs[k * kBlock + threadIdx.x];
    }
}

// Scene::intersect for queue[0..count) (queue == nullptr: identity) + the head of the Li loop.
struct MatQueues { uint32_t* q[kQueueKinds]; };

// The closest-hit kernel keeps the chunk as its unit although the shadow kernel of scenes in HBM streams (k_shadow_stream):
// a streaming build of this kernel -- waves reserving runs of 512 rays, finished paths handed to the material queues either
// from a list of the wave (queues lose their pixel order: shading and the next bounce 10-19 % slower) or run by run in
// position order (more state, 31 instead of 17 spilled registers: the kernel itself 17 % slower) -- lost either way on
// the 1.18 M-triangle scene (5.9 -> 7.4 / 8.4 ms per launch), while coherent camera rays gained nothing from it.
// Chunks go to blocks in plain order (chunk c -> block c mod gridDim), i.e. round-robin over the eight XCDs and their L2s.
// Measured alternatives on the 1.18 M-triangle scene: runs of 8 consecutive chunks per XCD inside groups of 64 blocks:
// no change (68.9 vs 68.7 ms per step); runs of 32: 1 % slower; one contiguous eighth of every sweep per XCD: 40 %
// slower (96 ms: the eighths of an image differ in cost, and an XCD that draws the expensive part finishes last).
// What misses an L2 is caught by the 256 MB of Infinity Cache, which all XCDs share.
// A block owns chunks of kTraceChunk consecutive queue positions (chunk c -> block c mod gridDim) and
// treats a chunk as a pool of rays: a wave whose lanes have finished their rays (kRefillIdle of them,
// or all) writes those rays' results and hands the idle lanes the next rays of the pool, so lanes do
// not sit out the slowest ray of their wave.  The pool cursor is an LDS atomic; nothing global is
// bumped per wave.  After the chunk the block bins the surviving paths by material kind with ONE
// tail bump per kind and block (per-wave ballots -> LDS -> block totals).
// 2048 rays = eight per lane: a chunk ends with a tail in which lanes run dry one after the other, and the longer
// the chunk the smaller the tail's share (BigMesh: 145 ms per step at 512, 129 at 1024, 122 at 2048); at 4096 the
// per-chunk table no longer leaves room for six blocks per CU and the gain is lost again.
constexpr uint32_t kTraceChunk = 2048;
constexpr int kTraceItems = kTraceChunk / kBlock;
static_assert(kTraceItems >= 1 && kTraceItems <= 16, "the chunk's material kinds are packed four bits per item into 64 bits");

constexpr uint32_t kNoRay = 0xffffffffu;
// medium boundaries a VolPathIntegrator ray crosses inside the closest-hit kernel before the path takes the boundary queue (0 = never)
#ifndef ENIGMA_INLINE_CROSSINGS
#define ENIGMA_INLINE_CROSSINGS 16
#endif
constexpr uint32_t kInlineCrossings = ENIGMA_INLINE_CROSSINGS;
// Kernels of scenes staged into LDS come in two builds: trees at most kLdsShallowLevels deep (the Cornell boxes: 3) keep
// that many stack levels per lane in LDS, which leaves room for six blocks per CU instead of five (80 registers, nothing
// spilled that matters: closest-hit kernel 0.634 -> 0.605 ms, shadow kernel 0.260 -> 0.242 ms per Cornell launch; seven
// blocks at 72 registers lose it again); deeper trees of small scenes keep kLdsSceneLevels and five blocks.
constexpr int kLdsSceneLevels = 12, kLdsShallowLevels = 6;
constexpr int ldsSceneBlocks(int ldsn) { return ldsn <= kLdsShallowLevels ? 6 : 5; }

// lanes in idleMask take consecutive pool positions; returns this lane's position (valid for idle lanes)
DEV uint32_t poolTake(uint32_t* poolNext, unsigned long long idleMask, uint32_t* baseOut) {
    uint32_t lane = threadIdx.x & 63u;
    int leader = __ffsll((long long)idleMask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(poolNext, (uint32_t)__popcll(idleMask));
    base = __shfl(base, leader);
    *baseOut = base;
    return base + __popcll(idleMask & ((1ull << lane) - 1ull));
}

// medium + 1 on the side of triangle `tri` that w points into, for a ray that arrived in medium rayMed (+1)
DEV uint32_t mediumBeyond(const DevScene& sc, int tri, uint32_t rayMed, F3 n, F3 w) {
    uint32_t in = rayMed, out = rayMed;
    if (sc.triMedium) {
        uint32_t tm = sc.triMedium[tri], a = tm & 0xffffu, b = tm >> 16;
        if (a != b) { in = a; out = b; }
    }
    return dot(w, n) > 0 ? out : in;
}
// Position, error bounds and normal of a hit on a surface without a material, for the rays that are not the path's own
// (their sphere hits have no record from k_sphere_surface: worked out here).
template <bool SPH>
DEV void boundaryGeometry(const DevScene& sc, const float4* tris, const Hit& h, F3 d, uint32_t path, F3* gp, F3* ge, F3* gn) {
    if (isSpherePrim<SPH>(sc, h.tri)) {
        F3 dpdu, wo;
        sphereSurface(sc.spheres[h.tri - (int)sc.nTris], F3(h.b0, h.b1, h.b2), d, gp, ge, gn, &dpdu, &wo);
    } else {
        HitGeom g = hitGeometry(sc, tris, h.tri, h.b0, h.b1, h.b2, d, path);
        *gp = g.p; *ge = g.pError; *gn = g.n;
    }
}

// The head of the Li loop for a ray that has finished its walk (Enigma/PathIntegrator.cpp:62-87; MODE 1: the
// DirectLightingIntegrator's, MODE 2: the VolPathIntegrator's with its medium sampling): emitted radiance, the depth
// test, the hit record.  Returns the queue the path goes to + 1 (its material kind, the boundary / resume queue, the
// medium queue), 0 when the path has ended.
template <int MODE, bool SPH, bool HAL, class Walker>
DEV uint32_t finishPathRay(const DevScene& sc, const DevPaths& P, const DevIntegrator& I, const DevSampler& S, const float4* tris, const Walker& w,
                           uint32_t p, uint32_t& state, const uint32_t* pendCount) {
    constexpr bool DL = MODE == 1, VOL = MODE == 2;
    uint32_t kind = 0;
    Hit h = w.h;
    if (sc.hitT) sc.hitT[p] = h.tri >= 0 ? h.t : EDEV_INF; // the primary-hit tap reads the walk's own distance; no render sets the pointer
    bool proceed = true; // VolPathIntegrator: false once the medium has ended the path or produced a medium interaction
    if (VOL && stMedium(state) != 0 && sc.media[stMedium(state) - 1u].opaque != 0.f) {
        // GridDensityMedium::Sample returns black and draws nothing (Enigma/Medium.cpp:56-59): the throughput is black, the loop ends
        float4 be = P.beta[p];
        P.beta[p] = make_float4(0.f, 0.f, 0.f, be.w);
        proceed = false;
    } else if (VOL && stMedium(state) != 0) {
        const DevMedium md = sc.media[stMedium(state) - 1u];
        uint32_t px, py, slot;
        pathPixel(I, S, p, &px, &py, &slot);
        uint32_t dim = stDim(state);
        float uc = sampleDimension<HAL>(S, px, py, slot, dim), ud = sampleDimension<HAL>(S, px, py, slot, dim + 1);
        state = packState(stMedium(state), dim + 2, stBounces(state), stSpec(state)) | (state & kStNoDifferentials);
        reinterpret_cast<float*>(&P.rayD[p])[3] = __uint_as_float(state);
        int channel = min((int)(uc * 3), 3 - 1);
        float dist = -logf(1 - ud) / md.sigmaT[channel];
        float len = length(w.r.d);
        float tHit = h.tri >= 0 ? h.t : EDEV_INF;
        float t = smin(dist / len, tHit);
        bool sampledMedium = t < tHit;
        float tt = smin(t, EDEV_MAXF);
        F3 Tr(expf(-md.sigmaT[0] * tt * len), expf(-md.sigmaT[1] * tt * len), expf(-md.sigmaT[2] * tt * len));
        F3 sigT(md.sigmaT[0], md.sigmaT[1], md.sigmaT[2]);
        F3 density = sampledMedium ? sigT * Tr : Tr;
        float pdf = 0;
        pdf += density.x; pdf += density.y; pdf += density.z;
        pdf *= 1 / (float)3;
        if (pdf == 0) pdf = 1;
        F3 wgt = sampledMedium ? Tr * F3(md.sigmaS[0], md.sigmaS[1], md.sigmaS[2]) / pdf : Tr / pdf;
        float4 be = P.beta[p];
        F3 beta = xyz(be) * wgt;
        P.beta[p] = mk4(beta, be.w);
        if (isBlack(beta)) proceed = false;
        else if (sampledMedium) {
            proceed = false;
            if ((int)stBounces(state) < I.maxDepth) {
                kind = (uint32_t)kQueueMedium + 1u;
                P.hit[p] = make_float4(__int_as_float(-1), t, 0.f, 0.f);
            }
        }
    } else if (VOL) {
        // the loop's test of the throughput (Enigma/VolPathIntegrator.cpp:71) sits outside its medium branch: a path whose
        // colour filters have multiplied to black (a red reflection, then a blue transmission) ends here, after the walk
        if (isBlack(xyz(P.beta[p]))) proceed = false;
    }
    if (proceed && h.tri >= 0) {
        uint32_t st = state;
        uint32_t bounces = stBounces(st);
        const bool sphere = isSpherePrim<SPH>(sc, h.tri); // nothing to fetch from the triangle array
        const float4* tp = tris + (size_t)(sphere ? 0 : h.tri) * 3;
        float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
        int lightIdx = sphere ? (int)sc.spheres[h.tri - (int)sc.nTris].light - 1 : __float_as_int(v2.w);
        uint32_t mw = sphere ? sc.spheres[h.tri - (int)sc.nTris].mw : __float_as_uint(v1.w);
        if ((DL || bounces == 0 || stSpec(st)) && lightIdx >= 0) { // L += beta * isect.Le(-ray.d)
            const DevLight& lt = sc.lights[lightIdx];
            F3 n;
            if (SPH && sphere) {
                F3 ps, pe, dpdu, wo;
                sphereSurface(sc.spheres[h.tri - (int)sc.nTris], F3(h.b0, h.b1, h.b2), w.r.d, &ps, &pe, &n, &dpdu, &wo);
            } else n = triNormal(xyz(v0), xyz(v1), xyz(v2), triFlip(mw));
            if (lt.twoSided || dot(n, -w.r.d) > 0) {
                float4 L = P.L[p], be = P.beta[p];
                F3 add = xyz(be) * F3(lt.L[0], lt.L[1], lt.L[2]);
                P.L[p] = make_float4(L.x + add.x, L.y + add.y, L.z + add.z, 0.f);
            }
        }
        if (DL || (int)bounces < I.maxDepth) {
            kind = triKind(mw) + 1u;
            if (DL && triKind(mw) == MK_NONE) kind = (uint32_t)kQueueMedium + 1u; // its queue MK_COUNT holds the paths that resume a branch
            P.hit[p] = make_float4(__int_as_float(h.tri), h.b0, h.b1, h.b2);
        }
    } else {
        if (proceed && sc.nInfinite > 0 && (DL || stBounces(state) == 0 || stSpec(state))) { // L += beta * light->Le(ray) over the infinite lights
            float4 L = P.L[p];
            F3 be = xyz(P.beta[p]), acc = xyz(L);
            for (int k = 0; k < sc.nInfinite; ++k) acc += be * infiniteLe(sc, sc.lights[sc.infiniteLights[k]], w.r.d);
            P.L[p] = mk4(acc, 0.f);
        }
        if (DL && pendCount[p] != 0) kind = MK_COUNT + 1u;
    }
    return kind;
}

// DL: the head of DirectLightingIntegrator::Li instead (Enigma/DirectLightingIntegrator.cpp:41-55): emitted
// radiance is added at every hit, every hit is shaded, and a path that misses resumes a pending branch.
// MODE 2: the head of VolPathIntegrator::Li (Enigma/VolPathIntegrator.cpp:63-111): a ray that travels in a medium
// samples it first (HomogeneousMedium::Sample, Enigma/Medium.cpp:83-108); a sampled medium interaction goes to the
// medium queue, otherwise the surface is handled as in the PathIntegrator, material-less surfaces in a queue of their own.
template <int LDSN, int OVF, bool LDS_SCENE, int MODE, bool SPH = false, bool HAL = false>
__global__ void __launch_bounds__(kBlock, LDS_SCENE ? ldsSceneBlocks(LDSN) : 6) k_intersect_closest(DevScene sc, DevPaths P, DevIntegrator I, DevSampler S, const uint32_t* queue,
                                                              const uint32_t* countPtr, uint32_t countImm, MatQueues mq, DevCounters* ctr, uint32_t kRefillIdle,
                                                              const uint32_t* pendCount, uint32_t queueMask) {
    constexpr bool DL = MODE == 1, VOL = MODE == 2;
    constexpr bool kCrossInline = VOL && LDS_SCENE && kInlineCrossings > 0;
    // queue MK_COUNT: surfaces without a material / (MODE 1) paths that resume a branch; queue MK_COUNT + 1: medium interactions /
    // (MODE 1) surfaces without a material
    constexpr int NK = MODE == 0 ? MK_COUNT + 1 : kQueueKinds;
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, LDS_SCENE, SPH>::Closest;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    __shared__ uint32_t binCnt[kQueueKinds * kWavesPerBlock];
    __shared__ uint32_t binBase[kQueueKinds];
    __shared__ uint32_t poolNext;
    __shared__ uint8_t kindOf[kTraceChunk]; // material kind + 1 of the path at a pool position, 0 = path ended
    extern __shared__ float4 sceneLds[];
    uint32_t spill[spillWords(OVF)];
    const uint32_t count = countPtr ? *countPtr : countImm;
    if (blockIdx.x * kTraceChunk >= count) return;
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    if (LDS_SCENE) { stageSceneToLds(sc, sceneLds); nodes = sceneLds; tris = sceneLds + sc.nNodes * kLdsNodePitch; }
    blockZeroVisits(stack + threadIdx.x, Walker::kRows, kBlock);
    uint32_t nodeSteps = 0, triSteps = 0, inlineRays = 0;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    for (uint32_t chunk = blockIdx.x * kTraceChunk; chunk < count; chunk += gridDim.x * kTraceChunk) {
        const uint32_t chunkCount = min(kTraceChunk, count - chunk);
        if (threadIdx.x == 0) poolNext = 0;
        __syncthreads();
        Walker w(stack + threadIdx.x, spill, SphereRefs(sc));
        uint32_t pos = kNoRay, p = 0, state = 0, crossings = 0;
        bool exhausted = false;
        for (;;) {
            bool idle = w.done();
            unsigned long long idleMask = __ballot(idle);
            uint32_t nIdle = __popcll(idleMask);
            if (nIdle == 64u || (!exhausted && nIdle >= kRefillIdle)) {
                if (idle && pos != kNoRay) { // the head of the Li loop for a finished ray (Enigma/PathIntegrator.cpp:62-87)
                    const uint32_t kind = finishPathRay<MODE, SPH, HAL>(sc, P, I, S, tris, w, p, state, pendCount);
                    // VolPathIntegrator: a ray that has reached a surface without a material -- a medium boundary -- and was not
                    // stopped by the medium before goes on from the far side, in the medium there, without a bounce counted or a
                    // sample drawn (Enigma/VolPathIntegrator.cpp:105-111).  The lane does that here and keeps its ray instead of
                    // sending the path through the boundary queue and the next launch: same arithmetic as k_interface, which
                    // still serves the paths that cross more than kInlineCrossings boundaries in a row and the primary-hit tap.
                    // For scenes in LDS only (Cornell in fog: 16 -> 6 launch sets per pass, 26.1 -> 22.7 ms per step); the build for
                    // scenes in global memory pays for the crossing code with twenty more spilled registers in every step of
                    // every ray (the 1.18 M-triangle scene in fog: 126.9 -> 131.0 ms) and keeps the queue.
                    if (kCrossInline && kind == (uint32_t)MK_NONE + 1u && sc.hitT == nullptr && crossings < kInlineCrossings) {
                        F3 gp, ge, gn;
                        const F3 d = w.r.d;
                        boundaryGeometry<SPH>(sc, tris, w.h, d, p, &gp, &ge, &gn);
                        const F3 no = offsetRayOrigin(gp, ge, gn, d); // isect.SpawnRay(ray.m_dir)
                        state = packState(mediumBeyond(sc, w.h.tri, stMedium(state), gn, d), stDim(state), stBounces(state), stSpec(state)) | kStNoDifferentials;
                        P.rayO[p] = mk4(no, 0.f);
                        P.rayD[p] = mk4(d, __uint_as_float(state));
                        w.start(no, d, EDEV_INF);
                        ++crossings; ++inlineRays;
                    } else {
                        kindOf[pos] = (uint8_t)kind;
                        pos = kNoRay;
                    }
                }
                if (kCrossInline) { // lanes that went on through a boundary take no new ray
                    idle = w.done();
                    idleMask = __ballot(idle);
                    nIdle = __popcll(idleMask);
                }
                if (!exhausted) {
                    uint32_t base;
                    uint32_t mine = poolTake(&poolNext, idleMask, &base);
                    if (idle && mine < chunkCount) {
                        pos = mine;
                        crossings = 0;
                        p = queue ? queue[chunk + mine] : chunk + mine;
                        float4 rd = P.rayD[p];
                        state = __float_as_uint(rd.w);
                        w.start(xyz(P.rayO[p]), xyz(rd), EDEV_INF);
                    }
                    exhausted = base + nIdle >= chunkCount;
                }
                if (__ballot(!w.done()) == 0) break;
            }
            w.template advance<false>(nodes, tris); // voted steps, also for a scene in LDS (2 % faster than whole rounds for closest hits)
        }
        nodeSteps += w.nodeSteps; triSteps += w.triSteps;
        __syncthreads();
        // ---- bin the chunk: order inside a material queue is wave, item, lane = the order of the positions in the chunk
        unsigned long long kinds = 0; // 4 bits per item
#pragma unroll
        for (int it = 0; it < kTraceItems; ++it) {
            uint32_t j = wid * (kTraceChunk / kWavesPerBlock) + it * 64u + lane; // a wave bins one contiguous quarter of the chunk
            kinds |= (unsigned long long)(j < chunkCount ? (uint32_t)kindOf[j] : 0u) << (4 * it);
        }
        // queueMask: the queues this scene and integrator can fill (bit k = queue k); the others cost a scalar test
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            uint32_t c = 0;
            if ((queueMask >> k) & 1u) {
#pragma unroll
                for (int it = 0; it < kTraceItems; ++it) c += __popcll(__ballot((uint32_t)((kinds >> (4 * it)) & 15u) == (uint32_t)k + 1u));
            }
            if (lane == 0) binCnt[k * kWavesPerBlock + wid] = c;
        }
        __syncthreads();
        if (threadIdx.x < NK) {
            uint32_t total = 0;
            for (int w2 = 0; w2 < kWavesPerBlock; ++w2) total += binCnt[threadIdx.x * kWavesPerBlock + w2];
            binBase[threadIdx.x] = total ? atomicAdd(&ctr->matCount[threadIdx.x].v, total) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (!((queueMask >> k) & 1u)) continue;
            uint32_t off = binBase[k];
            for (uint32_t w2 = 0; w2 < wid; ++w2) off += binCnt[k * kWavesPerBlock + w2];
#pragma unroll
            for (int it = 0; it < kTraceItems; ++it) {
                bool mine = (uint32_t)((kinds >> (4 * it)) & 15u) == (uint32_t)k + 1u;
                unsigned long long mask = __ballot(mine);
                if (mine) {
                    uint32_t i = chunk + wid * (kTraceChunk / kWavesPerBlock) + it * 64u + lane;
                    mq.q[k][off + __popcll(mask & lanesBelow)] = queue ? queue[i] : i;
                }
                off += __popcll(mask);
            }
        }
        __syncthreads(); // binCnt / binBase / kindOf / poolNext are rewritten by the next chunk
    }
    if (kCrossInline) { // the rays traced behind boundaries were in no queue: counted here
        for (int off = 32; off > 0; off >>= 1) inlineRays += __shfl_down(inlineRays, off);
        if (lane == 0 && inlineRays) atomicAdd(&ctr->raysClosest, (unsigned long long)inlineRays);
    }
    blockFlushVisits(stack + threadIdx.x, Walker::kRows, kBlock, sc.visits, kVisitClosest, binCnt);
    waveFlushSteps(sc.visits, kVisitClosest, nodeSteps, triSteps);
}

// SpatialLightDistribution::Lookup (Enigma/LightDist.cpp:116-131): the voxel that holds p
DEV uint32_t spatialVoxel(const DevScene& sc, F3 p) {
    F3 o = p - F3(sc.boundMin[0], sc.boundMin[1], sc.boundMin[2]); // BBox3::offset, Enigma/BBox.h:235-241
    if (sc.boundMax[0] > sc.boundMin[0]) o.x /= sc.boundMax[0] - sc.boundMin[0];
    if (sc.boundMax[1] > sc.boundMin[1]) o.y /= sc.boundMax[1] - sc.boundMin[1];
    if (sc.boundMax[2] > sc.boundMin[2]) o.z /= sc.boundMax[2] - sc.boundMin[2];
    int px = min(max((int)(o.x * sc.nVoxels[0]), 0), sc.nVoxels[0] - 1);
    int py = min(max((int)(o.y * sc.nVoxels[1]), 0), sc.nVoxels[1] - 1);
    int pz = min(max((int)(o.z * sc.nVoxels[2]), 0), sc.nVoxels[2] - 1);
    return ((uint32_t)px * (uint32_t)sc.nVoxels[1] + (uint32_t)py) * (uint32_t)sc.nVoxels[2] + (uint32_t)pz;
}
// ... and its table, filled before the kernel that asks started (k_spatial_mark saw the same point)
// A voxel without a row -- the table was full when the voxel was claimed (kSpatialClaimed), or no table was made for the
// point at all (kSpatialEmpty: the marking pass and this lookup would have to disagree about the point) -- is counted in
// spatialCtl[3] and answered from row 0 so that the read stays inside the table; the host sees the count before the pass's
// samples reach the film and fails the render instead of depositing samples lit from another voxel's table.
DEV const float* spatialLookup(const DevScene& sc, F3 p) {
    uint32_t row = sc.spatialSlot[spatialVoxel(sc, p)];
    if (row >= sc.spatialCapacity) { atomicAdd(&sc.spatialCtl[3], 1u); row = 0u; }
    return sc.spatialTable + (size_t)row * (2 * (size_t)sc.nLights + 2);
}

// Distribution1D::SampleDiscrete, Enigma/MonteCarlo.h:61-69 (std::upper_bound on the cdf)
DEV int sampleDiscrete(const float* lightCdf, const float* lightFunc, float lightFuncInt, int n, float u, float* pdf) {
    int lo = 0, len = n + 1; // first element > u in cdf[0..n]
    while (len > 0) {
        int half = len >> 1;
        if (!(u < lightCdf[lo + half])) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    int offset = max(0, lo - 1);
    offset = min(offset, n - 1); // unreachable for u < 1 (cdf[n] == 1); keeps the read in bounds
    *pdf = lightFunc[offset] / (lightFuncInt * n);
    return offset;
}

DEV float powerHeuristic(float fPdf, float gPdf) { float f = 1 * fPdf, g = 1 * gPdf; return (f * f) / (f * f + g * g); }

// EstimateDirect for a surface interaction (Enigma/Integrator.cpp:94-204, handleMedia = specular = false),
// up to the two rays it traces: the shadow ray of the light sample (Ld = its contribution if unoccluded)
// and the BSDF-sampled ray (miF = f * weight / scatteringPdf, multiplies the light's Le if the closest hit
// is the light's own triangle).  The caller scales both by throughput / light-choice pdf and queues them.
// ---------------------------------------------------------------- InfiniteAreaLight (Enigma/Lights.cpp:395-546)
DEV bool isDeltaLight(uint32_t type) { return type == 0u || type == 2u || type == 3u; } // point, spot, distant (Enigma/Lights.h:15-18)
DEV F3 mul3(const float* r0, const float* r1, const float* r2, F3 v) {
    return F3(r0[0] * v.x + r0[1] * v.y + r0[2] * v.z, r1[0] * v.x + r1[1] * v.y + r1[2] * v.z, r2[0] * v.x + r2[1] * v.y + r2[2] * v.z);
}
// MipMap::lookup(st) with width 0 = triangle(0, st), repeat wrap (Enigma/MipMap.h:310-322,328-339,355-359)
DEV F3 envLookup(const DevEnvMap& m, float s0f, float t0f) {
    float s = s0f * m.w - 0.5f, t = t0f * m.h - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(t);
    float ds = s - s0, dt = t - t0;
    auto texel = [&](int a, int b) { return xyz(m.texels[(size_t)((uint32_t)b % (uint32_t)m.h) * m.w + ((uint32_t)a % (uint32_t)m.w)]); };
    return (1 - ds) * (1 - dt) * texel(s0, t0) + (1 - ds) * dt * texel(s0, t0 + 1) + ds * (1 - dt) * texel(s0 + 1, t0) + ds * dt * texel(s0 + 1, t0 + 1);
}
DEV float sphericalTheta(F3 n) { return acosf(clampf(n.z, -1.0f, 1.0f)); }
DEV float sphericalPhi(F3 n) { const float p = atan2f(n.y, n.x); return (p < 0.0f) ? p + kPi * 2.0f : p; }
// InfiniteAreaLight::Le(ray) (Enigma/Lights.cpp:452-457)
DEV F3 infiniteLe(const DevScene& sc, const DevLight& lt, F3 d) {
    F3 w = normalize(mul3(lt.w2l0, lt.w2l1, lt.w2l2, d));
    return envLookup(sc.envMaps[lt.tri], sphericalPhi(w) * (1.0f / (kPi * 2.0f)), sphericalTheta(w) * kInvPi);
}
// Distribution1D::SampleContinuous (Enigma/MonteCarlo.h:42-59) on a cdf of n + 1 entries
DEV float sampleContinuous1D(const float* func, const float* cdf, float funcInt, int n, float u, float* pdf, int* off) {
    int lo = 0, len = n + 1; // std::upper_bound
    while (len > 0) {
        int half = len >> 1;
        if (!(u < cdf[lo + half])) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    int offset = max(0, lo - 1);
    offset = min(offset, n - 1); // unreachable for u < 1; keeps the reads in bounds
    *off = offset;
    float du = (u - cdf[offset]) / (cdf[offset + 1] - cdf[offset]);
    *pdf = func[offset] / funcInt;
    return (offset + du) / n;
}
// Distribution2D::Pdf (Enigma/MonteCarlo.h:94-102)
DEV float envPdf(const DevEnvMap& m, float u, float v) {
    int iu = min(max((int)(u * m.nu), 0), m.nu - 1);
    int iv = min(max((int)(v * m.nv), 0), m.nv - 1);
    if (m.condInt[iv] * m.margInt == 0.f) return 0.f;
    return (m.condFunc[(size_t)iv * m.nu + iu] * m.margFunc[iv]) / (m.condInt[iv] * m.margInt);
}
// InfiniteAreaLight::Pdf_Li (Enigma/Lights.cpp:464-472); theta is scaled by 1 / (2 pi) there, as in the reference
DEV float infinitePdfLi(const DevScene& sc, const DevLight& lt, F3 w) {
    F3 wi = mul3(lt.w2l0, lt.w2l1, lt.w2l2, w);
    float theta = sphericalTheta(wi), phi = sphericalPhi(wi);
    float sinTheta = sinf(theta);
    if (sinTheta == 0) return 0;
    const float kInvTwoPi = 1.0f / (kPi * 2.0f);
    return envPdf(sc.envMaps[lt.tri], phi * kInvTwoPi, theta * kInvTwoPi) / (2 * kPi * kPi * sinTheta);
}

struct DirectSample {
    bool shadow, mis;
    F3 shO, shD, Ld;
    F3 miO, miD, miF;
    F3 tgtP, tgtE, tgtN; // the light-side end of the shadow ray as an Interaction (p, pError, n): VisibilityTester::p1
};
// Light::Sample_Li for a reference point (Enigma/Lights.cpp:117-124,165-172,634-645,696-703): direction, radiance,
// pdf and the sampled point as an Interaction; q0..q2 are the area light's vertices (read again by Shape::pdf).
struct LightSample { F3 wi, Li, lp, lpErr, ln, q0, q1, q2; float pdf; };
// SPHL: lights on spheres are compiled in (the builds for scenes with spheres);
// refErr / refN: the reference point's error bounds and normal, which Sphere::sample offsets it by (zero for a medium point)
template <bool SPHL = false>
DEV LightSample sampleLight(const DevScene& sc, const DevLight& lt, const float4* tris, F3 refP, float ulx, float uly, F3 refErr = F3(0.f), F3 refN = F3(0.f)) {
    LightSample s;
    F3 wi(0.f), Li(0.f), lp(0.f), lpErr(0.f), ln(0.f);
    float lightPdf = 0.f;
    F3 q0(0.f), q1(0.f), q2(0.f);
    bool lflip = false;
    if (lt.type == 1) {
        const float4* tp = tris + (size_t)lt.tri * 3;
        float4 a0 = tp[0], a1 = tp[1], a2 = tp[2];
        q0 = xyz(a0); q1 = xyz(a1); q2 = xyz(a2);
        lflip = lt.sampleFlip != 0.f;
        // Triangle::sample (Enigma/TriangleMesh.cpp:604-632)
        float su1 = sqrtf(ulx);
        float tb0 = 1.f - su1, tb1 = uly * su1;
        lp = tb0 * q0 + tb1 * q1 + (1 - tb0 - tb1) * q2;
        ln = normalize(cross(q1 - q0, q2 - q0));
        if (lflip) ln = ln * -1.f;
        lpErr = gammaf(6) * (fabs3(tb0 * q0) + fabs3(tb1 * q1) + fabs3((1 - tb0 - tb1) * q2));
        lightPdf = 1 / lt.area;
        // Shape::sample(ref, u) (Enigma/Shape.cpp:54-69)
        F3 w = lp - refP;
        if (lengthSqr(w) == 0) lightPdf = 0;
        else {
            w = normalize(w);
            lightPdf *= lengthSqr(refP - lp) / absdot(ln, -w);
            if (isinf(lightPdf)) lightPdf = 0.f;
        }
        // DiffuseAreaLight::Sample_Li (Enigma/Lights.cpp:634-645)
        if (lightPdf == 0 || lengthSqr(lp - refP) == 0) lightPdf = 0;
        else {
            wi = normalize(lp - refP);
            if (lt.twoSided || dot(ln, -wi) > 0) Li = F3(lt.L[0], lt.L[1], lt.L[2]);
        }
    } else if (SPHL && lt.type == 5) { // Sphere::sample(ref, u) under DiffuseAreaLight::Sample_Li (Enigma/Lights.cpp:634-645)
        sphereSampleFrom(sc.spheres[lt.tri], refP, refErr, refN, ulx, uly, &lp, &lpErr, &ln, &lightPdf);
        if (lightPdf == 0 || lengthSqr(lp - refP) == 0) lightPdf = 0;
        else {
            wi = normalize(lp - refP);
            if (lt.twoSided || dot(ln, -wi) > 0) Li = F3(lt.L[0], lt.L[1], lt.L[2]);
        }
    } else if (lt.type == 4) { // InfiniteAreaLight::Sample_Li (Enigma/Lights.cpp:516-539)
        const DevEnvMap& em = sc.envMaps[lt.tri];
        float pv, pu;
        int v, uo;
        float d1 = sampleContinuous1D(em.margFunc, em.margCdf, em.margInt, em.nv, uly, &pv, &v);
        float d0 = sampleContinuous1D(em.condFunc + (size_t)v * em.nu, em.condCdf + (size_t)v * (em.nu + 1), em.condInt[v], em.nu, ulx, &pu, &uo);
        float mapPdf = pu * pv;
        if (mapPdf != 0) {
            float theta = d1 * kPi, phi = d0 * 2 * kPi;
            float cosTheta = cosf(theta), sinTheta = sinf(theta);
            float sinPhi = sinf(phi), cosPhi = cosf(phi);
            wi = mul3(lt.l2w0, lt.l2w1, lt.l2w2, F3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
            lightPdf = mapPdf / (2 * kPi * kPi * sinTheta);
            if (sinTheta == 0) lightPdf = 0;
            lp = refP + wi * (2 * lt.worldRadius);
            Li = envLookup(em, d0, d1);
        }
    } else if (lt.type == 3) { // DistantLight::Sample_Li (Enigma/Lights.cpp:696-703): wi = m_lightDir as stored
        wi = F3(lt.pos[0], lt.pos[1], lt.pos[2]);
        lightPdf = 1.f;
        lp = refP + wi * (2 * lt.worldRadius);
        Li = F3(lt.L[0], lt.L[1], lt.L[2]);
    } else { // PointLight::Sample_Li (Enigma/Lights.cpp:117-124), SpotLight::Sample_Li (:165-172)
        lp = F3(lt.pos[0], lt.pos[1], lt.pos[2]);
        wi = normalize(lp - refP);
        lightPdf = 1.f;
        Li = F3(lt.L[0], lt.L[1], lt.L[2]);
        if (lt.type == 2) { // SpotLight::Falloff(-wi) (Enigma/Lights.cpp:180-190)
            F3 w = -wi;
            F3 wl = normalize(F3(lt.w2l0[0] * w.x + lt.w2l0[1] * w.y + lt.w2l0[2] * w.z, lt.w2l1[0] * w.x + lt.w2l1[1] * w.y + lt.w2l1[2] * w.z,
                                 lt.w2l2[0] * w.x + lt.w2l2[1] * w.y + lt.w2l2[2] * w.z));
            float cosTheta = wl.z, fall;
            if (cosTheta < lt.cosTotalWidth) fall = 0.f;
            else if (cosTheta >= lt.cosFalloffStart) fall = 1.f;
            else {
                float delta = (cosTheta - lt.cosTotalWidth) / (lt.cosFalloffStart - lt.cosTotalWidth);
                fall = (delta * delta) * (delta * delta);
            }
            Li = Li * fall;
        }
        Li = Li / lengthSqr(lp - refP);
    }
    s.wi = wi; s.Li = Li; s.lp = lp; s.lpErr = lpErr; s.ln = ln; s.q0 = q0; s.q1 = q1; s.q2 = q2; s.pdf = lightPdf;
    return s;
}
// Shape::pdf(ref, wi) for a light's own triangle (Enigma/Shape.cpp:80-98): the ray from o along wi against (q0, q1, q2)
DEV float lightTrianglePdf(F3 o, F3 refP, F3 wi, F3 q0, F3 q1, F3 q2, float area) {
    RaySetup rs = setupRay(o, wi);
    float lt_t, lb0, lb1, lb2;
    float pdf = 0.f;
    if (triangleTest(rs, EDEV_INF, q0, q1, q2, &lt_t, &lb0, &lb1, &lb2)) {
        F3 pl = lb0 * q0 + lb1 * q1 + lb2 * q2;
        F3 nl = normalize(cross(q0 - q2, q1 - q2));
        pdf = lengthSqr(refP - pl) / (absdot(nl, -wi) * area);
        if (isinf(pdf)) pdf = 0.f;
    }
    return pdf;
}
template <int MK, bool SPHL = false, class BS = Bsdf>
DEV DirectSample estimateDirect(const DevScene& sc, const BS& bsdf, const HitGeom& g, F3 woN, F3 ns, const DevLight& lt, const float4* tris,
                                float ulx, float uly, float usx, float usy) {
    const int kNonSpecular = BSDF_ALL & ~BSDF_SPECULAR;
    DirectSample r;
    r.shadow = false; r.mis = false;
    // ---- light sampling (Enigma/Integrator.cpp:104-148) ----
    LightSample ls = sampleLight<SPHL>(sc, lt, tris, g.p, ulx, uly, g.pError, g.n);
    F3 wi = ls.wi, Li = ls.Li, lp = ls.lp, lpErr = ls.lpErr, ln = ls.ln, q0 = ls.q0, q1 = ls.q1, q2 = ls.q2;
    float lightPdf = ls.pdf;
    if (lightPdf > 0 && !isBlack(Li)) {
        F3 f = bsdfF<MK>(bsdf, woN, wi, kNonSpecular) * absdot(wi, ns);
        float scatteringPdf = bsdfPdf<MK>(bsdf, woN, wi, kNonSpecular);
        if (!isBlack(f)) {
            // VisibilityTester(ref, pShape) -> ref.SpawnRayTo(pShape) (Enigma/Interaction.cpp:58-64)
            F3 origin = offsetRayOrigin(g.p, g.pError, g.n, lp - g.p);
            F3 target = offsetRayOrigin(lp, lpErr, ln, origin - lp);
            if (isDeltaLight(lt.type)) r.Ld = f * Li / lightPdf;
            else r.Ld = f * Li * powerHeuristic(lightPdf, scatteringPdf) / lightPdf;
            r.shO = origin; r.shD = target - origin;
            r.tgtP = lp; r.tgtE = lpErr; r.tgtN = ln;
            r.shadow = true;
        }
    }
    // ---- BSDF sampling (Enigma/Integrator.cpp:150-199) ----
    if (!isDeltaLight(lt.type)) {
        F3 wi2(0.f);
        float scatPdf2; int sampledType;
        F3 f2 = bsdfSample<MK>(bsdf, woN, &wi2, usx, usy, &scatPdf2, kNonSpecular, &sampledType);
        f2 = f2 * absdot(wi2, ns);
        if (!isBlack(f2) && scatPdf2 > 0) {
            // light.Pdf_Li -> Shape::pdf (Enigma/Shape.cpp:80-98): intersect the light's own triangle
            F3 o2 = offsetRayOrigin(g.p, g.pError, g.n, wi2);
            float lightPdf2 = lt.type == 4 ? infinitePdfLi(sc, lt, wi2)
                              : (SPHL && lt.type == 5) ? spherePdfFrom(sc.spheres[lt.tri], g.p, g.pError, g.n, o2, wi2)
                                                       : lightTrianglePdf(o2, g.p, wi2, q0, q1, q2, lt.area);
            if (lightPdf2 != 0) {
                float weight = powerHeuristic(scatPdf2, lightPdf2);
                r.miO = o2; r.miD = wi2;
                r.miF = f2 * weight / scatPdf2;
                r.mis = true;
            }
        }
    }
    return r;
}

DEV void storeShadowRay(const DevRayQueue& q, size_t slot, F3 o, F3 d, F3 w) {
    q.o[slot] = make_float4(o.x, o.y, o.z, d.x);
    q.d[slot] = make_float4(d.y, d.z, w.x, w.y);
    q.wz[slot] = w.z;
}

// For scenes small enough (LDS_SCENE) the block first copies triangles, materials, lights and the light
// distribution into LDS: the chain hit triangle -> material -> light -> light triangle is then four LDS
// reads instead of four dependent trips to L2/HBM per path.
// (a macro, not a function: the pointers must be visibly derived from the __shared__ array for the
// compiler to address them as LDS)
#define ENIGMA_STAGE_SHADE_TABLES                                                                                        \
    const float4* tris = sc.tris;                                                                                        \
    const DevMaterial* materials = sc.materials;                                                                         \
    const DevLight* lights = sc.lights;                                                                                  \
    const float* lightCdf = sc.lightCdf;                                                                                 \
    const float* lightFunc = sc.lightFunc;                                                                               \
    if (LDS_SCENE) {                                                                                                     \
        const uint32_t nT = sc.nTris * 3u, nM = sc.nMaterials * 4u, nL = (uint32_t)sc.nLights * kLightVec4;                      \
        float4* ldsMat = sceneLds + nT;                                                                                  \
        float4* ldsLight = ldsMat + nM;                                                                                  \
        float* ldsCdf = reinterpret_cast<float*>(ldsLight + nL);                                                         \
        float* ldsFunc = ldsCdf + sc.nLights + 1;                                                                        \
        const float4* gMat = reinterpret_cast<const float4*>(sc.materials);                                              \
        const float4* gLight = reinterpret_cast<const float4*>(sc.lights);                                               \
        for (uint32_t i = threadIdx.x; i < nT; i += kBlock) sceneLds[i] = sc.tris[i];                                    \
        for (uint32_t i = threadIdx.x; i < nM; i += kBlock) ldsMat[i] = gMat[i];                                         \
        for (uint32_t i = threadIdx.x; i < nL; i += kBlock) ldsLight[i] = gLight[i];                                     \
        for (uint32_t i = threadIdx.x; i < (uint32_t)sc.nLights + 1u; i += kBlock) ldsCdf[i] = sc.lightCdf[i];           \
        for (uint32_t i = threadIdx.x; i < (uint32_t)sc.nLights; i += kBlock) ldsFunc[i] = sc.lightFunc[i];              \
        __syncthreads();                                                                                                 \
        tris = sceneLds;                                                                                                 \
        materials = reinterpret_cast<const DevMaterial*>(ldsMat);                                                        \
        lights = reinterpret_cast<const DevLight*>(ldsLight);                                                            \
        lightCdf = ldsCdf;                                                                                               \
        lightFunc = ldsFunc;                                                                                             \
    }

// VOL: the surface branch of VolPathIntegrator::Li (Enigma/VolPathIntegrator.cpp:113-136): direct lighting is
// sampled for every BSDF (no test for non-specular lobes), and every ray leaving the surface takes the medium on
// its side of the surface (Interaction::GetMedium, Enigma/Interaction.cpp:72-75).
// SPH: the build for scenes with spheres, the only one that can sample a light on a sphere
// Waves per SIMD asked of the compiler: four for the Lambertian and smooth-glass builds (128 registers, some thirty
// spilled), three for the microfacet ones (168).  The kernels wait on memory more than they compute; measured against
// the unconstrained builds (165 to 199 registers): Cornell 3 % faster per step, BigMesh 1 %; one wave more each
// (96 / 128 registers, a hundred spilled) is 15 % slower.
// (re-measured with device code at -O1, where nothing is spilled at four / three waves: five / four 9-16 % slower, three /
// three 3 % slower, four / four and four / two the same within 1 %)
template <int MK, bool LDS_SCENE, bool VOL = false, bool SPH = false, bool HAL = false>
#ifndef ENIGMA_EXP_SHADE_OCC_A
#define ENIGMA_EXP_SHADE_OCC_A 4
#endif
#ifndef ENIGMA_EXP_SHADE_OCC_B
#define ENIGMA_EXP_SHADE_OCC_B 3
#endif
__global__ void __launch_bounds__(kBlock, (MK == MK_MATTE || MK == MK_MATTE_ROUGH || MK == MK_GLASS_SMOOTH) ? ENIGMA_EXP_SHADE_OCC_A : ENIGMA_EXP_SHADE_OCC_B) k_shade(DevScene sc, DevPaths P, DevSampler S, DevIntegrator I, DevRayQueue shq, DevRayQueue misq,
                                                  const uint32_t* matQueue, uint32_t* nextQueue, uint32_t nextIdx, DevCounters* ctr) {
    __shared__ uint32_t lds[32];
    extern __shared__ float4 sceneLds[];
    const uint32_t count = ctr->matCount[MK].v;
    if (blockIdx.x * kBlock >= count) return;
    ENIGMA_STAGE_SHADE_TABLES
    const int kNonSpecular = BSDF_ALL & ~BSDF_SPECULAR;
    uint32_t parity = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock, parity ^= 16u) {
        uint32_t i = base + threadIdx.x;
        bool active = i < count;
        bool pushShadow = false, pushMis = false, pushNext = false;
        uint32_t p = 0;
        F3 shRayO(0.f), shRayD(0.f), shRayW(0.f);
        if (active) {
            p = matQueue[i];
            float4 rdv = P.rayD[p];
            F3 d = xyz(rdv);
            float4 hv = P.hit[p];
            int tri = __float_as_int(hv.x);
            uint32_t st = __float_as_uint(rdv.w);
            uint32_t dim = stDim(st), bounces = stBounces(st);
            uint32_t px, py, slot; // pixel relative to the sample-bounds origin
            pathPixel(I, S, p, &px, &py, &slot);
            float4 bv = P.beta[p];
            F3 beta = xyz(bv);
            float etaScale = bv.w;
            HitGeom g = hitGeometry(sc, tris, tri, hv.y, hv.z, hv.w, d, p);
            typename BsdfOf<MK>::T bsdf;
            makeBsdf<MK>(materials[g.matIdx], g.n, g.ns, g.dpdu, &bsdf, sc.texColors ? sc.texColors + (size_t)p * 2 : nullptr, sc.lobes, true);
            F3 ns = g.ns;
            F3 woN = g.wo; // Interaction::m_wo
            // medium + 1 on either side of the surface: the triangle's MediumInterface if it is a transition, else
            // the medium the ray arrived in (GeometricPrimitive::intersect, Enigma/Primitive.cpp:155-160)
            uint32_t medIn = 0, medOut = 0;
            if (VOL) {
                medIn = medOut = stMedium(st);
                if (sc.triMedium) {
                    uint32_t tm = sc.triMedium[tri], a = tm & 0xffffu, b = tm >> 16;
                    if (a != b) { medIn = a; medOut = b; }
                }
            }
            auto mediumTowards = [&](F3 w) { return dot(w, g.n) > 0 ? medOut : medIn; };

            // ---- UniformSampleOneLight (Enigma/Integrator.cpp:71-92) ----
            if ((VOL || numComponents<MK>(bsdf, kNonSpecular) > 0) && sc.nLights > 0) {
                float pick;
                // m_lightDistribution->Lookup(isect.m_p), Enigma/PathIntegrator.cpp:95.  The per-voxel tables are read by the
                // build of this kernel that keeps its tables in global memory (scenes with a spatial distribution are run
                // through it); the LDS build knows nothing of them and keeps its registers -- seven more cost it a wave per SIMD
                const float* pickFunc = lightFunc;
                const float* pickCdf = lightCdf;
                float pickInt = sc.lightFuncInt;
                if (!LDS_SCENE && sc.spatialTable) { pickFunc = spatialLookup(sc, g.p); pickCdf = pickFunc + sc.nLights; pickInt = pickFunc[2 * sc.nLights + 1]; }
                int lightNum = sampleDiscrete(pickCdf, pickFunc, pickInt, sc.nLights, sampleDimension<HAL>(S, px, py, slot, dim), &pick);
                dim += 1;
                if (pick != 0) {
                    float ulx = sampleDimension<HAL>(S, px, py, slot, dim), uly = sampleDimension<HAL>(S, px, py, slot, dim + 1);
                    float usx = sampleDimension<HAL>(S, px, py, slot, dim + 2), usy = sampleDimension<HAL>(S, px, py, slot, dim + 3);
                    dim += 4;
                    DirectSample ds = estimateDirect<MK, SPH>(sc, bsdf, g, woN, ns, lights[lightNum], tris, ulx, uly, usx, usy);
                    if (ds.shadow) {
                        if (VOL) storeShadowRay(shq, p, ds.shO, ds.shD, beta * (ds.Ld / pick));
                        else { shRayO = ds.shO; shRayD = ds.shD; shRayW = beta * (ds.Ld / pick); } // stored at its queue position below
                        if (VOL) {
                            shq.tgtP[p] = mk4(ds.tgtP, __uint_as_float(mediumTowards(ds.shD)));
                            shq.tgtE[p] = mk4(ds.tgtE, 0.f);
                            shq.tgtN[p] = mk4(ds.tgtN, 0.f);
                        }
                        pushShadow = true;
                    }
                    if (ds.mis) {
                        misq.o[p] = mk4(ds.miO, VOL ? __uint_as_float(mediumTowards(ds.miD)) : 0.f);
                        misq.d[p] = mk4(ds.miD, __int_as_float(lightNum));
                        misq.w[p] = mk4(beta * (ds.miF / pick), 0.f);
                        pushMis = true;
                    }
                }
            }

            // ---- BSDF sampling for the next path vertex (Enigma/PathIntegrator.cpp:114-175) ----
            F3 wo = -d, wi(0.f);
            float pdf; int flags;
            float u0 = sampleDimension<HAL>(S, px, py, slot, dim), u1 = sampleDimension<HAL>(S, px, py, slot, dim + 1);
            dim += 2;
            F3 f = bsdfSample<MK>(bsdf, wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
            if (!(isBlack(f) || pdf == 0.f)) {
                beta *= f * absdot(wi, ns) / pdf;
                uint32_t spec = (flags & BSDF_SPECULAR) ? 1u : 0u;
                if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                    float eta = bsdf.eta;
                    etaScale *= (dot(wo, g.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                }
                F3 no = offsetRayOrigin(g.p, g.pError, g.n, wi); // isect.SpawnRay(wi)
                bool alive = true;
                F3 rrBeta = beta * etaScale;
                if (maxComponentValue(rrBeta) < I.rrThreshold && bounces > 3) {
                    float q = smax(.05f, 1.0f - maxComponentValue(rrBeta));
                    float ur = sampleDimension<HAL>(S, px, py, slot, dim);
                    dim += 1;
                    if (ur < q) alive = false;
                    else beta = beta / (1 - q);
                }
                if (alive) {
                    P.rayO[p] = mk4(no, 0.f);
                    P.rayD[p] = mk4(wi, __uint_as_float(packState(VOL ? mediumTowards(wi) : 0u, dim, bounces + 1, spec)));
                    P.beta[p] = mk4(beta, etaScale);
                    pushNext = true;
                }
            }
        }
        uint32_t sSh, sMi, sNx;
        blockReserve3(pushShadow, pushMis, pushNext, &ctr->shadowCount.v, &ctr->misCount.v, &ctr->extCount[nextIdx].v, lds + parity, &sSh, &sMi, &sNx);
        if (pushShadow) {
            shq.idx[sSh] = p;
            if (!VOL) storeShadowRay(shq, sSh, shRayO, shRayD, shRayW);
        }
        if (pushMis) misq.idx[sMi] = p;
        if (pushNext) nextQueue[sNx] = p;
    }
}

// ---------------------------------------------------------------- DirectLightingIntegrator
// Block-wide append of a COUNT of entries per thread to three queues (shadow rays, MIS rays, continuing
// paths) behind one pair of barriers: wave-inclusive scans by shuffle, wave totals through LDS, one
// atomic per block and queue.  lds: 16 words.
DEV void blockReserveCounts(uint32_t na, uint32_t nb, uint32_t nc, uint32_t* ca, uint32_t* cb, uint32_t* cc, uint32_t* lds,
                            uint32_t* sa, uint32_t* sb, uint32_t* sc) {
    uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t ia = na, ib = nb, ic = nc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t ta = __shfl_up(ia, d), tb = __shfl_up(ib, d), tc = __shfl_up(ic, d);
        if ((int)lane >= d) { ia += ta; ib += tb; ic += tc; }
    }
    if (lane == 63u) { lds[wid] = ia; lds[4 + wid] = ib; lds[8 + wid] = ic; }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t total = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) total += lds[threadIdx.x * 4 + w];
        uint32_t* ctr = threadIdx.x == 0 ? ca : (threadIdx.x == 1 ? cb : cc);
        lds[12 + threadIdx.x] = total ? atomicAdd(ctr, total) : 0u;
    }
    __syncthreads();
    uint32_t oa = lds[12], ob = lds[13], oc = lds[14];
    for (uint32_t w = 0; w < wid; ++w) { oa += lds[w]; ob += lds[4 + w]; oc += lds[8 + w]; }
    *sa = oa + ia - na; *sb = ob + ib - nb; *sc = oc + ic - nc;
}

// Resume the innermost pending branch of a path whose current branch has ended: every pending entry
// stands for a SpecularTransmit call the recursion has yet to make (one Get2D each); placeholders of
// nodes without a transmission lobe only consume their two dimensions.  Returns false when the stack
// runs empty (the camera sample is complete).
DEV bool resumePending(const DevPaths& P, const DevDirect& D, const DevIntegrator& I, uint32_t p, uint32_t dim) {
    uint32_t cnt = D.pendCount[p];
    bool found = false;
    while (cnt > 0 && !found) {
        --cnt;
        dim += 2;
        size_t e = (size_t)p * I.pendingLevels + cnt;
        float4 o = D.pendO[e];
        if (o.w != 0.f) {
            float4 d = D.pendD[e], b = D.pendB[e];
            P.rayO[p] = make_float4(o.x, o.y, o.z, 0.f);
            P.rayD[p] = make_float4(d.x, d.y, d.z, __uint_as_float(packState(0u, dim, (uint32_t)d.w, 0u)));
            P.beta[p] = make_float4(b.x, b.y, b.z, 1.f);
            found = true;
        }
    }
    D.pendCount[p] = cnt;
    return found;
}

// The body of DirectLightingIntegrator::Li at a hit (Enigma/DirectLightingIntegrator.cpp:49-68) for the paths
// of one material kind: direct lighting from all lights (sample arrays) or one uniformly chosen light,
// then SpecularReflect / SpecularTransmit.  Smooth glass carries two specular lobes here (allowMultipleLobes
// is false on this path).  A path's shadow / MIS rays go to its raysPerPath ray slots, packed from slot 0.
template <int MK, bool LDS_SCENE, bool SPH = false, bool HAL = false>
__global__ void __launch_bounds__(kBlock) k_shade_direct(DevScene sc, DevPaths P, DevSampler S, DevIntegrator I, DevDirect D, DevRayQueue shq,
                                                         DevRayQueue misq, const uint32_t* matQueue, uint32_t* nextQueue, uint32_t nextIdx,
                                                         DevCounters* ctr) {
    constexpr int MKD = (MK == MK_GLASS_SMOOTH) ? (int)MK_DL_GLASS : MK;
    __shared__ uint32_t lds[32];
    extern __shared__ float4 sceneLds[];
    const uint32_t count = ctr->matCount[MK].v;
    if (blockIdx.x * kBlock >= count) return;
    ENIGMA_STAGE_SHADE_TABLES
    (void)lightCdf; (void)lightFunc;
    uint32_t parity = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock, parity ^= 16u) {
        uint32_t i = base + threadIdx.x;
        bool active = i < count;
        uint32_t nShadow = 0, nMis = 0, nNext = 0;
        uint32_t p = 0;
        if (active) {
            p = matQueue[i];
            float4 rdv = P.rayD[p];
            F3 d = xyz(rdv);
            float4 hv = P.hit[p];
            int tri = __float_as_int(hv.x);
            uint32_t st = __float_as_uint(rdv.w);
            uint32_t dim = stDim(st), depth = stBounces(st);
            uint32_t px, py, slot;
            pathPixel(I, S, p, &px, &py, &slot);
            F3 beta = xyz(P.beta[p]);
            HitGeom g = hitGeometry(sc, tris, tri, hv.y, hv.z, hv.w, d, p);
            typename BsdfOf<MKD>::T bsdf;
            makeBsdf<MKD>(materials[g.matIdx], g.n, g.ns, g.dpdu, &bsdf, sc.texColors ? sc.texColors + (size_t)p * 2 : nullptr, sc.lobes, false);
            F3 ns = g.ns;
            F3 woN = g.wo; // isect.m_wo
            const size_t raySlot0 = (size_t)p * I.raysPerPath;
            auto emit = [&](const DirectSample& ds, float div, int lightNum) { // div: sample count (all) or light-choice pdf (one)
                if (ds.shadow) {
                    size_t q = raySlot0 + nShadow++;
                    storeShadowRay(shq, q, ds.shO, ds.shD, beta * (ds.Ld / div));
                }
                if (ds.mis) {
                    size_t q = raySlot0 + nMis++;
                    misq.o[q] = mk4(ds.miO, 0.f);
                    misq.d[q] = mk4(ds.miD, __int_as_float(lightNum));
                    misq.w[q] = mk4(beta * (ds.miF / div), 0.f);
                }
            };
            if (sc.nLights > 0) {
                if (I.dlStrategy == 0) { // UniformSampleAllLights, Enigma/Integrator.cpp:42-69
                    uint32_t a = D.arrayOff[p];
                    const unsigned long long sampleNum = S.sampleBase + slot;
                    for (int l = 0; l < sc.nLights; ++l) {
                        const DevLight& lt = lights[l];
                        const uint32_t n = lt.nSamples;
                        const uint32_t dimL = 5u + 2u * (a + 2u * (uint32_t)l), dimS = dimL + 2u; // uLightArray, uScatteringArray
                        for (uint32_t k = 0; k < n; ++k) {
                            unsigned long long sn = sampleNum * n + k; // entry of the array that belongs to this sample
                            float ulx = sampleDimensionOf<HAL>(S, px, py, sn, dimL), uly = sampleDimensionOf<HAL>(S, px, py, sn, dimL + 1);
                            float usx = sampleDimensionOf<HAL>(S, px, py, sn, dimS), usy = sampleDimensionOf<HAL>(S, px, py, sn, dimS + 1);
                            DirectSample ds = estimateDirect<MKD, SPH>(sc, bsdf, g, woN, ns, lt, tris, ulx, uly, usx, usy);
                            emit(ds, (float)n, l);
                        }
                    }
                    D.arrayOff[p] = a + 2u * (uint32_t)sc.nLights;
                } else { // UniformSampleOneLight without a distribution, Enigma/Integrator.cpp:83-91
                    int nl = sc.nLights;
                    int lightNum = min((int)(sampleDimension<HAL>(S, px, py, slot, dim) * nl), nl - 1);
                    float lightPdf = 1.0f / nl;
                    float ulx = sampleDimension<HAL>(S, px, py, slot, dim + 1), uly = sampleDimension<HAL>(S, px, py, slot, dim + 2);
                    float usx = sampleDimension<HAL>(S, px, py, slot, dim + 3), usy = sampleDimension<HAL>(S, px, py, slot, dim + 4);
                    dim += 5;
                    DirectSample ds = estimateDirect<MKD, SPH>(sc, bsdf, g, woN, ns, lights[lightNum], tris, ulx, uly, usx, usy);
                    emit(ds, lightPdf, lightNum);
                }
            }
            // ---- SpecularReflect, SpecularTransmit (Enigma/Integrator.cpp:329-444) ----
            bool haveNext = false;
            if ((int)depth + 1 < I.maxDepth) {
                F3 wiR(0.f), wiT(0.f);
                float pdfR, pdfT; int tyR, tyT;
                // SpecularReflect's Get2D, then SpecularTransmit's.  A specular lobe ignores the values, but BSDF::sample_f picks
                // among several matching lobes with the first (Enigma/BxDF.cpp:895): only a list of lobes can hold two specular
                // lobes of one side.  With a reflection branch taken first, SpecularTransmit draws its values after that whole
                // subtree; materials for which that matters are refused by the host under this integrator.
                float uR0 = 0.5f, uR1 = 0.5f, uT0 = 0.5f, uT1 = 0.5f;
                if (MKD == MK_GENERAL) {
                    uR0 = sampleDimension<HAL>(S, px, py, slot, dim); uR1 = sampleDimension<HAL>(S, px, py, slot, dim + 1);
                    uT0 = sampleDimension<HAL>(S, px, py, slot, dim + 2); uT1 = sampleDimension<HAL>(S, px, py, slot, dim + 3);
                }
                dim += 2;
                F3 fR = bsdfSample<MKD>(bsdf, woN, &wiR, uR0, uR1, &pdfR, BSDF_REFLECTION | BSDF_SPECULAR, &tyR);
                bool validR = pdfR > 0.f && !isBlack(fR) && absdot(wiR, ns) != 0.f;
                F3 fT = bsdfSample<MKD>(bsdf, woN, &wiT, uT0, uT1, &pdfT, BSDF_TRANSMISSION | BSDF_SPECULAR, &tyT);
                bool validT = pdfT > 0.f && !isBlack(fT) && absdot(wiT, ns) != 0.f;
                F3 betaT = validT ? beta * (fT * absdot(wiT, ns) / pdfT) : F3(0.f);
                F3 oT = validT ? offsetRayOrigin(g.p, g.pError, g.n, wiT) : F3(0.f);
                if (validR) {
                    // the transmission branch (or its placeholder) waits until the reflection subtree is through
                    uint32_t cnt = D.pendCount[p];
                    size_t e = (size_t)p * I.pendingLevels + cnt;
                    D.pendO[e] = mk4(oT, validT ? 1.f : 0.f);
                    D.pendD[e] = mk4(wiT, (float)(depth + 1));
                    D.pendB[e] = mk4(betaT, 0.f);
                    D.pendCount[p] = cnt + 1;
                    P.rayO[p] = mk4(offsetRayOrigin(g.p, g.pError, g.n, wiR), 0.f);
                    P.rayD[p] = mk4(wiR, __uint_as_float(packState(0u, dim, depth + 1, 0u)));
                    P.beta[p] = mk4(beta * (fR * absdot(wiR, ns) / pdfR), 1.f);
                    haveNext = true;
                } else {
                    dim += 2; // SpecularTransmit's Get2D
                    if (validT) {
                        P.rayO[p] = mk4(oT, 0.f);
                        P.rayD[p] = mk4(wiT, __uint_as_float(packState(0u, dim, depth + 1, 0u)));
                        P.beta[p] = mk4(betaT, 1.f);
                        haveNext = true;
                    }
                }
            }
            if (!haveNext) haveNext = resumePending(P, D, I, p, dim);
            nNext = haveNext ? 1u : 0u;
        }
        uint32_t sSh, sMi, sNx;
        blockReserveCounts(nShadow, nMis, nNext, &ctr->shadowCount.v, &ctr->misCount.v, &ctr->extCount[nextIdx].v, lds + parity, &sSh, &sMi, &sNx);
        for (uint32_t j = 0; j < nShadow; ++j) shq.idx[sSh + j] = p * (uint32_t)I.raysPerPath + j;
        for (uint32_t j = 0; j < nMis; ++j) misq.idx[sMi + j] = p * (uint32_t)I.raysPerPath + j;
        if (nNext) nextQueue[sNx] = p;
    }
}

// Paths whose ray left the scene while branches are pending (queue MK_COUNT of k_intersect_closest<DL>).
__global__ void __launch_bounds__(kBlock) k_resume_pending(DevPaths P, DevIntegrator I, DevDirect D, const uint32_t* queue, uint32_t* nextQueue,
                                                           uint32_t nextIdx, DevCounters* ctr) {
    __shared__ uint32_t lds[kWavesPerBlock + 1];
    const uint32_t count = ctr->matCount[MK_COUNT].v;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock) {
        uint32_t i = base + threadIdx.x;
        bool found = false;
        uint32_t p = 0;
        if (i < count) {
            p = queue[i];
            found = resumePending(P, D, I, p, stDim(__float_as_uint(P.rayD[p].w)));
        }
        uint32_t s = blockReserve(found, &ctr->extCount[nextIdx].v, lds);
        if (found) nextQueue[s] = p;
    }
}

// VisibilityTester::Unoccluded for the shadow rays of one bounce (any-hit; unoccluded rays add their
// contribution to the path's L).  Same ray-pool scheme as k_intersect_closest.
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
__global__ void __launch_bounds__(kBlock, LDS_SCENE ? ldsSceneBlocks(LDSN) : 6) k_shadow(DevScene sc, DevPaths P, DevRayQueue shq, DevCounters* ctr, uint32_t kRefillIdle) {
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, LDS_SCENE, SPH>::Any;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    __shared__ uint32_t poolNext;
    __shared__ uint32_t red[2 * kWavesPerBlock];
    extern __shared__ float4 sceneLds[];
    uint32_t spill[spillWords(OVF)];
    const uint32_t count = ctr->shadowCount.v;
    if (blockIdx.x * kTraceChunk >= count) return;
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    if (LDS_SCENE) { stageSceneToLds(sc, sceneLds); nodes = sceneLds; tris = sceneLds + sc.nNodes * kLdsNodePitch; }
    blockZeroVisits(stack + threadIdx.x, Walker::kRows, kBlock);
    uint32_t nodeSteps = 0, triSteps = 0;
    for (uint32_t chunk = blockIdx.x * kTraceChunk; chunk < count; chunk += gridDim.x * kTraceChunk) {
        const uint32_t chunkCount = min(kTraceChunk, count - chunk);
        if (threadIdx.x == 0) poolNext = 0;
        __syncthreads();
        Walker w(stack + threadIdx.x, spill, SphereRefs(sc));
        uint32_t p = kNoRay;
        F3 contrib(0.f); // fetched with the ray: no dependent load when the ray is through
        bool exhausted = false;
        for (;;) {
            bool idle = w.done();
            unsigned long long idleMask = __ballot(idle);
            uint32_t nIdle = __popcll(idleMask);
            if (nIdle == 64u || (!exhausted && nIdle >= kRefillIdle)) {
                if (idle && p != kNoRay) {
                    if (!w.occluded) {
                        if (shq.perPath == 1) { // the path's only ray in this kernel: nobody else touches its L
                            float4 L = P.L[p];
                            P.L[p] = make_float4(L.x + contrib.x, L.y + contrib.y, L.z + contrib.z, 0.f);
                        } else {
                            float* L = reinterpret_cast<float*>(P.L + p / shq.perPath);
                            atomicAdd(L + 0, contrib.x); atomicAdd(L + 1, contrib.y); atomicAdd(L + 2, contrib.z);
                        }
                    }
                    p = kNoRay;
                }
                if (!exhausted) {
                    uint32_t base;
                    uint32_t mine = poolTake(&poolNext, idleMask, &base);
                    if (idle && mine < chunkCount) {
                        p = shq.idx[chunk + mine];
                        const size_t at = shq.dense ? (size_t)(chunk + mine) : (size_t)p;
                        float4 a = shq.o[at], b = shq.d[at];
                        contrib = F3(b.z, b.w, shq.wz[at]);
                        w.start(F3(a.x, a.y, a.z), F3(a.w, b.x, b.y), 1 - kShadowEpsilon);
                    }
                    exhausted = base + nIdle >= chunkCount;
                }
                if (__ballot(!w.done()) == 0) break;
            }
            w.template advance<LDS_SCENE>(nodes, tris); // any-hit rays of a scene in LDS: whole rounds (9 % faster than voted steps there)
        }
        nodeSteps += w.nodeSteps; triSteps += w.triSteps;
        __syncthreads(); // poolNext is rewritten by the next chunk
    }
    blockFlushVisits(stack + threadIdx.x, Walker::kRows, kBlock, sc.visits, kVisitShadow, red);
    waveFlushSteps(sc.visits, kVisitShadow, nodeSteps, triSteps);
}

// Rays a wave reserves from the shadow queue at a time.
constexpr uint32_t kShadowRun = 512;
template <int LDSN, int OVF, bool SPH = false>
__global__ void __launch_bounds__(kBlock, 6) k_shadow_stream(DevScene sc, DevPaths P, DevRayQueue shq, DevCounters* ctr, uint32_t kRefillIdle) {
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, false, SPH, false>::Any;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    uint32_t spill[spillWords(OVF)];
    const uint32_t count = ctr->shadowCount.v;
    if (blockIdx.x * kBlock >= count) return; // more blocks than rays
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    // Every wave streams on its own: it reserves runs of kShadowRun queue positions from the queue's cursor and hands them
    // to its lanes as they fall idle, until the cursor has passed the end of the queue.  No wave waits for another, and
    // lanes run dry only once, at the end of the launch.
    const uint32_t lane = threadIdx.x & 63u;
    Walker w(stack + threadIdx.x, spill, SphereRefs(sc), stackFullCounter(sc.visits, kVisitShadow));
    uint32_t p = kNoRay;
    F3 contrib(0.f); // fetched with the ray: no dependent load when the ray is through
    uint32_t runLo = 0, runHi = 0;
    bool exhausted = false;
    for (;;) {
        bool idle = w.done();
        unsigned long long idleMask = __ballot(idle);
        uint32_t nIdle = __popcll(idleMask);
        if (nIdle == 64u || (!exhausted && nIdle >= kRefillIdle)) {
            if (idle && p != kNoRay) {
                if (!w.occluded) {
                    if (shq.perPath == 1) { // the path's only ray in this kernel: nobody else touches its L
                        float4 L = P.L[p];
                        P.L[p] = make_float4(L.x + contrib.x, L.y + contrib.y, L.z + contrib.z, 0.f);
                    } else {
                        float* L = reinterpret_cast<float*>(P.L + p / shq.perPath);
                        atomicAdd(L + 0, contrib.x); atomicAdd(L + 1, contrib.y); atomicAdd(L + 2, contrib.z);
                    }
                }
                p = kNoRay;
            }
            if (!exhausted && runLo == runHi) {
                uint32_t r = 0;
                if (lane == 0) r = atomicAdd(&ctr->shadowCursor.v, kShadowRun);
                r = __shfl(r, 0);
                if (r >= count) exhausted = true;
                else { runLo = r; runHi = min(r + kShadowRun, count); }
            }
            if (!exhausted) {
                const uint32_t rank = __popcll(idleMask & ((1ull << lane) - 1ull));
                const uint32_t take = min(nIdle, runHi - runLo);
                if (idle && rank < take) {
                    const uint32_t at = runLo + rank;
                    p = shq.idx[at];
                    const size_t rec = shq.dense ? (size_t)at : (size_t)p;
                    float4 a = shq.o[rec], b = shq.d[rec];
                    contrib = F3(b.z, b.w, shq.wz[rec]);
                    w.start(F3(a.x, a.y, a.z), F3(a.w, b.x, b.y), 1 - kShadowEpsilon);
                }
                runLo += take;
            }
            if (__ballot(!w.done()) == 0) {
                if (exhausted) break;
                continue; // the run ended exactly here: reserve the next one
            }
        }
        w.step(nodes, tris);
    }
    waveFlushVisits(sc.visits, kVisitShadow, w.nodeVisits, w.triVisits);
    waveFlushSteps(sc.visits, kVisitShadow, w.nodeSteps, w.triSteps);
}

// The BSDF-sampled MIS rays of EstimateDirect (Enigma/Integrator.cpp:168-196): the closest hit must be
// the sampled light's triangle.  Few rays per bounce (only those that hit the light's own triangle
// are queued), one per lane.
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
__global__ void __launch_bounds__(kBlock) k_mis(DevScene sc, DevPaths P, DevRayQueue misq, DevCounters* ctr) {
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, LDS_SCENE, SPH>::Closest;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    __shared__ uint32_t red[2 * kWavesPerBlock];
    extern __shared__ float4 sceneLds[];
    const uint32_t count = ctr->misCount.v;
    if (blockIdx.x * kBlock >= count) return;
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    if (LDS_SCENE) { stageSceneToLds(sc, sceneLds); nodes = sceneLds; tris = sceneLds + sc.nNodes * kLdsNodePitch; }
    blockZeroVisits(stack + threadIdx.x, Walker::kRows, kBlock);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
        uint32_t p = misq.idx[i];
        float4 o = misq.o[p], d = misq.d[p];
        Hit h = traceClosest<LDSN, OVF, kBlock, LDS_SCENE, SPH>(nodes, tris, xyz(o), xyz(d), EDEV_INF, stack + threadIdx.x, SphereRefs(sc));
        {
            if (isSpherePrim<SPH>(sc, h.tri)) { // lightIsect.Le(-wi) of an emissive sphere, if it is the sampled light
                const DevSphere& sp = sc.spheres[h.tri - (int)sc.nTris];
                if ((int)sp.light - 1 != __float_as_int(d.w)) continue;
                const DevLight& lt = sc.lights[sp.light - 1];
                F3 ps, pe, n, dpdu, wo;
                sphereSurface(sp, F3(h.b0, h.b1, h.b2), xyz(d), &ps, &pe, &n, &dpdu, &wo);
                if (lt.twoSided || dot(n, -xyz(d)) > 0) {
                    F3 add = xyz(misq.w[p]) * F3(lt.L[0], lt.L[1], lt.L[2]);
                    float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
                    atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
                }
                continue;
            }
        }
        if (h.tri < 0) { // Li = light.Le(ray): only an infinite area light returns anything (Enigma/Integrator.cpp:194-195)
            const DevLight& lt = sc.lights[__float_as_int(d.w)];
            if (lt.type == 4) {
                F3 add = xyz(misq.w[p]) * infiniteLe(sc, lt, xyz(d));
                float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
                atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
            }
            continue;
        }
        const float4* tp = tris + (size_t)h.tri * 3;
        float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
        int lightIdx = __float_as_int(v2.w);
        if (lightIdx != __float_as_int(d.w)) continue; // lightIsect.m_primitive->getAreaLight() == &light
        const DevLight& lt = sc.lights[lightIdx];
        F3 n = triNormal(xyz(v0), xyz(v1), xyz(v2), triFlip(__float_as_uint(v1.w)));
        if (lt.twoSided || dot(n, -xyz(d)) > 0) { // lightIsect.Le(-wi)
            float4 c = misq.w[p];
            F3 add = xyz(c) * F3(lt.L[0], lt.L[1], lt.L[2]);
            float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
            atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
        }
    }
    blockFlushVisits(stack + threadIdx.x, Walker::kRows, kBlock, sc.visits, kVisitMis, red); // every thread gets here
}

// FilmTile::AddSample footprint (Enigma/Film.cpp:263-300).  SINK(x, y, weight) per covered pixel.
template <class SINK>
DEV void filmFootprint(const DevFilm& F, float fx, float fy, SINK sink) {
    float pdx = fx - 0.5f, pdy = fy - 0.5f;
    int p0x = (int)ceilf(pdx - F.radiusX), p0y = (int)ceilf(pdy - F.radiusY);
    int p1x = (int)floorf(pdx + F.radiusX) + 1, p1y = (int)floorf(pdy + F.radiusY) + 1;
    p0x = max(p0x, F.x0); p0y = max(p0y, F.y0);
    p1x = min(p1x, F.x1); p1y = min(p1y, F.y1);
    for (int y = p0y; y < p1y; ++y) {
        float fyv = fabsf(((float)y - pdy) * F.invRadiusY * 16.f);
        int iy = min((int)floorf(fyv), 15);
        for (int x = p0x; x < p1x; ++x) {
            float fxv = fabsf(((float)x - pdx) * F.invRadiusX * 16.f);
            int ix = min((int)floorf(fxv), 15);
            sink(x, y, F.filterTable[iy * 16 + ix]);
        }
    }
}

// One thread per pixel of the render region walks the pass's sample numbers of that pixel (path
// slot * nPixels + pixel).  Contributions that land on the thread's own pixel -- all of them for the
// default half-pixel box filter, bar samples exactly on a pixel edge -- are summed in registers and
// added once; the rest go out as atomics.  The film sees one atomic set per pixel and pass instead of
// one per sample.
__global__ void __launch_bounds__(kBlock) k_film_splat(DevPaths P, DevFilm F, DevIntegrator I, uint32_t nPixels, uint32_t nSlots) {
    uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= nPixels) return;
    uint32_t tx, ty;
    pixelOfIndex(I, pix, &tx, &ty);
    const int ownX = I.rx0 + (int)tx, ownY = I.ry0 + (int)ty;
    const int width = F.x1 - F.x0;
    F3 acc(0.f);
    float accW = 0.f;
    bool any = false;
    for (uint32_t s = 0; s < nSlots; ++s) {
        size_t i = (size_t)s * nPixels + pix;
        F3 L = xyz(P.L[i]);
        // Enigma/Integrator.cpp:284-292
        float y = luminance(L);
        if (isnan(L.x) || isnan(L.y) || isnan(L.z)) L = F3(0.f);
        else if ((double)y < -1e-5) L = F3(0.f);
        else if (isinf(y)) L = F3(0.f);
        float2 pf = P.pfilm[i];
        float ly = luminance(L);
        if (ly > F.maxSampleLuminance) L = L * (F.maxSampleLuminance / ly);
        filmFootprint(F, pf.x, pf.y, [&](int x, int yy, float w) {
            F3 c = L * 1.f * w; // L * sampleWeight * filterWeight
            if (x == ownX && yy == ownY) { acc += c; accW += w; any = true; }
            else {
                float* px = reinterpret_cast<float*>(F.pixels + ((size_t)(yy - F.y0) * width + (x - F.x0)));
                atomicAdd(px + 0, c.x); atomicAdd(px + 1, c.y); atomicAdd(px + 2, c.z); atomicAdd(px + 3, w);
            }
        });
    }
    if (any && ownX >= F.x0 && ownX < F.x1 && ownY >= F.y0 && ownY < F.y1) {
        float* px = reinterpret_cast<float*>(F.pixels + ((size_t)(ownY - F.y0) * width + (ownX - F.x0)));
        atomicAdd(px + 0, acc.x); atomicAdd(px + 1, acc.y); atomicAdd(px + 2, acc.z); atomicAdd(px + 3, accW);
    }
}

// Between stages: account rays, reset the queues the next stage fills.
// phase 0: before a bounce's k_intersect_closest.  phase 1: after the shade kernels.  phase 3: only the shadow queue's cursor.
__global__ void k_stage(DevCounters* ctr, int phase, uint32_t curIdx, uint32_t countImm, int useImm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (phase == 0) {
        ctr->raysClosest += useImm ? countImm : ctr->extCount[curIdx].v;
        for (int k = 0; k < kQueueKinds; ++k) { ctr->shaded += ctr->matCount[k].v; ctr->matCount[k].v = 0; } // what the step before this one shaded
        ctr->extCount[curIdx ^ 1u].v = 0;
        ctr->shadowCount.v = 0; ctr->misCount.v = 0;
    } else if (phase == 1) {
        ctr->raysShadow += ctr->shadowCount.v;
        ctr->raysMis += ctr->misCount.v;
        ctr->shadowCursor.v = 0;
    } else if (phase == 3) { // before a streaming shadow kernel that counts its rays itself (VolPathIntegrator)
        ctr->shadowCursor.v = 0;
    }
}

// ---------------------------------------------------------------- debug taps
template <bool HAL>
__global__ void k_debug_sobol(DevSampler S, uint32_t n, const int32_t* pixels, const uint32_t* slots, uint32_t nDims, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint32_t d = 0; d < nDims; ++d) out[(size_t)i * nDims + d] = sampleDimension<HAL>(S, (uint32_t)(pixels[2 * i] - S.sbx0), (uint32_t)(pixels[2 * i + 1] - S.sby0), slots[i], d);
}
template <bool HAL>
__global__ void k_debug_camera(DevSampler S, DevCamera C, uint32_t n, const int32_t* pixels, const uint32_t* slots, float* outRay, float* outPfilm) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px = pixels[2 * i], py = pixels[2 * i + 1];
    uint32_t relx = (uint32_t)(px - S.sbx0), rely = (uint32_t)(py - S.sby0);
    uint32_t slot = slots[i];
    float fx = (float)px + sampleDimension<HAL>(S, relx, rely, slot, 0);
    float fy = (float)py + sampleDimension<HAL>(S, relx, rely, slot, 1);
    float lu = 0.f, lv = 0.f;
    if (C.lensRadius > 0.f) { lu = sampleDimension<HAL>(S, relx, rely, slot, 3); lv = sampleDimension<HAL>(S, relx, rely, slot, 4); }
    F3 o, d;
    cameraRay(C, fx, fy, lu, lv, &o, &d);
    float* r = outRay + 6 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    outPfilm[2 * (size_t)i] = fx; outPfilm[2 * (size_t)i + 1] = fy;
}
template <int LDSN, int OVF, bool SPH = false>
__global__ void __launch_bounds__(kBlock) k_debug_intersect(DevScene sc, uint32_t n, const float* rays7, int anyHit, int32_t* triOut,
                                                            uint32_t* primOut, float* tOut, float* baryOut) {
    __shared__ uint32_t stack[WalkerSel<LDSN, OVF, kBlock, false, SPH>::Closest::kWords * kBlock]; // the walkers tally into the last two rows
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* r = rays7 + 7 * (size_t)i;
    F3 o(r[0], r[1], r[2]), d(r[3], r[4], r[5]);
    if (anyHit) {
        triOut[i] = traceAny<LDSN, OVF, kBlock, SPH>(sc.nodes, sc.tris, o, d, r[6], stack + threadIdx.x, SphereRefs(sc)) ? 1 : 0;
        return;
    }
    Hit h = traceClosest<LDSN, OVF, kBlock, false, SPH>(sc.nodes, sc.tris, o, d, r[6], stack + threadIdx.x, SphereRefs(sc));
    int orig = -1; // the primitive's number in the caller's arrays: triangles first, then spheres
    if (isSpherePrim<SPH>(sc, h.tri)) orig = h.tri - (int)sc.nTris + (int)(sc.nTris - sc.nSpheres);
    else if (h.tri >= 0) orig = __float_as_int(sc.tris[(size_t)h.tri * 3].w);
    triOut[i] = orig;
    primOut[i] = orig >= 0 ? sc.primId[orig] : 0u;
    tOut[i] = h.tri >= 0 ? h.t : EDEV_INF;
    baryOut[3 * (size_t)i] = h.b0; baryOut[3 * (size_t)i + 1] = h.b1; baryOut[3 * (size_t)i + 2] = h.b2;
}
// The hit records that the closest-hit kernel of a pass left for the camera rays of one sample number (path index =
// pixel number), brought into row-major pixel order over the render region and into the caller's primitive numbering.
__global__ void __launch_bounds__(kBlock) k_debug_gather_primary(DevScene sc, DevPaths P, DevIntegrator I, uint32_t nPixels, int32_t* triOut,
                                                                 uint32_t* primOut, float* tOut, float* baryOut) {
    const uint32_t pix = blockIdx.x * kBlock + threadIdx.x;
    if (pix >= nPixels) return;
    uint32_t x, y;
    pixelOfIndex(I, pix, &x, &y);
    const size_t o = (size_t)y * (uint32_t)I.rw + x;
    const float t = sc.hitT[pix];
    int orig = -1;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (t != EDEV_INF) { // a miss leaves the hit record of the path unwritten
        const float4 hv = P.hit[pix];
        const int tri = __float_as_int(hv.x);
        b0 = hv.y; b1 = hv.z; b2 = hv.w;
        if (sc.nSpheres && tri >= (int)sc.nTris) orig = tri - (int)sc.nTris + (int)(sc.nTris - sc.nSpheres);
        else orig = __float_as_int(sc.tris[(size_t)tri * 3].w);
    }
    triOut[o] = orig;
    primOut[o] = orig >= 0 ? sc.primId[orig] : 0u;
    tOut[o] = t;
    baryOut[3 * o] = b0; baryOut[3 * o + 1] = b1; baryOut[3 * o + 2] = b2;
}
// BSDF::f / Pdf / sample_f (Enigma/BxDF.cpp:885-985) of one material at a surface whose geometric and shading normal
// are +z and whose dpdu is +x, for n (wo, wi, u) triples: out = f.rgb, Pdf, sampled f.rgb, sampled pdf, sampled wi.xyz, sampled type
template <int MK>
__global__ void k_debug_bsdf(DevMaterial m, uint32_t n, const float* wo3, const float* wi3, const float* u2, float* out12, const DevLobe* lobes = nullptr,
                             bool multipleLobes = true) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename BsdfOf<MK>::T b;
    makeBsdf<MK>(m, F3(0.f, 0.f, 1.f), F3(0.f, 0.f, 1.f), F3(1.f, 0.f, 0.f), &b, nullptr, lobes, multipleLobes);
    const F3 wo(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), wi(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
    float* o = out12 + 12 * (size_t)i;
    const F3 f = bsdfF<MK>(b, wo, wi, BSDF_ALL);
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = bsdfPdf<MK>(b, wo, wi, BSDF_ALL);
    F3 wis(0.f);
    float pdf = 0.f;
    int type = 0;
    const F3 fs = bsdfSample<MK>(b, wo, &wis, u2[2 * i], u2[2 * i + 1], &pdf, BSDF_ALL, &type);
    o[4] = fs.x; o[5] = fs.y; o[6] = fs.z; o[7] = pdf; o[8] = wis.x; o[9] = wis.y; o[10] = wis.z; o[11] = (float)type;
}
__global__ void k_debug_footprint(DevFilm F, uint32_t n, const float* pfilm, uint32_t maxPer, int32_t* pixOut, float* wOut, uint32_t* countOut) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
    filmFootprint(F, pfilm[2 * i], pfilm[2 * i + 1], [&](int x, int y, float w) {
        if (c < maxPer) {
            pixOut[((size_t)i * maxPer + c) * 2] = x; pixOut[((size_t)i * maxPer + c) * 2 + 1] = y;
            wOut[(size_t)i * maxPer + c] = w;
        }
        ++c;
    });
    countOut[i] = c;
}

} // namespace edev
