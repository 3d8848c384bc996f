// VolPathIntegrator kernels (Enigma/VolPathIntegrator.cpp:45-177) beside the PathIntegrator's:
//   k_intersect_closest<.., 2>  Scene::intersect + HomogeneousMedium::Sample (kernels.h)
//   k_shade<kind, .., true>     the surface branch (kernels.h)
//   k_shade_medium              the medium branch: direct lighting with the Henyey-Greenstein phase function and the
//                               next direction (Enigma/VolPathIntegrator.cpp:74-87, Enigma/MediumInterface.cpp:159-200)
//   k_interface                 surfaces without a material: continue through them (Enigma/VolPathIntegrator.cpp:105-111)
//   k_shadow_tr                 VisibilityTester::Tr (Enigma/Lights.cpp:718-740)
//   k_mis_tr                    Scene::intersectTr for the BSDF / phase sampled ray of EstimateDirect (Enigma/Scene.cpp:46-64)
#pragma once
#include "kernels.h"

namespace edev {

// PhaseHG (Enigma/MediumInterface.cpp:196-200)
DEV float phaseHG(float cosTheta, float g) {
    const float kInvFourPi = 1.0f / (kPi * 4.0f);
    const float denom = 1 + g * g + 2 * g * cosTheta;
    return kInvFourPi * (1 - g * g) / (denom * sqrtf(denom));
}
// HenyeyGreenstein::Sample_p (Enigma/MediumInterface.cpp:159-177): the direction is built around wo itself
DEV float hgSample(float g, F3 wo, F3* wi, float u0, float u1) {
    float cosTheta;
    if (fabsf(g) < 1e-3) cosTheta = 1 - 2 * u0;
    else {
        float sqrTerm = (1 - g * g) / (1 + g - 2 * g * u0);
        cosTheta = -(1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    float sinTheta = sqrtf(smax(0.f, 1 - cosTheta * cosTheta));
    float phi = 2 * kPi * u1;
    F3 v1, v2;
    coordinateSystem(wo, &v1, &v2);
    *wi = (sinTheta * cosf(phi) * v1) + (sinTheta * sinf(phi) * v2) + (cosTheta * wo); // SphericalDirection, Enigma/MathCommon.cpp:13-18
    return phaseHG(cosTheta, g);
}

// HomogeneousMedium::Tr (Enigma/Medium.cpp:78-81) of a ray of direction d and parametric length tMax
DEV F3 mediumTr(const DevMedium& m, F3 d, float tMax) {
    if (m.opaque != 0.f) return F3(0.f); // GridDensityMedium::Tr, Enigma/Medium.cpp:51-54
    float s = smin(tMax * length(d), EDEV_MAXF);
    return F3(expf(-m.sigmaT[0] * s), expf(-m.sigmaT[1] * s), expf(-m.sigmaT[2] * s));
}

// The medium branch for the paths of the medium-interaction queue.  A medium interaction has no normal and no
// error bounds: rays start at the point itself and stay in the interaction's medium.
template <bool LDS_SCENE, bool SPH = false, bool HAL = false>
__global__ void __launch_bounds__(kBlock) k_shade_medium(DevScene sc, DevPaths P, DevSampler S, DevIntegrator I, DevRayQueue shq, DevRayQueue misq,
                                                         const uint32_t* queue, uint32_t* nextQueue, uint32_t nextIdx, DevCounters* ctr) {
    __shared__ uint32_t lds[32];
    extern __shared__ float4 sceneLds[];
    const uint32_t count = ctr->matCount[kQueueMedium].v;
    if (blockIdx.x * kBlock >= count) return;
    ENIGMA_STAGE_SHADE_TABLES
    (void)materials;
    uint32_t parity = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock, parity ^= 16u) {
        uint32_t i = base + threadIdx.x;
        bool active = i < count;
        bool pushShadow = false, pushMis = false, pushNext = false;
        uint32_t p = 0;
        if (active) {
            p = queue[i];
            float4 rdv = P.rayD[p];
            F3 d = xyz(rdv);
            uint32_t st = __float_as_uint(rdv.w);
            uint32_t dim = stDim(st), bounces = stBounces(st), med = stMedium(st);
            float g = sc.media[med - 1u].g;
            F3 pm = xyz(P.rayO[p]) + d * P.hit[p].y; // ray.scale(t)
            F3 wo = -d;                               // MediumInteraction::m_wo, not normalised (Enigma/Interaction.cpp:33-41)
            uint32_t px, py, slot;
            pathPixel(I, S, p, &px, &py, &slot);
            float4 bv = P.beta[p];
            F3 beta = xyz(bv);
            float etaScale = bv.w;
            // ---- UniformSampleOneLight + EstimateDirect for a medium interaction (Enigma/Integrator.cpp:71-204) ----
            if (sc.nLights > 0) {
                float pick;
                const float* pickFunc = lightFunc;
                const float* pickCdf = lightCdf;
                float pickInt = sc.lightFuncInt;
                if (!LDS_SCENE && sc.spatialTable) { // lightDistrib->Lookup(mi.m_p), Enigma/VolPathIntegrator.cpp:76 (global-memory build only)
                    pickFunc = spatialLookup(sc, pm); pickCdf = pickFunc + sc.nLights; pickInt = pickFunc[2 * sc.nLights + 1];
                }
                int lightNum = sampleDiscrete(pickCdf, pickFunc, pickInt, sc.nLights, sampleDimension<HAL>(S, px, py, slot, dim), &pick);
                dim += 1;
                if (pick != 0) {
                    float ulx = sampleDimension<HAL>(S, px, py, slot, dim), uly = sampleDimension<HAL>(S, px, py, slot, dim + 1);
                    float usx = sampleDimension<HAL>(S, px, py, slot, dim + 2), usy = sampleDimension<HAL>(S, px, py, slot, dim + 3);
                    dim += 4;
                    const DevLight& lt = lights[lightNum];
                    LightSample ls = sampleLight<SPH>(sc, lt, tris, pm, ulx, uly);
                    if (ls.pdf > 0 && !isBlack(ls.Li)) {
                        float ph = phaseHG(dot(wo, ls.wi), g);
                        if (ph != 0.f) {
                            F3 f(ph, ph, ph);
                            F3 target = offsetRayOrigin(ls.lp, ls.lpErr, ls.ln, pm - ls.lp);
                            F3 Ld = isDeltaLight(lt.type) ? f * ls.Li / ls.pdf : f * ls.Li * powerHeuristic(ls.pdf, ph) / ls.pdf;
                            storeShadowRay(shq, p, pm, target - pm, beta * (Ld / pick));
                            shq.tgtP[p] = mk4(ls.lp, __uint_as_float(med));
                            shq.tgtE[p] = mk4(ls.lpErr, 0.f);
                            shq.tgtN[p] = mk4(ls.ln, 0.f);
                            pushShadow = true;
                        }
                    }
                    if (!isDeltaLight(lt.type)) {
                        F3 wi2(0.f);
                        float ph2 = hgSample(g, wo, &wi2, usx, usy);
                        if (ph2 != 0.f && ph2 > 0) {
                            float lightPdf2 = lt.type == 4 ? infinitePdfLi(sc, lt, wi2)
                                              : (SPH && lt.type == 5) ? spherePdfFrom(sc.spheres[lt.tri], pm, F3(0.f), F3(0.f), pm, wi2)
                                                                             : lightTrianglePdf(pm, pm, wi2, ls.q0, ls.q1, ls.q2, lt.area);
                            if (lightPdf2 != 0) {
                                float weight = powerHeuristic(ph2, lightPdf2);
                                misq.o[p] = mk4(pm, __uint_as_float(med));
                                misq.d[p] = mk4(wi2, __int_as_float(lightNum));
                                misq.w[p] = mk4(beta * (F3(ph2, ph2, ph2) * weight / ph2 / pick), 0.f);
                                pushMis = true;
                            }
                        }
                    }
                }
            }
            // ---- next direction (Enigma/VolPathIntegrator.cpp:83-86) and Russian roulette (:165-173) ----
            F3 wi(0.f);
            float u0 = sampleDimension<HAL>(S, px, py, slot, dim), u1 = sampleDimension<HAL>(S, px, py, slot, dim + 1);
            dim += 2;
            hgSample(g, wo, &wi, u0, u1);
            bool alive = true;
            F3 rrBeta = beta * etaScale;
            if (maxComponentValue(rrBeta) < I.rrThreshold && bounces > 3) {
                float q = smax(.05f, 1 - maxComponentValue(rrBeta));
                float ur = sampleDimension<HAL>(S, px, py, slot, dim);
                dim += 1;
                if (ur < q) alive = false;
                else beta = beta / (1 - q);
            }
            if (alive) {
                P.rayO[p] = mk4(pm, 0.f);
                P.rayD[p] = mk4(wi, __uint_as_float(packState(med, dim, bounces + 1, 0u)));
                P.beta[p] = mk4(beta, etaScale);
                pushNext = true;
            }
        }
        uint32_t sSh, sMi, sNx;
        blockReserve3(pushShadow, pushMis, pushNext, &ctr->shadowCount.v, &ctr->misCount.v, &ctr->extCount[nextIdx].v, lds + parity, &sSh, &sMi, &sNx);
        if (pushShadow) shq.idx[sSh] = p;
        if (pushMis) misq.idx[sMi] = p;
        if (pushNext) nextQueue[sNx] = p;
    }
}

// Paths at a surface without a material: the ray goes on in its direction from the far side of the surface, in
// the medium on that side; the vertex does not count as a bounce and consumes no samples.
// (Enigma/PathIntegrator.cpp:87-93, Enigma/VolPathIntegrator.cpp:105-111, Enigma/DirectLightingIntegrator.cpp:51-54: the
// recursion goes on at the same depth.)  kind: the queue these paths wait in.
__global__ void __launch_bounds__(kBlock) k_interface(DevScene sc, DevPaths P, const uint32_t* queue, int kind, uint32_t* nextQueue, uint32_t nextIdx, DevCounters* ctr) {
    __shared__ uint32_t lds[kWavesPerBlock + 1];
    const uint32_t count = ctr->matCount[kind].v;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock) {
        uint32_t i = base + threadIdx.x;
        bool active = i < count;
        uint32_t p = 0;
        if (active) {
            p = queue[i];
            float4 rdv = P.rayD[p];
            F3 d = xyz(rdv);
            float4 hv = P.hit[p];
            int tri = __float_as_int(hv.x);
            uint32_t st = __float_as_uint(rdv.w);
            HitGeom g = hitGeometry(sc, sc.tris, tri, hv.y, hv.z, hv.w, d, p);
            P.rayO[p] = mk4(offsetRayOrigin(g.p, g.pError, g.n, d), 0.f); // isect.SpawnRay(ray.m_dir)
            P.rayD[p] = mk4(d, __uint_as_float(packState(mediumBeyond(sc, tri, stMedium(st), g.n, d), stDim(st), stBounces(st), stSpec(st)) | kStNoDifferentials));
        }
        uint32_t s = blockReserve(active, &ctr->extCount[nextIdx].v, lds);
        if (active) nextQueue[s] = p;
    }
}

// VisibilityTester::Tr for the shadow rays of one bounce.  A shadow ray is a chain of segments towards the same
// end point p1: each segment is a closest-hit query; a surface with a material blocks the light, a surface
// without one ends the segment, the medium of the segment attenuates it, and the next segment starts on the
// far side of that surface (isect.SpawnRayTo(p1)).  Every query counts as a shadow ray.
// Every wave streams on its own, like k_shadow_stream: it reserves runs of kShadowRun queue positions from the queue's
// cursor and hands them to its lanes as their chains end, so a lane does not sit out the longest chain of its wave (one
// chain per lane and launch, the wave waiting for its last: 5.2 ms per launch on the 1.18 M-triangle fog scene).
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
__global__ void __launch_bounds__(kBlock) k_shadow_tr(DevScene sc, DevPaths P, DevRayQueue shq, DevCounters* ctr, uint32_t kRefillIdle) {
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, LDS_SCENE, SPH>::Closest;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    __shared__ uint32_t red[2 * kWavesPerBlock];
    extern __shared__ float4 sceneLds[];
    uint32_t spill[spillWords(OVF)];
    const uint32_t count = ctr->shadowCount.v;
    if (blockIdx.x * kBlock >= count) return; // more blocks than rays
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    if (LDS_SCENE) { stageSceneToLds(sc, sceneLds); nodes = sceneLds; tris = sceneLds + sc.nNodes * kLdsNodePitch; }
    blockZeroVisits(stack + threadIdx.x, Walker::kRows, kBlock);
    uint32_t nQueries = 0;
    const uint32_t lane = threadIdx.x & 63u;
    Walker w(stack + threadIdx.x, spill, SphereRefs(sc));
    bool active = false; // the lane is working on a chain
    uint32_t p = 0, med = 0;
    F3 o(0.f), d(0.f), contrib(0.f), Tr(1.f), lp(0.f), lpErr(0.f), ln(0.f);
    uint32_t runLo = 0, runHi = 0;
    bool exhausted = false;
    for (;;) {
        // Ends of segments are worked on together, like refills: what a lane does there (the medium's transmittance, the
        // boundary's geometry, the next segment's set-up with its three divisions) runs once per wave however few lanes
        // need it, so the wave waits until kRefillIdle lanes have a segment to end or a ray to fetch -- or nobody can step.
        const bool segmentEnd = active && w.done();
        const uint32_t nEnds = __popcll(__ballot(segmentEnd));
        const bool noneSteps = __ballot(active && !w.done()) == 0;
        const bool turn = noneSteps || nEnds + (exhausted ? 0u : (uint32_t)__popcll(__ballot(!active))) >= kRefillIdle;
        if (turn && segmentEnd) { // a segment is through: the chain ends here, or goes on behind the boundary
            Hit h = w.h;
            bool hit = h.tri >= 0;
            uint32_t kind = MK_NONE;
            if (hit) kind = triKind(primWord<SPH>(sc, tris, h.tri));
            if (hit && kind != MK_NONE) { Tr = F3(0.f); active = false; }
            else {
                if (med) Tr = Tr * mediumTr(sc.media[med - 1u], d, hit ? h.t : 1 - kShadowEpsilon);
                if (!hit) active = false;
                else { // isect.SpawnRayTo(p1), Enigma/Interaction.cpp:58-64
                    F3 gp, ge, gn;
                    boundaryGeometry<SPH>(sc, tris, h, d, p, &gp, &ge, &gn);
                    o = offsetRayOrigin(gp, ge, gn, lp - gp);
                    F3 target = offsetRayOrigin(lp, lpErr, ln, o - lp);
                    d = target - o;
                    med = mediumBeyond(sc, h.tri, med, gn, d);
                    w.start(o, d, 1 - kShadowEpsilon);
                    ++nQueries;
                }
            }
            if (!active && !isBlack(Tr)) {
                F3 add = contrib * Tr;
                float4 L = P.L[p]; // the path's only shadow ray in this launch
                P.L[p] = make_float4(L.x + add.x, L.y + add.y, L.z + add.z, 0.f);
            }
        }
        const unsigned long long idleMask = __ballot(!active);
        const uint32_t nIdle = __popcll(idleMask);
        if (nIdle == 64u || (turn && !exhausted && nIdle != 0u)) {
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
                if (!active && rank < take) {
                    p = shq.idx[runLo + rank];
                    float4 a = shq.o[p], b = shq.d[p], tp = shq.tgtP[p];
                    o = F3(a.x, a.y, a.z); d = F3(a.w, b.x, b.y);
                    contrib = F3(b.z, b.w, shq.wz[p]);
                    lp = xyz(tp); med = __float_as_uint(tp.w);
                    lpErr = xyz(shq.tgtE[p]); ln = xyz(shq.tgtN[p]);
                    Tr = F3(1.f);
                    w.start(o, d, 1 - kShadowEpsilon);
                    active = true;
                    ++nQueries;
                }
                runLo += take;
            }
            if (__ballot(active) == 0) {
                if (exhausted) break;
                continue; // the run ended exactly here: reserve the next one
            }
        }
        w.template advance<LDS_SCENE>(nodes, tris);
    }
    for (int off = 32; off > 0; off >>= 1) nQueries += __shfl_down(nQueries, off);
    if ((threadIdx.x & 63u) == 0 && nQueries) atomicAdd(&ctr->raysShadow, (unsigned long long)nQueries);
    blockFlushVisits(stack + threadIdx.x, Walker::kRows, kBlock, sc.visits, kVisitShadow, red);
}

// Scene::intersectTr for the MIS rays of one bounce: closest-hit queries through material-less surfaces
// (isect.SpawnRay(ray.m_dir)), attenuated by the media on the way; the first surface with a material must be
// the sampled light's triangle.  Every query counts as a MIS ray.
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
__global__ void __launch_bounds__(kBlock) k_mis_tr(DevScene sc, DevPaths P, DevRayQueue misq, DevCounters* ctr) {
    using Walker = typename WalkerSel<LDSN, OVF, kBlock, LDS_SCENE, SPH>::Closest;
    __shared__ uint32_t stack[Walker::kWords * kBlock];
    __shared__ uint32_t red[2 * kWavesPerBlock];
    extern __shared__ float4 sceneLds[];
    uint32_t spill[spillWords(OVF)];
    const uint32_t count = ctr->misCount.v;
    if (blockIdx.x * kBlock >= count) return;
    const float4* nodes = sc.nodes;
    const float4* tris = sc.tris;
    if (LDS_SCENE) { stageSceneToLds(sc, sceneLds); nodes = sceneLds; tris = sceneLds + sc.nNodes * kLdsNodePitch; }
    blockZeroVisits(stack + threadIdx.x, Walker::kRows, kBlock);
    uint32_t nQueries = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < count; base += gridDim.x * kBlock) {
        uint32_t i = base + threadIdx.x;
        bool active = i < count;
        Walker w(stack + threadIdx.x, spill, SphereRefs(sc));
        uint32_t p = 0, med = 0;
        int wantLight = -1;
        F3 o(0.f), d(0.f), Tr(1.f);
        if (active) {
            p = misq.idx[i];
            float4 ov = misq.o[p], dv = misq.d[p];
            o = xyz(ov); d = xyz(dv); med = __float_as_uint(ov.w); wantLight = __float_as_int(dv.w);
            w.start(o, d, EDEV_INF);
            ++nQueries;
        }
        while (__ballot(active) != 0) {
            if (active && w.done()) {
                Hit h = w.h;
                bool hit = h.tri >= 0;
                if (med) Tr = Tr * mediumTr(sc.media[med - 1u], d, hit ? h.t : EDEV_INF);
                if (!hit) {
                    active = false;
                    const DevLight& lt = sc.lights[wantLight];
                    if (lt.type == 4) { // Li = light.Le(ray)
                        F3 add = xyz(misq.w[p]) * infiniteLe(sc, lt, d) * Tr;
                        float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
                        atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
                    }
                } else if (isSpherePrim<SPH>(sc, h.tri) && triKind(sc.spheres[h.tri - (int)sc.nTris].mw) != MK_NONE) {
                    active = false; // a surface with a material: the chain ends here, with lightIsect.Le(-wi) if it is the sampled light
                    const DevSphere& sp = sc.spheres[h.tri - (int)sc.nTris];
                    if ((int)sp.light - 1 == wantLight) {
                        const DevLight& lt = sc.lights[wantLight];
                        F3 ps, pe, n, dpdu, wo;
                        sphereSurface(sp, F3(h.b0, h.b1, h.b2), d, &ps, &pe, &n, &dpdu, &wo);
                        if (lt.twoSided || dot(n, -d) > 0) {
                            F3 add = xyz(misq.w[p]) * F3(lt.L[0], lt.L[1], lt.L[2]) * Tr;
                            float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
                            atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
                        }
                    }
                } else {
                    uint32_t mw = primWord<SPH>(sc, tris, h.tri); // a triangle, or a sphere without a material
                    if (triKind(mw) != MK_NONE) {
                        const float4* tp = tris + (size_t)h.tri * 3;
                        float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                        active = false;
                        int lightIdx = __float_as_int(v2.w);
                        if (lightIdx == wantLight) { // lightIsect.m_primitive->getAreaLight() == &light
                            const DevLight& lt = sc.lights[lightIdx];
                            F3 n = triNormal(xyz(v0), xyz(v1), xyz(v2), triFlip(mw));
                            if (lt.twoSided || dot(n, -d) > 0) { // lightIsect.Le(-wi)
                                F3 add = xyz(misq.w[p]) * F3(lt.L[0], lt.L[1], lt.L[2]) * Tr;
                                float* L = reinterpret_cast<float*>(P.L + p / misq.perPath);
                                atomicAdd(L + 0, add.x); atomicAdd(L + 1, add.y); atomicAdd(L + 2, add.z);
                            }
                        }
                    } else {
                        F3 gp, ge, gn;
                        boundaryGeometry<SPH>(sc, tris, h, d, p, &gp, &ge, &gn);
                        o = offsetRayOrigin(gp, ge, gn, d);
                        med = mediumBeyond(sc, h.tri, med, gn, d);
                        w.start(o, d, EDEV_INF);
                        ++nQueries;
                    }
                }
            }
            w.template advance<LDS_SCENE>(nodes, tris);
        }
    }
    for (int off = 32; off > 0; off >>= 1) nQueries += __shfl_down(nQueries, off);
    if ((threadIdx.x & 63u) == 0 && nQueries) atomicAdd(&ctr->raysMis, (unsigned long long)nQueries);
    blockFlushVisits(stack + threadIdx.x, Walker::kRows, kBlock, sc.visits, kVisitMis, red);
}

} // namespace edev
