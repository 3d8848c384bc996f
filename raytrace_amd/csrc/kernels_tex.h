// Image textures bound to material colours: ImageTexture<RGBSpectrum, Spectrum>::Evaluate over a UVMapping2D
// (Enigma/Texture.h:405-415, Enigma/Texture.cpp:22-29) with the trilinear MipMap lookup (Enigma/MipMap.h:309-376) and the
// uv differentials of SurfaceInteraction::ComputeDifferentials (Enigma/Interaction.cpp:161-209).  Only textures made with
// "trilinear" reach the device: the reference's EWA path always lands on the coarsest level, so those textures are
// constants and the host folds them into the material.
//
// k_texture runs between the closest-hit kernel and the shade kernels, over the material queues that hold textured
// materials, and leaves the evaluated colours of each path in DevScene::texColors (two float4 per path); the shade
// kernels swap them in for the material's constants.  Keeping the lookup out of the shade kernels keeps their register
// budget what it is for untextured scenes.
#pragma once
#include "kernels.h"

namespace edev {

DEV int texLevelW(const DevTexture& t, int l) { return max(1, (int)(t.w >> l)); }
DEV int texLevelH(const DevTexture& t, int l) { return max(1, (int)(t.h >> l)); }

// MipMap::texel, Enigma/MipMap.h:327-352 (repeat: int % unsigned)
DEV F3 texTexel(const DevTexture& t, int l, int s, int tt) {
    const int lw = texLevelW(t, l), lh = texLevelH(t, l);
    if (t.wrap == 2u) { s = min(max(s, 0), lw - 1); tt = min(max(tt, 0), lh - 1); }
    else if (t.wrap == 1u) { if (s < 0 || s >= lw || tt < 0 || tt >= lh) return F3(0.f); }
    const uint32_t us = (uint32_t)s % (uint32_t)lw, ut = (uint32_t)tt % (uint32_t)lh;
    return xyz(t.texels[t.levelOff[l] + (size_t)ut * lw + us]);
}
// MipMap::triangle, Enigma/MipMap.h:309-322
DEV F3 texTriangle(const DevTexture& t, int l, float s0f, float t0f) {
    l = min(max(l, 0), (int)t.nLevels - 1);
    float s = s0f * texLevelW(t, l) - 0.5f, tt = t0f * texLevelH(t, l) - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(tt);
    float ds = s - s0, dt = tt - t0;
    return (1 - ds) * (1 - dt) * texTexel(t, l, s0, t0) + (1 - ds) * dt * texTexel(t, l, s0, t0 + 1) + ds * (1 - dt) * texTexel(t, l, s0 + 1, t0) +
           ds * dt * texTexel(t, l, s0 + 1, t0 + 1);
}
// MipMap::lookup(st, width), Enigma/MipMap.h:354-367; Log2 = logf(x) * (1 / logf(2)), Enigma/MathCommon.h:117-120
DEV F3 texLookup(const DevTexture& t, float s, float tt, float width) {
    const float invLog2 = 1.f / logf(2.f);
    float level = (float)t.nLevels - 1.0f + logf(smax(width, 1e-8f)) * invLog2;
    if (level < 0) return texTriangle(t, 0, s, tt);
    else if (level >= (float)(t.nLevels - 1u)) return texTexel(t, (int)t.nLevels - 1, 0, 0);
    int iLevel = (int)floorf(level);
    float delta = level - iLevel;
    return (1.f - delta) * texTriangle(t, iLevel, s, tt) + delta * texTriangle(t, iLevel + 1, s, tt);
}

// SolveLinearSystem2x2, Enigma/Transform.cpp:190-200
DEV bool solve2x2(float a00, float a01, float a10, float a11, float b0, float b1, float* x0, float* x1) {
    float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (a11 * b0 - a01 * b1) / det;
    *x1 = (a00 * b1 - a10 * b0) / det;
    if (isnan(*x0) || isnan(*x1)) return false;
    return true;
}
DEV float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

template <bool HAL>
__global__ void __launch_bounds__(kBlock) k_texture(DevScene sc, DevPaths P, DevSampler S, DevIntegrator I, DevCamera C, float diffScale,
                                                    const uint32_t* matQueue, int kind, DevCounters* ctr) {
    const uint32_t count = ctr->matCount[kind].v;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
        const uint32_t p = matQueue[i];
        const float4 hv = P.hit[p];
        const int tri = __float_as_int(hv.x);
        F3 pHit, n, dpdu(0.f), dpdv(0.f);
        float uHit, vHit;
        uint32_t tex0, tex1;
        const float4 rdv = P.rayD[p];
        if (sc.nSpheres && tri >= (int)sc.nTris) { // Sphere::intersect's parametrisation (Enigma/ShapeImp.cpp:94-107)
            const DevSphere& sp = sc.spheres[tri - (int)sc.nTris];
            const DevMaterial& m = sc.materials[triMaterial(sp.mw)];
            if (!(m.tex0 | m.tex1)) continue;
            tex0 = m.tex0; tex1 = m.tex1;
            F3 pErr, wo;
            sphereSurface(sp, F3(hv.y, hv.z, hv.w), xyz(rdv), &pHit, &pErr, &n, &dpdu, &wo, &dpdv, &uHit, &vHit);
        } else {
            const float b0 = hv.y, b1 = hv.z, b2 = hv.w;
            const float4* tp = sc.tris + (size_t)tri * 3;
            const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            const uint32_t mw = __float_as_uint(v1.w);
            const DevMaterial& m = sc.materials[triMaterial(mw)];
            if (!(m.tex0 | m.tex1)) continue;
            tex0 = m.tex0; tex1 = m.tex1;
            const F3 p0 = xyz(v0), p1 = xyz(v1), p2 = xyz(v2);
            // Triangle::intersect: uvs, dpdu / dpdv, hit point, geometric normal (Enigma/TriangleMesh.cpp:238-289)
            float u0 = 0.f, w0 = 0.f, u1 = 1.f, w1 = 0.f, u2 = 1.f, w2 = 1.f; // getUVs default
            const bool shaded = sc.triShade != nullptr;
            float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0;
            if (shaded && (triHasUvs(mw) || triHasNormals(mw))) {
                const float4* sp = sc.triShade + (size_t)tri * 4;
                s0 = sp[0]; s1 = sp[1]; s2 = sp[2];
                if (triHasUvs(mw)) { float4 vv = sp[3]; u0 = s0.w; u1 = s1.w; u2 = s2.w; w0 = vv.x; w1 = vv.y; w2 = vv.z; }
            }
            const F3 dp02 = p0 - p2, dp12 = p1 - p2;
            const float du02 = u0 - u2, dv02 = w0 - w2, du12 = u1 - u2, dv12 = w1 - w2;
            const float determinant = du02 * dv12 - dv02 * du12;
            const bool degenerateUV = fabsf(determinant) < 1e-8;
            if (!degenerateUV) {
                float invdet = 1 / determinant;
                dpdu = (dv12 * dp02 - dv02 * dp12) * invdet;
                dpdv = (-du12 * dp02 + du02 * dp12) * invdet;
            }
            if (degenerateUV || lengthSqr(cross(dpdu, dpdv)) == 0) {
                F3 ng = cross(p2 - p0, p1 - p0);
                coordinateSystem(normalize(ng), &dpdu, &dpdv);
            }
            pHit = b0 * p0 + b1 * p1 + b2 * p2;
            uHit = b0 * u0 + b1 * u1 + b2 * u2; vHit = b0 * w0 + b1 * w1 + b2 * w2;
            n = triNormal(p0, p1, p2, triFlip(mw));
            if (shaded && triHasNormals(mw)) { // the geometric normal faces the shading normal (SetShadingGeometry, Enigma/Interaction.cpp:139-141)
                F3 ns = b0 * xyz(s0) + b1 * xyz(s1) + b2 * xyz(s2);
                ns = lengthSqr(ns) > 0 ? normalize(ns) : n;
                F3 ss = normalize(dpdu);
                F3 ts = cross(ss, ns);
                if (lengthSqr(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
                else coordinateSystem(ns, &ss, &ts);
                if (triReverse(mw)) ts = -ts;
                if (dot(n, normalize(cross(ss, ts))) < 0.f) n = -n;
            }
        }
        // ComputeDifferentials: only the unmodified camera ray carries differentials (Enigma/Ray.cpp:44-47)
        float dudx = 0.f, dvdx = 0.f, dudy = 0.f, dvdy = 0.f;
        const uint32_t st = __float_as_uint(rdv.w);
        if (stBounces(st) == 0u && !stNoDifferentials(st)) {
            const F3 o = xyz(P.rayO[p]), d = xyz(rdv);
            const float2 pf = P.pfilm[p];
            // PerspectiveCamera::GenerateRayDifferential, Enigma/Camera.cpp:102-109,144-192
            const F3 origin0 = xfPoint(C.r2c, F3(0.f, 0.f, 0.f));
            const F3 dxCamera = xfPoint(C.r2c, F3(1.f, 0.f, 0.f)) - origin0, dyCamera = xfPoint(C.r2c, F3(0.f, 1.f, 0.f)) - origin0;
            const F3 Pcamera = xfPoint(C.r2c, F3(pf.x, pf.y, 0.f));
            F3 rxD, ryD;
            if (C.type == 1u) {
                // EnvironmentCamera: the rays one pixel further in x, and -- from the x that ++ and -- leave behind -- in y
                // (Camera::GenerateRayDifferential, Enigma/Camera.cpp:35-57)
                const float xs = pf.x + 1.f, xb = xs - 1.f;
                rxD = environmentDirection(C, xs, pf.y);
                ryD = environmentDirection(C, xb, pf.y + 1.f);
            } else if (C.lensRadius > 0.f) {
                uint32_t px, py, slot;
                pathPixel(I, S, p, &px, &py, &slot);
                float lx, ly;
                concentricSampleDisk(sampleDimension<HAL>(S, px, py, slot, 3), sampleDimension<HAL>(S, px, py, slot, 4), &lx, &ly);
                lx *= C.lensRadius; ly *= C.lensRadius;
                const F3 lens(lx, ly, 0.f);
                F3 dx = normalize(Pcamera + dxCamera);
                float ft = C.focalDistance / dx.z;
                rxD = normalize(F3(0.f) + (ft * dx) - lens);
                F3 dy = normalize(Pcamera + dyCamera);
                ft = C.focalDistance / dy.z;
                ryD = normalize(F3(0.f) + (ft * dy) - lens);
            } else {
                rxD = normalize(Pcamera + dxCamera);
                ryD = normalize(Pcamera + dyCamera);
            }
            rxD = xfVector(C.c2w, rxD); ryD = xfVector(C.c2w, ryD);
            // both auxiliary origins are the ray's own origin; scaleDifferentials (Enigma/Ray.cpp:60-66, Enigma/Integrator.cpp:276)
            const F3 rxO = o + (o - o) * diffScale, ryO = rxO;
            rxD = d + (rxD - d) * diffScale; ryD = d + (ryD - d) * diffScale;
            const float dd = dot(n, pHit);
            const float tx = -(dot(n, rxO) - dd) / dot(n, rxD);
            const float ty = -(dot(n, ryO) - dd) / dot(n, ryD);
            if (!(isinf(tx) || isnan(tx) || isinf(ty) || isnan(ty))) {
                const F3 pxp = rxO + tx * rxD, pyp = ryO + ty * ryD;
                int d0, d1;
                if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
                else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
                else { d0 = 0; d1 = 1; }
                const float a00 = comp(dpdu, d0), a01 = comp(dpdv, d0), a10 = comp(dpdu, d1), a11 = comp(dpdv, d1);
                if (!solve2x2(a00, a01, a10, a11, comp(pxp, d0) - comp(pHit, d0), comp(pxp, d1) - comp(pHit, d1), &dudx, &dvdx)) dudx = dvdx = 0.f;
                if (!solve2x2(a00, a01, a10, a11, comp(pyp, d0) - comp(pHit, d0), comp(pyp, d1) - comp(pHit, d1), &dudy, &dvdy)) dudy = dvdy = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t ti = k == 0 ? tex0 : tex1;
            if (!ti) continue;
            const DevTexture& t = sc.textures[ti - 1u];
            // UVMapping2D::map, then MipMap::lookup(st, dst0, dst1) with m_doTrilinear (Enigma/MipMap.h:369-376)
            const float width = smax(smax(fabsf(t.su * dudx), fabsf(t.sv * dvdx)), smax(fabsf(t.su * dudy), fabsf(t.sv * dvdy)));
            sc.texColors[(size_t)p * 2 + k] = mk4(texLookup(t, t.su * uHit + t.du, t.sv * vHit + t.dv, width), 0.f);
        }
    }
}

// ---------------------------------------------------------------- SpatialLightDistribution (Enigma/LightDist.cpp:116-266)
// The reference computes a voxel's table when a path first asks for it.  Here, between the closest-hit kernel of a
// bounce and the kernels that shade its hits: k_spatial_mark claims a row of the table for every voxel that one of
// the queued interactions lies in and that has none yet, k_spatial_contrib and k_spatial_finish fill the rows
// claimed.  A voxel's table depends on the voxel only, not on when it is made.
// Claims the voxels of the interactions in queues [firstKind, lastKind]: surface hits, and the medium interactions of
// the VolPathIntegrator's queue kQueueMedium.
__global__ void __launch_bounds__(kBlock) k_spatial_mark(DevScene sc, DevPaths P, MatQueues mq, int firstKind, int lastKind, const DevCounters* ctr) {
    for (int k = firstKind; k <= lastKind; ++k) {
        if (k == MK_NONE) continue; // material-less surfaces are passed through, never lit
        const uint32_t count = ctr->matCount[k].v;
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
            const uint32_t p = mq.q[k][i];
            const float4 rdv = P.rayD[p], hv = P.hit[p];
            F3 at;
            if (k == kQueueMedium) at = xyz(P.rayO[p]) + xyz(rdv) * hv.y; // as k_shade_medium places it
            else at = hitGeometry(sc, sc.tris, __float_as_int(hv.x), hv.y, hv.z, hv.w, xyz(rdv), p).p;
            const uint32_t voxel = spatialVoxel(sc, at);
            if (sc.spatialSlot[voxel] != kSpatialEmpty) continue;
            if (atomicCAS(&sc.spatialSlot[voxel], kSpatialEmpty, kSpatialClaimed) != kSpatialEmpty) continue;
            const uint32_t row = atomicAdd(&sc.spatialCtl[0], 1u);
            if (row < sc.spatialCapacity) { sc.spatialNew[atomicAdd(&sc.spatialCtl[1], 1u)] = voxel; sc.spatialSlot[voxel] = row; }
            else atomicAdd(&sc.spatialCtl[2], 1u); // the voxel stays claimed; the host checks this count before the pass's samples reach the film
        }
    }
}
// halton: RadicalInverse(b, i) for b = 0..4 and i = 0..127, [b * 128 + i]; one (claimed voxel, light) pair per thread and turn
__global__ void __launch_bounds__(kBlock) k_spatial_contrib(DevScene sc, const float* halton) {
    const uint32_t nl = (uint32_t)sc.nLights;
    const size_t work = (size_t)sc.spatialCtl[1] * nl;
    auto lerp1 = [](float a, float b, float t) { if (t < 0.f) return a; if (t > 1.f) return b; const float diff = (b - a) * t; return a + diff; }; // Enigma/MathCommon.h:31-39
    for (size_t id = (size_t)blockIdx.x * kBlock + threadIdx.x; id < work; id += (size_t)gridDim.x * kBlock) {
        const uint32_t voxel = sc.spatialNew[id / nl], j = (uint32_t)(id % nl);
        const int pz = (int)(voxel % (uint32_t)sc.nVoxels[2]), py = (int)((voxel / (uint32_t)sc.nVoxels[2]) % (uint32_t)sc.nVoxels[1]),
                  px = (int)(voxel / ((uint32_t)sc.nVoxels[2] * (uint32_t)sc.nVoxels[1]));
        float lo[3], hi[3];
        const int pi[3] = {px, py, pz};
        for (int c = 0; c < 3; ++c) {
            float a = lerp1(sc.boundMin[c], sc.boundMax[c], float(pi[c]) / float(sc.nVoxels[c]));
            float b = lerp1(sc.boundMin[c], sc.boundMax[c], float(pi[c] + 1) / float(sc.nVoxels[c]));
            lo[c] = smin(a, b); hi[c] = smax(a, b);
        }
        const DevLight& lt = sc.lights[j];
        float contrib = 0.f;
        for (int i = 0; i < 128; ++i) {
            F3 po(lerp1(lo[0], hi[0], halton[i]), lerp1(lo[1], hi[1], halton[128 + i]), lerp1(lo[2], hi[2], halton[256 + i]));
            LightSample s = sampleLight<true>(sc, lt, sc.tris, po, halton[384 + i], halton[512 + i]);
            if (s.pdf > 0) contrib += (0.212671f * s.Li.x + 0.715160f * s.Li.y + 0.072169f * s.Li.z) / s.pdf;
        }
        sc.spatialTable[(size_t)sc.spatialSlot[voxel] * (2 * (size_t)nl + 2) + j] = contrib;
    }
}
// one claimed voxel per thread: the sums run in the reference's order
__global__ void __launch_bounds__(kBlock) k_spatial_finish(DevScene sc) {
    const uint32_t nl = (uint32_t)sc.nLights, nNew = sc.spatialCtl[1];
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < nNew; v += gridDim.x * kBlock) {
        float* func = sc.spatialTable + (size_t)sc.spatialSlot[sc.spatialNew[v]] * (2 * (size_t)nl + 2);
        float* cdf = func + nl;
        float sumContrib = 0.f;
        for (uint32_t i = 0; i < nl; ++i) sumContrib += func[i];
        const float avgContrib = sumContrib / (float)(128u * nl); // the reference divides by a size_t product
        const float minContrib = (avgContrib > 0) ? (float)(.001 * (double)avgContrib) : 1.f;
        for (uint32_t i = 0; i < nl; ++i) func[i] = smax(func[i], minContrib);
        // Distribution1D (Enigma/MonteCarlo.h:17-38)
        cdf[0] = 0.f;
        for (uint32_t i = 1; i < nl + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)nl;
        const float funcInt = cdf[nl];
        if (funcInt == 0.f) for (uint32_t i = 1; i < nl + 1; ++i) cdf[i] = float(i) / float(nl);
        else for (uint32_t i = 1; i < nl + 1; ++i) cdf[i] /= funcInt;
        cdf[nl + 1] = funcInt;
    }
}
__global__ void k_spatial_filled(DevScene sc) { sc.spatialCtl[1] = 0u; }

} // namespace edev
