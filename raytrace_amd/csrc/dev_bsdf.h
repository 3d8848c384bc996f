// BSDF evaluation / sampling on the device.
//
// Replaces the arena-allocated BxDF object graph (Enigma/BxDF.cpp:843-985 BSDF;
// :729-732 Lambertian; :1075-1121 MicrofacetReflection; :1007-1073 MicrofacetTransmission;
// :1124-1158 FresnelSpecular; :453-503 Fresnel terms; Enigma/MicrofacetDist.cpp:165-328
// Trowbridge-Reitz) with a POD of at most two lobes held in registers; the lobe kinds are compile-time per material kind.  Formulas keep the
// reference's operation order and its deviations from pbrt:
//  * the local frame's z axis is the GEOMETRIC normal (BxDF.cpp:872-882);
//  * FresnelSpecular scales its reflection by T, not R (BxDF.cpp:1135);
//  * MicrofacetTransmission::pdf uses etaA/etaB where f uses etaB/etaA (BxDF.cpp:1027,1063).
#pragma once
#include "dev_types.h"

namespace edev {

enum : int { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum LobeKind : int { LOBE_LAMBERT, LOBE_MICRO_REFL_CONDUCTOR, LOBE_MICRO_REFL_DIELECTRIC, LOBE_MICRO_TRANS, LOBE_FRESNEL_SPECULAR, LOBE_SPEC_REFL, LOBE_SPEC_TRANS, LOBE_OREN_NAYAR,
                      // lobes of the MK_GENERAL materials only: LambertianTransmission (Enigma/BxDF.cpp:1160-1190), SpecularReflection over
                      // FresnelNoOp (the mirror, Enigma/Material.cpp:326-327), FresnelBlend (Enigma/BxDF.cpp:781-839)
                      LOBE_LAMBERT_TRANS, LOBE_SPEC_REFL_NOOP, LOBE_FRESNEL_BLEND };

DEV constexpr int lobeType(int k) {
    return (k == LOBE_LAMBERT || k == LOBE_OREN_NAYAR) ? (BSDF_REFLECTION | BSDF_DIFFUSE)
         : (k == LOBE_MICRO_REFL_CONDUCTOR || k == LOBE_MICRO_REFL_DIELECTRIC || k == LOBE_FRESNEL_BLEND) ? (BSDF_REFLECTION | BSDF_GLOSSY)
         : k == LOBE_MICRO_TRANS ? (BSDF_TRANSMISSION | BSDF_GLOSSY)
         : (k == LOBE_SPEC_REFL || k == LOBE_SPEC_REFL_NOOP) ? (BSDF_REFLECTION | BSDF_SPECULAR)
         : k == LOBE_SPEC_TRANS ? (BSDF_TRANSMISSION | BSDF_SPECULAR)
         : k == LOBE_LAMBERT_TRANS ? (BSDF_TRANSMISSION | BSDF_DIFFUSE)
         : (BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR);
}

// ---- BxDF.h:28-106 helpers ----
DEV float cosTheta(F3 w) { return w.z; }
DEV float absCosTheta(F3 w) { return fabsf(w.z); }
DEV float cos2Theta(F3 w) { return w.z * w.z; }
DEV float sin2Theta(F3 w) { return smax(0.f, 1.f - cos2Theta(w)); }
DEV float sinTheta(F3 w) { return sqrtf(sin2Theta(w)); }
DEV float tanTheta(F3 w) { return sinTheta(w) / cosTheta(w); }
DEV float tan2Theta(F3 w) { return sin2Theta(w) / cos2Theta(w); }
DEV float cosPhi(F3 w) { float s = sinTheta(w); return (s == 0.f) ? 1.f : clampf(w.x / s, -1.f, 1.f); }
DEV float sinPhi(F3 w) { float s = sinTheta(w); return (s == 0.f) ? 0.f : clampf(w.y / s, -1.f, 1.f); }
DEV float cos2Phi(F3 w) { float c = cosPhi(w); return c * c; }
DEV float sin2Phi(F3 w) { float s = sinPhi(w); return s * s; }
DEV bool sameHemisphere(F3 w, F3 wp) { return w.z * wp.z > 0.f; }
DEV F3 reflect(F3 wo, F3 n) { return -wo + 2 * dot(wo, n) * n; }
DEV bool refract(F3 wi, F3 n, float eta, F3* wt) {
    float cosThetaI = dot(n, wi);
    float sin2ThetaI = smax(0.f, 1 - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

// BxDF.cpp:453-475
DEV float frDielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    bool entering = cosThetaI > 0.f;
    if (!entering) { float t = etaI; etaI = etaT; etaT = t; cosThetaI = fabsf(cosThetaI); }
    float sinThetaI = sqrtf(smax(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1.0f;
    float cosThetaT = sqrtf(smax(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// BxDF.cpp:478-503 with etai = 1 (Material.cpp:131)
DEV F3 frConductor(float cosThetaI, F3 etat, F3 k) {
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    F3 eta = etat / F3(1.f), etak = k / F3(1.f);
    float cosThetaI2 = cosThetaI * cosThetaI;
    float sinThetaI2 = 1.f - cosThetaI2; // 1. - x in double then float == the float difference (exact in double)
    F3 eta2 = eta * eta, etak2 = etak * etak;
    F3 t0 = eta2 - etak2 - F3(sinThetaI2);
    F3 a2plusb2 = sqrt3(t0 * t0 + (eta2 * 4.f) * etak2);
    F3 t1 = a2plusb2 + F3(cosThetaI2);
    F3 a = sqrt3((a2plusb2 + t0) * 0.5f);
    F3 t2 = a * (2.f * cosThetaI);
    F3 Rs = (t1 - t2) / (t1 + t2);
    F3 t3 = a2plusb2 * cosThetaI2 + F3(sinThetaI2 * sinThetaI2);
    F3 t4 = t2 * sinThetaI2;
    F3 Rp = Rs * (t3 - t4) / (t3 + t4);
    return (Rp + Rs) * 0.5f;
}

// ---- Trowbridge-Reitz, MicrofacetDist.cpp:165-328 (sample visible area) ----
DEV float trD(float ax, float ay, F3 wh) {
    float t2 = tan2Theta(wh);
    if (isinf(t2)) return 0.f;
    float cos4Theta = cos2Theta(wh) * cos2Theta(wh);
    float e = (cos2Phi(wh) / (ax * ax) + sin2Phi(wh) / (ay * ay)) * t2;
    return 1 / (kPi * ax * ay * cos4Theta * (1 + e) * (1 + e));
}
DEV float trLambda(float ax, float ay, F3 w) {
    float absTanTheta = fabsf(tanTheta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(cos2Phi(w) * ax * ax + sin2Phi(w) * ay * ay);
    float a2t2 = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + a2t2)) / 2;
}
DEV float trG1(float ax, float ay, F3 w) { return 1 / (1 + trLambda(ax, ay, w)); }
DEV float trG(float ax, float ay, F3 wo, F3 wi) { return 1 / (1 + trLambda(ax, ay, wo) + trLambda(ax, ay, wi)); }
DEV float trPdf(float ax, float ay, F3 wo, F3 wh) { return trD(ax, ay, wh) * trG1(ax, ay, wo) * absdot(wo, wh) / absCosTheta(wo); }
DEV void trSample11(float cosT, float U1, float U2, float* slope_x, float* slope_y) {
    if (cosT > .9999f) { // the reference compares in double against .9999; same decision for every float
        float r = sqrtf(U1 / (1 - U1));
        float phi = (float)(6.28318530718 * (double)U2);
        *slope_x = r * cosf(phi);
        *slope_y = r * sinf(phi);
        return;
    }
    float sinT = sqrtf(smax(0.f, 1.f - cosT * cosT));
    float tanT = sinT / cosT;
    float a = 1 / tanT;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10f) tmp = 1e10f;
    float B = tanT;
    float D = sqrtf(smax(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanT) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
DEV F3 trSampleWh(float ax, float ay, F3 wo, float u0, float u1) {
    bool flip = wo.z < 0;
    F3 wi = flip ? -wo : wo;
    F3 wiS = normalize(F3(ax * wi.x, ay * wi.y, wi.z));
    float sx, sy;
    trSample11(cosTheta(wiS), u0, u1, &sx, &sy);
    float tmp = cosPhi(wiS) * sx - sinPhi(wiS) * sy;
    sy = sinPhi(wiS) * sx + cosPhi(wiS) * sy;
    sx = tmp;
    sx = ax * sx;
    sy = ay * sy;
    F3 wh = normalize(F3(-sx, -sy, 1.f));
    return flip ? -wh : wh;
}

// ---- MonteCarlo.cpp:50-68,156-161 ----
DEV void concentricSampleDisk(float ux, float uy, float* dx, float* dy) {
    float ox = 2.f * ux - 1, oy = 2.f * uy - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = kPiOver4 * (oy / ox); }
    else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
    *dx = r * cosf(theta); *dy = r * sinf(theta);
}
DEV F3 cosineSampleHemisphere(float ux, float uy) {
    float dx, dy;
    concentricSampleDisk(ux, uy, &dx, &dy);
    float z = sqrtf(smax(0.f, 1 - dx * dx - dy * dy));
    return F3(dx, dy, z);
}

// ---- the per-hit BSDF ----
// The lobes a material can own are fixed by its kind, in the order Material::computeScatteringFunctions
// adds them; which of them exist at a hit (non-black colours) is run-time.  Keeping the lobe kind a
// compile-time function of (material kind, slot) lets every switch below fold away per shade kernel
// and keeps the struct in registers (no run-time indexed members).
template <int MK, int I> struct LobeOf { static constexpr int kind = -1; };
template <> struct LobeOf<MK_MATTE, 0> { static constexpr int kind = LOBE_LAMBERT; };
template <> struct LobeOf<MK_MATTE_ROUGH, 0> { static constexpr int kind = LOBE_OREN_NAYAR; };
template <> struct LobeOf<MK_METAL, 0> { static constexpr int kind = LOBE_MICRO_REFL_CONDUCTOR; };
template <> struct LobeOf<MK_GLASS_SMOOTH, 0> { static constexpr int kind = LOBE_FRESNEL_SPECULAR; };
template <> struct LobeOf<MK_GLASS_ROUGH, 0> { static constexpr int kind = LOBE_MICRO_REFL_DIELECTRIC; };
template <> struct LobeOf<MK_GLASS_ROUGH, 1> { static constexpr int kind = LOBE_MICRO_TRANS; };
// smooth glass without allowMultipleLobes (DirectLightingIntegrator): two specular lobes, Material.cpp:204-218
template <> struct LobeOf<MK_DL_GLASS, 0> { static constexpr int kind = LOBE_SPEC_REFL; };
template <> struct LobeOf<MK_DL_GLASS, 1> { static constexpr int kind = LOBE_SPEC_TRANS; };
template <> struct LobeOf<MK_PLASTIC, 0> { static constexpr int kind = LOBE_LAMBERT; };
template <> struct LobeOf<MK_PLASTIC, 1> { static constexpr int kind = LOBE_MICRO_REFL_DIELECTRIC; };

struct Bsdf {
    bool has0, has1;       // slot present at this hit
    F3 col0, col1;         // R / T / Kd / Ks of the slot (glass smooth: col0 = R, col1 = T, one lobe)
    F3 cEta, cK;           // conductor eta, k
    float ax, ay;          // shared microfacet distribution of the material
    float eta;             // BSDF::m_eta (glass index, else 1)
    float dEtaI, dEtaT;    // dielectric Fresnel / transmission interface
    F3 ng, ss, ts;         // the reference's local frame (Enigma/BxDF.cpp:872-882): shading tangent, shading bitangent ns x ss, and the
                           // GEOMETRIC normal as its third axis (not the shading normal; not orthonormal when they differ)
};

template <class B> DEV F3 toLocal(const B& b, F3 v) { return F3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ng)); }
template <class B> DEV F3 toWorld(const B& b, F3 v) {
    return F3(b.ss.x * v.x + b.ts.x * v.y + b.ng.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ng.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ng.z * v.z);
}

// Material::computeScatteringFunctions for the four materials (Enigma/Material.cpp:76-93,115-136,176-220,232-255).
template <int MK>
DEV void makeBsdf(const DevMaterial& m, F3 ng, F3 ns, F3 dpdu, Bsdf* b, const float4* texColors = nullptr, const DevLobe* = nullptr, bool = true) {
    b->ng = ng;
    b->ss = normalize(dpdu);
    b->ts = cross(ns, b->ss);
    b->eta = 1.f;
    b->ax = m.alphax; b->ay = m.alphay;
    b->col0 = F3(m.c0[0], m.c0[1], m.c0[2]); b->col1 = F3(m.c1[0], m.c1[1], m.c1[2]);
    b->cEta = F3(0.f); b->cK = F3(0.f);
    b->dEtaI = 1.f; b->dEtaT = 1.f;
    bool nb0 = m.flags & 1u, nb1 = m.flags & 2u;
    if (texColors && (m.tex0 | m.tex1)) { // the texture's value at this hit replaces the constant; Spectrum::Clamp() for all but the metal's eta / k
        if (m.tex0) { float4 c = texColors[0]; b->col0 = MK == MK_METAL ? F3(c.x, c.y, c.z) : F3(smax(c.x, 0.f), smax(c.y, 0.f), smax(c.z, 0.f)); nb0 = b->col0.x != 0.f || b->col0.y != 0.f || b->col0.z != 0.f; }
        if (m.tex1) { float4 c = texColors[1]; b->col1 = MK == MK_METAL ? F3(c.x, c.y, c.z) : F3(smax(c.x, 0.f), smax(c.y, 0.f), smax(c.z, 0.f)); nb1 = b->col1.x != 0.f || b->col1.y != 0.f || b->col1.z != 0.f; }
    }
    b->has0 = nb0; b->has1 = false;
    if (MK == MK_METAL) {
        b->cEta = b->col0; b->cK = b->col1; b->col0 = F3(1.f); b->has0 = true;
    } else if (MK == MK_GLASS_SMOOTH) {
        b->eta = m.eta; b->dEtaT = m.eta;
        b->has0 = nb0 || nb1;
    } else if (MK == MK_GLASS_ROUGH || MK == MK_DL_GLASS) {
        b->eta = m.eta; b->dEtaT = m.eta;
        b->has1 = nb1;
    } else if (MK == MK_PLASTIC) { // FresnelDielectric(1.5f, 1.f), Material.cpp:244
        b->dEtaI = 1.5f; b->dEtaT = 1.f;
        b->has1 = nb1;
    }
}

DEV constexpr bool lobeMatches(int kind, int flags) { return kind >= 0 && (lobeType(kind) & flags) == lobeType(kind); } // BxDF::matchesFlags
// slot I of material kind MK exists at this hit and matches flags
template <int MK, int I>
DEV bool lobeOn(const Bsdf& b, int flags) { return lobeMatches(LobeOf<MK, I>::kind, flags) && (I == 0 ? b.has0 : b.has1); }
template <int MK>
DEV int numComponents(const Bsdf& b, int flags) { return (lobeOn<MK, 0>(b, flags) ? 1 : 0) + (lobeOn<MK, 1>(b, flags) ? 1 : 0); }

template <int KIND>
DEV F3 microReflF(const Bsdf& b, F3 col, F3 wo, F3 wi) { // BxDF.cpp:1084-1100
    float cosThetaO = absCosTheta(wo), cosThetaI = absCosTheta(wi);
    F3 wh = wi + wo;
    if (cosThetaI == 0 || cosThetaO == 0) return F3(0.f);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return F3(0.f);
    wh = normalize(wh);
    float c = dot(wi, faceForward(wh, F3(0, 0, 1)));
    F3 F = (KIND == LOBE_MICRO_REFL_CONDUCTOR) ? frConductor(fabsf(c), b.cEta, b.cK) : F3(frDielectric(c, b.dEtaI, b.dEtaT));
    return col * trD(b.ax, b.ay, wh) * trG(b.ax, b.ay, wo, wi) * F / (4 * cosThetaI * cosThetaO);
}
DEV float microReflPdf(const Bsdf& b, F3 wo, F3 wi) { // BxDF.cpp:1116-1121
    if (!sameHemisphere(wo, wi)) return 0;
    F3 wh = normalize(wo + wi);
    return trPdf(b.ax, b.ay, wo, wh) / (4 * dot(wo, wh));
}
DEV F3 microTransF(const Bsdf& b, F3 col, F3 wo, F3 wi) { // BxDF.cpp:1018-1043 (radiance transport)
    if (sameHemisphere(wo, wi)) return F3(0.f);
    float cosThetaO = cosTheta(wo), cosThetaI = cosTheta(wi);
    if (cosThetaI == 0 || cosThetaO == 0) return F3(0.f);
    float eta = cosTheta(wo) > 0 ? (b.dEtaT / b.dEtaI) : (b.dEtaI / b.dEtaT);
    F3 wh = normalize(wo + wi * eta);
    if (wh.z < 0) wh = -wh;
    if (dot(wo, wh) * dot(wi, wh) > 0) return F3(0.f);
    float F = frDielectric(dot(wo, wh), b.dEtaI, b.dEtaT);
    float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    float factor = 1 / eta;
    float s = fabsf(trD(b.ax, b.ay, wh) * trG(b.ax, b.ay, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                    (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    return (F3(1.f) - F3(F)) * col * s;
}
DEV float microTransPdf(const Bsdf& b, F3 wo, F3 wi) { // BxDF.cpp:1059-1073
    if (sameHemisphere(wo, wi)) return 0;
    float eta = cosTheta(wo) > 0 ? (b.dEtaI / b.dEtaT) : (b.dEtaT / b.dEtaI);
    F3 wh = normalize(wo + wi * eta);
    if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
    float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
    return trPdf(b.ax, b.ay, wo, wh) * dwh_dwi;
}

// OrenNayer::f, Enigma/BxDF.cpp:755-779; A and B (Enigma/BxDF.cpp:744-753) ride in the slots of the microfacet alphas
DEV F3 orenNayarF(const Bsdf& b, F3 col, F3 wo, F3 wi) {
    float sinThetai = sinTheta(wi), sinThetao = sinTheta(wo);
    float maxcos = 0.f;
    if (sinThetai > 1e-4 && sinThetao > 1e-4) {
        float dcos = cosPhi(wi) * cosPhi(wo) + sinPhi(wi) * sinPhi(wo);
        maxcos = smax(0.f, dcos);
    }
    float sinalpha, tanbeta;
    if (absCosTheta(wi) > absCosTheta(wo)) { sinalpha = sinThetao; tanbeta = sinThetai / absCosTheta(wi); }
    else { sinalpha = sinThetai; tanbeta = sinThetao / absCosTheta(wo); }
    return col * kInvPi * (b.ax + b.ay * maxcos * sinalpha * tanbeta);
}

template <int KIND>
DEV F3 lobeF(const Bsdf& b, F3 col, F3 wo, F3 wi) {
    if (KIND == LOBE_LAMBERT) return col * kInvPi;
    if (KIND == LOBE_OREN_NAYAR) return orenNayarF(b, col, wo, wi);
    if (KIND == LOBE_MICRO_REFL_CONDUCTOR || KIND == LOBE_MICRO_REFL_DIELECTRIC) return microReflF<KIND>(b, col, wo, wi);
    if (KIND == LOBE_MICRO_TRANS) return microTransF(b, col, wo, wi);
    return F3(0.f);
}
template <int KIND>
DEV float lobePdf(const Bsdf& b, F3 wo, F3 wi) {
    if (KIND == LOBE_LAMBERT || KIND == LOBE_OREN_NAYAR) return sameHemisphere(wo, wi) ? absCosTheta(wi) * kInvPi : 0.f; // BxDF.cpp:568-571
    if (KIND == LOBE_MICRO_REFL_CONDUCTOR || KIND == LOBE_MICRO_REFL_DIELECTRIC) return microReflPdf(b, wo, wi);
    if (KIND == LOBE_MICRO_TRANS) return microTransPdf(b, wo, wi);
    return 0.f;
}
// BxDF::sample_f of one lobe.  Returns f; *pdf stays 0 when the lobe fails to sample.
// whShared: the microfacet normal when the caller has sampled it already (the two lobes of rough glass draw the same
// wh from the same numbers; sampled before the lanes split over the lobes, it is computed once for the wave)
template <int KIND>
DEV F3 lobeSample(const Bsdf& b, F3 col, F3 wo, float u0, float u1, F3* wi, float* pdf, int* sampledType, const F3* whShared = nullptr) {
    if (KIND == LOBE_LAMBERT || KIND == LOBE_OREN_NAYAR) { // BxDF.cpp:529-536
        *wi = cosineSampleHemisphere(u0, u1);
        if (wo.z < 0.f) wi->z *= -1.f;
        *pdf = lobePdf<KIND>(b, wo, *wi);
        return lobeF<KIND>(b, col, wo, *wi);
    } else if (KIND == LOBE_MICRO_REFL_CONDUCTOR || KIND == LOBE_MICRO_REFL_DIELECTRIC) { // BxDF.cpp:1102-1114
        if (wo.z == 0) return F3(0.f);
        F3 wh = whShared ? *whShared : trSampleWh(b.ax, b.ay, wo, u0, u1);
        if (dot(wo, wh) < 0) return F3(0.f);
        *wi = reflect(wo, wh);
        if (!sameHemisphere(wo, *wi)) return F3(0.f);
        *pdf = trPdf(b.ax, b.ay, wo, wh) / (4 * dot(wo, wh));
        return microReflF<KIND>(b, col, wo, *wi);
    } else if (KIND == LOBE_MICRO_TRANS) { // BxDF.cpp:1045-1056
        if (wo.z == 0) return F3(0.f);
        F3 wh = whShared ? *whShared : trSampleWh(b.ax, b.ay, wo, u0, u1);
        if (dot(wo, wh) < 0) return F3(0.f);
        float eta = cosTheta(wo) > 0 ? (b.dEtaI / b.dEtaT) : (b.dEtaT / b.dEtaI);
        if (!refract(wo, wh, eta, wi)) return F3(0.f);
        *pdf = microTransPdf(b, wo, *wi);
        return microTransF(b, col, wo, *wi);
    } else if (KIND == LOBE_SPEC_REFL) { // SpecularReflection::sample_f with FresnelDielectric(1, eta), BxDF.cpp:674-679,649-659
        *pdf = 1.0f;
        *wi = F3(-wo.x, -wo.y, wo.z);
        return (F3(frDielectric(cosTheta(wo), b.dEtaI, b.dEtaT)) * col) / absCosTheta(*wi);
    } else if (KIND == LOBE_SPEC_TRANS) { // SpecularTransmission::sample_f, BxDF.cpp:705-722 (radiance transport)
        bool entering = cosTheta(wo) > 0;
        float etaI = entering ? b.dEtaI : b.dEtaT;
        float etaT = entering ? b.dEtaT : b.dEtaI;
        if (!refract(wo, faceForward(F3(0, 0, 1), wo), etaI / etaT, wi)) return F3(0.f);
        *pdf = 1;
        F3 ft = col * (F3(1.f) - F3(frDielectric(cosTheta(*wi), b.dEtaI, b.dEtaT)));
        ft *= (etaI * etaI) / (etaT * etaT);
        return ft / absCosTheta(*wi);
    } else if (KIND == LOBE_FRESNEL_SPECULAR) { // BxDF.cpp:1124-1158; col0 = R (unused by the reference), col1 = T
        float F = frDielectric(cosTheta(wo), b.dEtaI, b.dEtaT);
        if (u0 < F) {
            *wi = F3(-wo.x, -wo.y, wo.z);
            *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
            *pdf = F;
            return (b.col1 * F) / absCosTheta(*wi);
        } else {
            bool entering = cosTheta(wo) > 0;
            float etaI = entering ? b.dEtaI : b.dEtaT;
            float etaT = entering ? b.dEtaT : b.dEtaI;
            if (!refract(wo, faceForward(F3(0, 0, 1), wo), etaI / etaT, wi)) return F3(0.f);
            F3 ft = b.col1 * (1 - F);
            ft *= (etaI * etaI) / (etaT * etaT);
            *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1 - F;
            return ft / absCosTheta(*wi);
        }
    }
    return F3(0.f);
}

// ---- rough glass: MicrofacetReflection + MicrofacetTransmission over one distribution, evaluated together ----
// For a pair of directions at most one of the two lobes is non-zero (reflection when both lie on one side of the
// surface, transmission otherwise), and both lobes are built from the same pieces: D and G of the distribution at a
// half vector, one dielectric Fresnel term, trPdf at a half vector.  The lanes of a wave differ in which lobe applies;
// with the half vector and the Fresnel argument picked per lane, the shared pieces run once for the whole wave instead
// of once per side.  Each lane's values come from the same operations on the same operands as in microReflF /
// microTransF / microReflPdf / microTransPdf above (x * 1 and x + 0 are exact).
DEV F3 roughGlassF(const Bsdf& b, F3 wo, F3 wi, bool refl, bool on0, bool on1) {
    bool live = (refl ? on0 : on1) && wi.z != 0 && wo.z != 0;
    if (!refl && sameHemisphere(wo, wi)) live = false;
    const float eta = cosTheta(wo) > 0 ? (b.dEtaT / b.dEtaI) : (b.dEtaI / b.dEtaT);
    const F3 whRaw = refl ? wi + wo : wo + wi * eta;
    if (refl && whRaw.x == 0 && whRaw.y == 0 && whRaw.z == 0) live = false;
    F3 wh = normalize(whRaw);
    if (!refl && wh.z < 0) wh = -wh;
    const float dO = dot(wo, wh), dI = dot(wi, wh);
    if (!refl && dO * dI > 0) live = false;
    const float D = trD(b.ax, b.ay, wh), G = trG(b.ax, b.ay, wo, wi);
    const float F = frDielectric(refl ? dot(wi, faceForward(wh, F3(0, 0, 1))) : dO, b.dEtaI, b.dEtaT);
    if (!live) return F3(0.f);
    if (refl) return b.col0 * D * G * F3(F) / (4 * absCosTheta(wi) * absCosTheta(wo));
    const float cosThetaO = cosTheta(wo), cosThetaI = cosTheta(wi);
    const float sqrtDenom = dO + eta * dI;
    const float factor = 1 / eta;
    const float sv = fabsf(D * G * eta * eta * fabsf(dI) * fabsf(dO) * factor * factor / (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    return (F3(1.f) - F3(F)) * b.col1 * sv;
}
// whSampled: the half vector MicrofacetReflection::sample_f has in hand (it does not rebuild it from wo + wi); nullptr when
// evaluating a given pair of directions
DEV float roughGlassPdf(const Bsdf& b, F3 wo, F3 wi, bool on0, bool on1, const F3* whSampled = nullptr) {
    const bool same = sameHemisphere(wo, wi);
    const float eta = cosTheta(wo) > 0 ? (b.dEtaI / b.dEtaT) : (b.dEtaT / b.dEtaI);
    F3 wh = normalize(same ? wo + wi : wo + wi * eta);
    if (whSampled && same) wh = *whSampled;
    const float dO = dot(wo, wh), dI = dot(wi, wh);
    const bool live = (same ? on0 : on1) && (same || !(dO * dI > 0));
    const float tp = trPdf(b.ax, b.ay, wo, wh);
    if (!live) return 0.f;
    if (same) return tp / (4 * dO);
    const float sqrtDenom = dO + eta * dI;
    return tp * fabsf((eta * eta * dI) / (sqrtDenom * sqrtDenom));
}

// sum of lobe f over the slots matching flags on the side (reflection / transmission) of the pair
template <int MK, int I>
DEV F3 sideF(const Bsdf& b, F3 wo, F3 wi, int flags, bool refl) {
    constexpr int K = LobeOf<MK, I>::kind;
    if (!lobeOn<MK, I>(b, flags)) return F3(0.f);
    constexpr int t = lobeType(K < 0 ? 0 : K);
    if ((refl && (t & BSDF_REFLECTION)) || (!refl && (t & BSDF_TRANSMISSION))) return lobeF<K>(b, I == 0 ? b.col0 : b.col1, wo, wi);
    return F3(0.f);
}
// BSDF::f, BxDF.cpp:972-985
template <int MK>
DEV F3 bsdfF(const Bsdf& b, F3 woW, F3 wiW, int flags) {
    F3 wi = toLocal(b, wiW), wo = toLocal(b, woW);
    if (wo.z == 0) return F3(0.f);
    bool refl = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    if (MK == MK_GLASS_ROUGH) return roughGlassF(b, wo, wi, refl, lobeOn<MK, 0>(b, flags), lobeOn<MK, 1>(b, flags));
    F3 f(0.f);
    f += sideF<MK, 0>(b, wo, wi, flags, refl);
    f += sideF<MK, 1>(b, wo, wi, flags, refl);
    return f;
}
// BSDF::Pdf, BxDF.cpp:954-970
template <int MK>
DEV float bsdfPdf(const Bsdf& b, F3 woW, F3 wiW, int flags) {
    if (!b.has0 && !b.has1) return 0.f;
    F3 wo = toLocal(b, woW), wi = toLocal(b, wiW);
    if (MK == MK_GLASS_ROUGH) {
        const bool on0 = lobeOn<MK, 0>(b, flags), on1 = lobeOn<MK, 1>(b, flags);
        const int n = (on0 ? 1 : 0) + (on1 ? 1 : 0);
        return n > 0 ? roughGlassPdf(b, wo, wi, on0, on1) / n : 0.f;
    }
    float pdf = 0.f; int matching = 0;
    if (lobeOn<MK, 0>(b, flags)) { ++matching; pdf += lobePdf<LobeOf<MK, 0>::kind>(b, wo, wi); }
    if (lobeOn<MK, 1>(b, flags)) { ++matching; pdf += lobePdf<LobeOf<MK, 1>::kind>(b, wo, wi); }
    return matching > 0 ? pdf / matching : 0.f;
}
// BSDF::sample_f, BxDF.cpp:885-952.  *pdf == 0 or a black return mean "no sample".
template <int MK>
DEV F3 bsdfSample(const Bsdf& b, F3 woW, F3* wiW, float u0, float u1, float* pdf, int flags, int* sampledType) {
    constexpr int K0 = LobeOf<MK, 0>::kind, K1 = LobeOf<MK, 1>::kind;
    *pdf = 0; *sampledType = 0;
    bool on0 = lobeOn<MK, 0>(b, flags), on1 = lobeOn<MK, 1>(b, flags);
    int matching = (on0 ? 1 : 0) + (on1 ? 1 : 0);
    if (matching == 0) return F3(0.f);
    int comp = min((int)floorf(u0 * matching), matching - 1);
    bool pick1 = on1 && (!on0 || comp == 1); // the comp-th matching slot
    float ur0 = smin(u0 * matching - comp, kOneMinusEpsilon);
    F3 wo = toLocal(b, woW), wi(0.f);
    if (wo.z == 0) return F3(0.f);
    if (MK == MK_GLASS_ROUGH) { // MicrofacetReflection::sample_f / MicrofacetTransmission::sample_f (BxDF.cpp:1102-1114,1045-1056), then BSDF::sample_f's sums
        *sampledType = pick1 ? lobeType(LOBE_MICRO_TRANS) : lobeType(LOBE_MICRO_REFL_DIELECTRIC);
        const F3 wh = trSampleWh(b.ax, b.ay, wo, ur0, u1); // either lobe draws this wh
        bool ok = !(dot(wo, wh) < 0);
        const float eta = cosTheta(wo) > 0 ? (b.dEtaI / b.dEtaT) : (b.dEtaT / b.dEtaI);
        F3 wt(0.f);
        const bool refracted = refract(wo, wh, eta, &wt);
        wi = pick1 ? wt : reflect(wo, wh);
        if (pick1 ? !refracted : !sameHemisphere(wo, wi)) ok = false;
        // the sampled lobe's pdf; the other lobe's is zero on this side of the surface, which roughGlassPdf decides by itself
        const bool same = sameHemisphere(wo, wi);
        if (pick1 ? same : !same) ok = false; // microTransPdf: 0 on the reflection side (the reflection side was checked above)
        const float own = roughGlassPdf(b, wo, wi, !pick1, pick1, &wh);
        if (!ok || own == 0) { *pdf = 0; *sampledType = 0; return F3(0.f); }
        *pdf = own;
        *wiW = toWorld(b, wi);
        if (matching > 1) *pdf /= matching; // + the other lobe's pdf: exactly 0 here
        const bool refl = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
        return roughGlassF(b, wo, wi, refl, on0, on1);
    }
    F3 f;
    bool specular;
    constexpr bool micro0 = K0 == LOBE_MICRO_REFL_CONDUCTOR || K0 == LOBE_MICRO_REFL_DIELECTRIC || K0 == LOBE_MICRO_TRANS;
    constexpr bool micro1 = K1 == LOBE_MICRO_REFL_CONDUCTOR || K1 == LOBE_MICRO_REFL_DIELECTRIC || K1 == LOBE_MICRO_TRANS;
    F3 whBoth(0.f);
    if (micro0 && micro1) whBoth = trSampleWh(b.ax, b.ay, wo, ur0, u1); // one distribution, the same numbers: either lobe would draw this wh
    const F3* whShared = (micro0 && micro1) ? &whBoth : nullptr;
    if (pick1) {
        *sampledType = lobeType(K1 < 0 ? 0 : K1);
        specular = (lobeType(K1 < 0 ? 0 : K1) & BSDF_SPECULAR) != 0;
        f = lobeSample<K1>(b, b.col1, wo, ur0, u1, &wi, pdf, sampledType, whShared);
    } else {
        *sampledType = lobeType(K0);
        specular = (lobeType(K0) & BSDF_SPECULAR) != 0;
        f = lobeSample<K0>(b, b.col0, wo, ur0, u1, &wi, pdf, sampledType, whShared);
    }
    if (*pdf == 0) { *sampledType = 0; return F3(0.f); }
    *wiW = toWorld(b, wi);
    if (!specular && matching > 1) {
        if (pick1) *pdf += lobePdf<K0>(b, wo, wi);
        else *pdf += lobePdf<K1>(b, wo, wi);
    }
    if (matching > 1) *pdf /= matching;
    if (!specular) {
        bool refl = dot(*wiW, b.ng) * dot(woW, b.ng) > 0;
        f = F3(0.f);
        f += sideF<MK, 0>(b, wo, wi, flags, refl);
        f += sideF<MK, 1>(b, wo, wi, flags, refl);
    }
    return f;
}

// ---------------------------------------------------------------- MK_GENERAL: a list of lobes walked at run time
// Mirror, translucent, substrate, uber and mix materials (Enigma/Material.cpp:318-328, :276-309, :424-448, :536-592,
// :146-163).  Which lobes exist was decided when the scene was uploaded (constant parameters: the IsBlack tests of
// computeScatteringFunctions have one answer per material); the shade kernel of this kind follows BSDF::f / Pdf / sample_f
// (Enigma/BxDF.cpp:885-985) over the list as the reference does over its BxDF pointers.
struct BsdfG {
    const DevLobe* lobes;  // in global memory; `local` when a texture decides colours and lobes at this hit
    int n;
    float eta;             // BSDF::m_eta
    F3 ng, ss, ts;         // as in Bsdf
    DevLobe local[kMaxLobes];
};
template <int MK> struct BsdfOf { using T = Bsdf; };
template <> struct BsdfOf<MK_GENERAL> { using T = BsdfG; };

template <int MK>
DEV void makeBsdf(const DevMaterial& m, F3 ng, F3 ns, F3 dpdu, BsdfG* b, const float4* texColors = nullptr, const DevLobe* lobes = nullptr, bool multipleLobes = true) {
    b->ng = ng;
    b->ss = normalize(dpdu);
    b->ts = cross(ns, b->ss);
    b->eta = m.eta;
    const uint32_t nMulti = m.lobeInfo & 0xffu, nSingle = (m.lobeInfo >> 8) & 0xffu;
    b->lobes = lobes + m.lobeOff + (multipleLobes ? 0u : nMulti);
    b->n = (int)(multipleLobes ? nMulti : nSingle);
    if (!texColors || !(m.tex0 | m.tex1)) return;
    // textured colours: the lobes that computeScatteringFunctions would add with the textures' values at this hit
    int kept = 0;
    for (int i = 0; i < b->n; ++i) {
        DevLobe l = b->lobes[i];
        const uint32_t t = l.tex;
        bool lit0 = l.col[0] != 0.f || l.col[1] != 0.f || l.col[2] != 0.f, lit1 = l.col2[0] != 0.f || l.col2[1] != 0.f || l.col2[2] != 0.f;
        for (int k = 0; k < 2; ++k) {
            const uint32_t slot = (t >> (8 * k)) & 0xffu;
            if (!slot) continue;
            const float4 c = texColors[slot - 1u];
            F3 v(c.x, c.y, c.z);
            if (!(t & kLobeNoClamp)) v = F3(smax(v.x, 0.f), smax(v.y, 0.f), smax(v.z, 0.f)); // Spectrum::Clamp()
            const bool rawLit = v.x != 0.f || v.y != 0.f || v.z != 0.f;
            float* dst = k == 0 ? l.col : l.col2;
            dst[0] *= v.x; dst[1] *= v.y; dst[2] *= v.z;
            const bool lit = (t & kLobeRaw) ? rawLit : (dst[0] != 0.f || dst[1] != 0.f || dst[2] != 0.f);
            if (k == 0) lit0 = lit; else lit1 = lit;
        }
        if ((t & kLobeOwn) && !lit0) continue;
        if ((t & kLobeEither) && !lit0 && !lit1) continue;
        b->local[kept++] = l;
    }
    b->lobes = b->local;
    b->n = kept;
}

// the lobe's parameters in the shape the per-kind lobe functions above read them
DEV Bsdf lobeParams(const DevLobe& l) {
    Bsdf t;
    t.has0 = t.has1 = true;
    t.col0 = F3(l.col[0], l.col[1], l.col[2]); t.col1 = F3(l.col2[0], l.col2[1], l.col2[2]);
    t.cEta = t.col0; t.cK = t.col1;
    t.ax = l.ax; t.ay = l.ay;
    t.eta = 1.f;
    t.dEtaI = l.etaI; t.dEtaT = l.etaT;
    t.ng = t.ss = t.ts = F3(0.f);
    return t;
}
DEV float pow5(float v) { return (v * v) * (v * v) * v; }
// FresnelBlend::f, Enigma/BxDF.cpp:788-809
DEV F3 fresnelBlendF(const Bsdf& t, F3 Rd, F3 Rs, F3 wo, F3 wi) {
    F3 diffuse = (28.f / (23.f * kPi)) * Rd * (F3(1.f) - Rs) * (1 - pow5(1 - .5f * absCosTheta(wi))) * (1 - pow5(1 - .5f * absCosTheta(wo)));
    F3 wh = wi + wo;
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return F3(0.f);
    wh = normalize(wh);
    F3 schlick = Rs + pow5(1 - dot(wi, wh)) * (F3(1.f) - Rs);
    F3 specular = trD(t.ax, t.ay, wh) / (4 * absdot(wi, wh) * smax(absCosTheta(wi), absCosTheta(wo))) * schlick;
    return diffuse + specular;
}
// FresnelBlend::pdf, Enigma/BxDF.cpp:833-839
DEV float fresnelBlendPdf(const Bsdf& t, F3 wo, F3 wi) {
    if (!sameHemisphere(wo, wi)) return 0.f;
    F3 wh = normalize(wo + wi);
    float pdf_wh = trPdf(t.ax, t.ay, wo, wh);
    return .5f * (absCosTheta(wi) * kInvPi + pdf_wh / (4 * dot(wo, wh)));
}
DEV F3 lobeScale(const DevLobe& l, F3 f) { return l.scaled ? F3(l.scale[0], l.scale[1], l.scale[2]) * f : f; } // ScaledBxDF::f / sample_f, Enigma/BxDF.cpp:617-625
DEV F3 lobeFRt(const DevLobe& l, F3 wo, F3 wi) {
    const Bsdf t = lobeParams(l);
    F3 f(0.f);
    switch (l.kind) {
    case LOBE_LAMBERT: case LOBE_LAMBERT_TRANS: f = t.col0 * kInvPi; break;
    case LOBE_OREN_NAYAR: f = orenNayarF(t, t.col0, wo, wi); break;
    case LOBE_MICRO_REFL_CONDUCTOR: f = microReflF<LOBE_MICRO_REFL_CONDUCTOR>(t, F3(1.f), wo, wi); break; // R = 1 (Enigma/Material.cpp:131)
    case LOBE_MICRO_REFL_DIELECTRIC: f = microReflF<LOBE_MICRO_REFL_DIELECTRIC>(t, t.col0, wo, wi); break;
    case LOBE_MICRO_TRANS: f = microTransF(t, t.col0, wo, wi); break;
    case LOBE_FRESNEL_BLEND: f = fresnelBlendF(t, t.col0, t.col1, wo, wi); break;
    default: break; // the specular lobes
    }
    return lobeScale(l, f);
}
DEV float lobePdfRt(const DevLobe& l, F3 wo, F3 wi) {
    if (l.scaled) return sameHemisphere(wo, wi) ? absCosTheta(wi) * kInvPi : 0.f; // BxDF::pdf: ScaledBxDF has none of its own
    const Bsdf t = lobeParams(l);
    switch (l.kind) {
    case LOBE_LAMBERT: case LOBE_OREN_NAYAR: return sameHemisphere(wo, wi) ? absCosTheta(wi) * kInvPi : 0.f;
    case LOBE_LAMBERT_TRANS: return !sameHemisphere(wo, wi) ? absCosTheta(wi) * kInvPi : 0.f;
    case LOBE_MICRO_REFL_CONDUCTOR: case LOBE_MICRO_REFL_DIELECTRIC: return microReflPdf(t, wo, wi);
    case LOBE_MICRO_TRANS: return microTransPdf(t, wo, wi);
    case LOBE_FRESNEL_BLEND: return fresnelBlendPdf(t, wo, wi);
    default: return 0.f;
    }
}
// the wrapped lobe's sample_f; its density is passed through unscaled (Enigma/BxDF.cpp:622-625)
DEV F3 lobeSampleRt(const DevLobe& l, F3 wo, float u0, float u1, F3* wi, float* pdf, int* sampledType) {
    const Bsdf t = lobeParams(l);
    F3 f(0.f);
    switch (l.kind) {
    case LOBE_LAMBERT: f = lobeSample<LOBE_LAMBERT>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_OREN_NAYAR: f = lobeSample<LOBE_OREN_NAYAR>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_LAMBERT_TRANS: // Enigma/BxDF.cpp:1179-1190
        *wi = cosineSampleHemisphere(u0, u1);
        if (wo.z > 0) wi->z *= -1;
        *pdf = !sameHemisphere(wo, *wi) ? absCosTheta(*wi) * kInvPi : 0.f;
        f = t.col0 * kInvPi;
        break;
    case LOBE_MICRO_REFL_CONDUCTOR: f = lobeSample<LOBE_MICRO_REFL_CONDUCTOR>(t, F3(1.f), wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_MICRO_REFL_DIELECTRIC: f = lobeSample<LOBE_MICRO_REFL_DIELECTRIC>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_MICRO_TRANS: f = lobeSample<LOBE_MICRO_TRANS>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_SPEC_REFL: f = lobeSample<LOBE_SPEC_REFL>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_SPEC_TRANS: f = lobeSample<LOBE_SPEC_TRANS>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_FRESNEL_SPECULAR: f = lobeSample<LOBE_FRESNEL_SPECULAR>(t, t.col0, wo, u0, u1, wi, pdf, sampledType); break;
    case LOBE_SPEC_REFL_NOOP: // SpecularReflection::sample_f over FresnelNoOp, Enigma/BxDF.cpp:674-679
        *pdf = 1.0f;
        *wi = F3(-wo.x, -wo.y, wo.z);
        f = (F3(1.f) * t.col0) / absCosTheta(*wi);
        break;
    case LOBE_FRESNEL_BLEND: { // Enigma/BxDF.cpp:813-831
        bool ok = true;
        if (u0 < .5f) {
            *wi = cosineSampleHemisphere(smin(2 * u0, kOneMinusEpsilon), u1);
            if (wo.z < 0) wi->z *= -1;
        } else {
            F3 wh = trSampleWh(t.ax, t.ay, wo, smin(2 * (u0 - .5f), kOneMinusEpsilon), u1);
            *wi = reflect(wo, wh);
            if (!sameHemisphere(wo, *wi)) ok = false;
        }
        if (ok) { *pdf = fresnelBlendPdf(t, wo, *wi); f = fresnelBlendF(t, t.col0, t.col1, wo, *wi); }
        break;
    }
    default: break;
    }
    return lobeScale(l, f);
}

template <int MK>
DEV int numComponents(const BsdfG& b, int flags) {
    int n = 0;
    for (int i = 0; i < b.n; ++i) n += lobeMatches((int)b.lobes[i].kind, flags) ? 1 : 0;
    return n;
}
DEV F3 generalSideF(const BsdfG& b, F3 wo, F3 wi, int flags, bool refl) {
    F3 f(0.f);
    for (int i = 0; i < b.n; ++i) {
        const DevLobe& l = b.lobes[i];
        const int t = lobeType((int)l.kind);
        if (lobeMatches((int)l.kind, flags) && ((refl && (t & BSDF_REFLECTION)) || (!refl && (t & BSDF_TRANSMISSION)))) f += lobeFRt(l, wo, wi);
    }
    return f;
}
template <int MK>
DEV F3 bsdfF(const BsdfG& b, F3 woW, F3 wiW, int flags) { // BSDF::f, Enigma/BxDF.cpp:972-985
    F3 wi = toLocal(b, wiW), wo = toLocal(b, woW);
    if (wo.z == 0) return F3(0.f);
    return generalSideF(b, wo, wi, flags, dot(wiW, b.ng) * dot(woW, b.ng) > 0);
}
template <int MK>
DEV float bsdfPdf(const BsdfG& b, F3 woW, F3 wiW, int flags) { // BSDF::Pdf, Enigma/BxDF.cpp:954-970
    if (b.n == 0) return 0.f;
    F3 wo = toLocal(b, woW), wi = toLocal(b, wiW);
    float pdf = 0.f; int matching = 0;
    for (int i = 0; i < b.n; ++i)
        if (lobeMatches((int)b.lobes[i].kind, flags)) { ++matching; pdf += lobePdfRt(b.lobes[i], wo, wi); }
    return matching > 0 ? pdf / matching : 0.f;
}
template <int MK>
DEV F3 bsdfSample(const BsdfG& b, F3 woW, F3* wiW, float u0, float u1, float* pdf, int flags, int* sampledType) { // BSDF::sample_f, Enigma/BxDF.cpp:885-952
    *pdf = 0; *sampledType = 0;
    const int matching = numComponents<MK>(b, flags);
    if (matching == 0) return F3(0.f);
    const int comp = min((int)floorf(u0 * matching), matching - 1);
    int chosen = -1, count = comp;
    for (int i = 0; i < b.n && chosen < 0; ++i)
        if (lobeMatches((int)b.lobes[i].kind, flags) && count-- == 0) chosen = i;
    const float ur0 = smin(u0 * matching - comp, kOneMinusEpsilon);
    F3 wo = toLocal(b, woW), wi(0.f);
    if (wo.z == 0) return F3(0.f);
    const DevLobe& l = b.lobes[chosen];
    const int type = lobeType((int)l.kind);
    *sampledType = type;
    F3 f = lobeSampleRt(l, wo, ur0, u1, &wi, pdf, sampledType);
    if (*pdf == 0) { *sampledType = 0; return F3(0.f); }
    *wiW = toWorld(b, wi);
    const bool specular = (type & BSDF_SPECULAR) != 0;
    if (!specular && matching > 1)
        for (int i = 0; i < b.n; ++i)
            if (i != chosen && lobeMatches((int)b.lobes[i].kind, flags)) *pdf += lobePdfRt(b.lobes[i], wo, wi);
    if (matching > 1) *pdf /= matching;
    if (!specular) f = generalSideF(b, wo, wi, flags, dot(*wiW, b.ng) * dot(woW, b.ng) > 0);
    return f;
}

} // namespace edev
