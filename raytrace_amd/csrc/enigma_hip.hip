// C ABI of the MI355X PathIntegrator back end (see include/enigma_hip.h).
// Host orchestration of the wavefront kernels in kernels.h; no CPU rendering path exists here:
// every entry point that computes runs on the GPU or fails.
#include "../../include/enigma_hip.h"

#include "bvh_build.h"
#include "kernels_vol.h"
#include "kernels_tex.h"
#include "envmap_host.h"
#include <memory>
#include "sobol_host.h"
#include "halton_host.h"
#include "spectrum_host.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace edev;

// ENIGMA_EXPERIMENT_BUILD (never the product): only the PathIntegrator on triangle scenes whose tree is at most 11 levels
// deep or that fit LDS is compiled -- what a traversal experiment on the bench scenes needs, in a fifth of the compile time.
// Everything else returns ENIGMA_E_UNSUPPORTED.
#ifdef ENIGMA_EXPERIMENT_BUILD
#define ENIGMA_FULL_BUILD 0
#else
#define ENIGMA_FULL_BUILD 1
#endif

namespace {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

uint32_t roundUpPow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; } // MathCommon.h:92-98
int log2Int(float v) { static const float invLog2 = 1.f / logf(2.f); return (int)floorf(logf(v) * invLog2); }               // MathCommon.h:117-124

} // namespace

struct EnigmaCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error = "";

    // scene
    bool haveScene = false;
    DevBuf<float4> nodes, tris;
    DevBuf<DevMaterial> materials;
    DevBuf<DevLobe> lobes;
    DevBuf<DevLight> lights;
    DevBuf<float> lightCdf, lightFunc;
    DevBuf<uint32_t> primId;
    DevBuf<DevMedium> media;
    // infinite area lights: per light the radiance map and its sampling distribution
    std::vector<std::unique_ptr<DevBuf<float4>>> envTexels;
    std::vector<std::unique_ptr<DevBuf<float>>> envFloats;
    DevBuf<DevEnvMap> envMaps;
    DevBuf<uint32_t> infiniteLights;
    DevBuf<uint32_t> triMedium;
    DevBuf<float4> triShade;
    DevBuf<DevSphere> spheres;
    DevBuf<float> spatialTable, halton;
    DevBuf<uint32_t> spatialSlot, spatialCtl, spatialNew; // see DevScene
    float worldLo[3] = {0, 0, 0}, worldHi[3] = {0, 0, 0}; // Scene::worldBound(): the union of the primitives' bounds
    DevBuf<float4> sphereGeom;
    DevBuf<float4> texTexels, texColors;
    DevBuf<DevTexture> textures;
    bool kindTextured[MK_COUNT] = {};
    bool haveTrilinear = false, smoothGlass = false;
    bool haveNoMaterial = false; // some triangle has no material (a medium boundary): VolPathIntegrator only
    std::vector<DevLight> hostLights;
    std::vector<float> lightPower; // Light::Power().y() per light
    bool kindPresent[MK_COUNT] = {};
    bool haveAreaLights = false;
    uint32_t refillIdle = 16;
    int residentTraceBlocks = 256 * 6; // blocks of a traversal kernel the chip holds at once: its waves stream rays until the queue is empty
    DevScene scene{};
    uint32_t bvhDepth = 0, bvhNodes = 0, bvhStack = 0;
    // grid of the shade kernels, in blocks per CU (their threads stride over the queues).  Measured per bounce on Cornell /
    // the 1.18 M-triangle scene: 4: 1.02 / 2.91 ms, 8: 0.95 / 2.72, 16: 0.93 / 2.67, 32: 0.92 / 2.62, 64: 0.93 / 2.62, 512: 0.99 /
    // 2.63 (a block of a scene in LDS stages its tables first)
    int shadeBlocksPerCu = 32;
    double msBvhBuild = 0;

    // camera / film / integrator
    bool haveCamera = false, haveFilm = false, haveInteg = false;
    DevCamera camera{};
    EnigmaFilmDesc filmDesc{};
    DevBuf<float4> filmPixels;
    DevBuf<float> filterTable;
    DevFilm film{};
    EnigmaIntegratorDesc integDesc{};
    DevIntegrator integ{};
    int64_t spp = 0;
    uint32_t log2Res = 0, resolution = 0, nDims = 0;
    uint32_t samplerSelected = ENIGMA_SAMPLER_SOBOL, samplerKind = ENIGMA_SAMPLER_SOBOL; // chosen for the next configuration / of the configured integrator
    bool centerSelected = false, centerKind = false;                                       // "samplepixelcenter": chosen / configured
    enigma::HaltonTables haltonSampler;
    DevBuf<uint32_t> tx, ty, sampleConst;
    DevSampler sampler{};
    bool lightDistDirty = true;
    // VolPathIntegrator
    bool useVolPath = false;
    DevBuf<float4> shRec;
    size_t volCapacity = 0;
    // DirectLightingIntegrator
    bool useDirect = false, directDirty = true;
    EnigmaDirectLightingDesc directDesc{};
    bool glassBranches = false, glassSpecular = false; // a smooth glass with Kr and Kt / with either
    bool deferredLobeChoice = false; // a material with a specular reflection lobe and several specular transmission lobes
    DevBuf<uint32_t> bitConst, arrayOff, pendCount;
    DevBuf<float4> pendO, pendD, pendB;
    size_t rayCapacity = 0, pendCapacity = 0;

    // wavefront state
    size_t capacity = 0;
    DevBuf<float4> pathRec, L, shO, shD, miRec; // pathRec: four float4 per path (rayO, rayD, beta, hit)
    DevBuf<float> shWz;
    DevBuf<uint32_t> shIdx, miIdx;
    DevBuf<uint32_t> extQ[2], matQ[kQueueKinds];
    DevBuf<float2> pfilm;
    DevBuf<DevCounters> counters;
    DevBuf<unsigned long long> visits; // [kVisitKinds][kVisitTallies][kVisitSlots], see dev_trace.h

    // multi-GPU: the communicator this handle's rank belongs to (enigma_comm_init), nullptr = single GPU
    ncclComm_t comm = nullptr;
    int commRank = 0, commWorld = 1;

    // stats
    EnigmaStats stats{};
    std::vector<hipEvent_t> evPool;      // kEventsPerStep per traversal step of a render call
    std::vector<uint8_t> shadowTimed;   // per step: the shadow kernel was launched (its two events were recorded)
    std::vector<uint8_t> shadeTimed;    // per step: the shading kernels were launched

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
};

#define HIPCHK(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return (h)->fail(ENIGMA_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

static hipError_t resolveScratch(int device);
extern "C" int enigma_comm_destroy(EnigmaCtx* h);

// No C++ exception may cross the C boundary: a failed allocation of a host-side table becomes an error code.
template <class F>
static int guarded(EnigmaCtx* h, F body) {
    try { return body(); }
    catch (const std::bad_alloc&) { return h->fail(ENIGMA_E_DEVICE, "out of host memory"); }
    catch (const std::exception& e) { return h->fail(ENIGMA_E_DEVICE, "%s", e.what()); }
    catch (...) { return h->fail(ENIGMA_E_DEVICE, "unknown exception"); }
}

extern "C" int enigma_ctx_create(int device_id, EnigmaCtx** out) {
    if (!out) return ENIGMA_E_INVALID;
    *out = nullptr;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0 || device_id < 0 || device_id >= nDev) return ENIGMA_E_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return ENIGMA_E_DEVICE;
    if (resolveScratch(device_id) != hipSuccess) return ENIGMA_E_DEVICE;
    EnigmaCtx* h = new EnigmaCtx();
    h->device = device_id;
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return ENIGMA_E_DEVICE; }
    if (h->counters.alloc(1) != hipSuccess) { delete h; return ENIGMA_E_DEVICE; }
    (void)hipMemset(h->counters.p, 0, sizeof(DevCounters));
    if (h->visits.alloc((size_t)kVisitKinds * kVisitTallies * kVisitSlots) != hipSuccess) { h->counters.release(); delete h; return ENIGMA_E_DEVICE; }
    (void)hipMemset(h->visits.p, 0, h->visits.n * sizeof(unsigned long long));
    h->scene.visits = h->visits.p;
    *out = h;
    return ENIGMA_OK;
}

extern "C" void enigma_ctx_destroy(EnigmaCtx* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    (void)enigma_comm_destroy(h);
    for (hipEvent_t e : h->evPool) (void)hipEventDestroy(e);
    DevBuf<float4>* f4[] = {&h->nodes, &h->tris, &h->filmPixels, &h->pathRec, &h->L, &h->shO, &h->shD, &h->miRec};
    for (auto* b : f4) b->release();
    h->shWz.release(); h->shRec.release();
    h->media.release(); h->triMedium.release(); h->triShade.release(); h->texTexels.release(); h->texColors.release(); h->textures.release(); h->spheres.release(); h->sphereGeom.release(); h->spatialTable.release(); h->halton.release(); h->spatialSlot.release(); h->spatialCtl.release(); h->spatialNew.release();
    for (auto& b : h->envTexels) b->release();
    for (auto& b : h->envFloats) b->release();
    h->envMaps.release(); h->infiniteLights.release();
    h->materials.release(); h->lobes.release(); h->lights.release(); h->lightCdf.release(); h->lightFunc.release(); h->primId.release();
    h->filterTable.release(); h->tx.release(); h->ty.release(); h->sampleConst.release();
    h->extQ[0].release(); h->extQ[1].release();
    for (auto& q : h->matQ) q.release();
    h->pfilm.release(); h->counters.release(); h->visits.release(); h->shIdx.release(); h->miIdx.release();
    h->bitConst.release(); h->arrayOff.release(); h->pendCount.release(); h->pendO.release(); h->pendD.release(); h->pendB.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char* enigma_last_error(const EnigmaCtx* h) { return h ? h->error.c_str() : "null context"; }

// Material::computeScatteringFunctions' parameter evaluation for constant parameters (Enigma/Material.cpp:76-255): kind,
// clamped colours, roughness -> alpha.  Texture bindings and lobe presence are the caller's business.
static bool convertMaterial(const EnigmaMaterial& m, DevMaterial& d) {
    auto clamp01inf = [](float v) { return v < 0.f ? 0.f : v; }; // Spectrum::Clamp(0, inf)
    auto remap = [](float r) { // TrowbridgeReitzDistribution::RoughnessToAlpha, Enigma/MicrofacetDist.cpp:323-328
    r = std::max(r, (float)1e-3);
    float x = std::log(r);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    };
    float ur = m.uroughness, vr = m.vroughness;
    float orenA = 0.f, orenB = 0.f;
    bool specularGlass = false;
    switch (m.type) {
    case ENIGMA_MAT_MATTE:
        d.kind = MK_MATTE;
        for (int c = 0; c < 3; ++c) d.c0[c] = clamp01inf(m.kd[c]);
        {   // sigma clamped to [0, 90] degrees; != 0 selects OrenNayer (Enigma/Material.cpp:86-91, Enigma/BxDF.cpp:744-753)
            const float sig = std::min(std::max(m.sigma, 0.f), 90.f);
            if (sig != 0.f) {
                d.kind = MK_MATTE_ROUGH;
                const float sigma = sig * (3.1415926535897932384f / 180.0f), sigma2 = sigma * sigma;
                orenA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                orenB = (0.45f * sigma2) / (sigma2 + 0.09f);
            }
        }
        break;
    case ENIGMA_MAT_METAL:
        d.kind = MK_METAL;
        for (int c = 0; c < 3; ++c) { d.c0[c] = m.eta[c]; d.c1[c] = m.k[c]; }
        if (m.remap_roughness) { ur = remap(ur); vr = remap(vr); }
        break;
    case ENIGMA_MAT_GLASS:
        specularGlass = (ur == 0.f && vr == 0.f);
        d.kind = specularGlass ? MK_GLASS_SMOOTH : MK_GLASS_ROUGH;
        for (int c = 0; c < 3; ++c) { d.c0[c] = clamp01inf(m.kr[c]); d.c1[c] = clamp01inf(m.kt[c]); }
        d.eta = m.index;
        if (!specularGlass && m.remap_roughness) { ur = remap(ur); vr = remap(vr); }
        break;
    case ENIGMA_MAT_PLASTIC:
        d.kind = MK_PLASTIC;
        for (int c = 0; c < 3; ++c) { d.c0[c] = clamp01inf(m.kd[c]); d.c1[c] = clamp01inf(m.ks[c]); }
        if (m.remap_roughness) ur = remap(ur);
        vr = ur;
        break;
    default:
        return false;
    }
    d.alphax = std::max(float(0.001), ur); // TrowbridgeReitzDistribution ctor, Enigma/MicrofacetDist.cpp:256-263
    d.alphay = std::max(float(0.001), vr);
    if (d.kind == MK_MATTE_ROUGH) { d.alphax = orenA; d.alphay = orenB; } // the Oren-Nayar coefficients travel in the alpha slots
    return true;
}

// An emissive triangle of a mesh with normals: the side of its plane that its vertex normals lie on, +1 for the side of
// Cross(p1 - p0, p2 - p0), -1 for the other, 0 when they disagree or graze the plane.  Triangle::sample and
// Triangle::intersect turn the geometric normal to face the interpolated one (Enigma/TriangleMesh.cpp:618-623,
// Enigma/Interaction.cpp:139-141); with all three on one side that is the same turn at every point of the triangle.
// The cross product is the single-precision one the sampling code itself forms (same operations, same order): for a
// sliver whose exact normal drowns in the cancellation, the turn must follow the normal that code sees, not the exact one.
static int emitterNormalSide(const EnigmaSceneDesc* s, uint32_t t) {
    const float* p[3]; const float* n[3];
    for (int k = 0; k < 3; ++k) { p[k] = &s->vertices[3 * (size_t)s->indices[3 * (size_t)t + k]]; n[k] = &s->normals[3 * (size_t)s->indices[3 * (size_t)t + k]]; }
    const float e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
    const float e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
    const float gx = e1[1] * e2[2] - e2[1] * e1[2], gy = e1[2] * e2[0] - e2[2] * e1[0], gz = e1[0] * e2[1] - e2[0] * e1[1];
    const double g[3] = {(double)gx, (double)gy, (double)gz};
    const double gl = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (gl == 0) return 1; // no area: never hit, and its sampled normal is NaN whichever way it is turned (the pdf test drops the sample)
    int pos = 0, neg = 0;
    for (int k = 0; k < 3; ++k) {
        const double nl = std::sqrt((double)n[k][0] * n[k][0] + (double)n[k][1] * n[k][1] + (double)n[k][2] * n[k][2]);
        const double c = (g[0] * n[k][0] + g[1] * n[k][1] + g[2] * n[k][2]) / (gl * nl);
        if (c > 1e-3) ++pos; else if (c < -1e-3) ++neg; // NaN (a zero normal) counts for neither
    }
    return pos == 3 ? 1 : neg == 3 ? -1 : 0;
}

// Material::computeScatteringFunctions as the list of BxDFs it adds, in its order (Enigma/Material.cpp:76-592).  Used for
// the materials the shade kernels walk at run time: mirror, translucent, substrate, uber, and mix over any of the nine.
// A constant colour decides its IsBlack tests here; a colour behind a trilinear texture (tex_* then holds the DEVICE
// texture index + 1; textures that evaluate to a constant were folded into the values by the caller) leaves them to the
// shade kernel: the lobe carries the texture's slot and the rule for its existence (DevLobe::tex).  *eta: BSDF::m_eta.
struct LobeList {
    std::vector<DevLobe> lobes;
    std::vector<std::pair<uint32_t, uint32_t>> tex; // per lobe: device texture + 1 behind col / col2
};
static bool materialLobes(const EnigmaMaterial* table, uint32_t idx, bool multipleLobes, LobeList& out, float* eta, std::string& why) {
    const EnigmaMaterial& m = table[idx];
    struct Col { float v[3]; uint32_t tex; };
    auto colour = [](const float v[3], uint32_t tex, bool clamp = true) {
        Col c;
        for (int k = 0; k < 3; ++k) c.v[k] = (clamp && v[k] < 0.f) ? 0.f : v[k]; // Spectrum::Clamp()
        c.tex = tex;
        return c;
    };
    auto black = [](const Col& c) { return !c.tex && c.v[0] == 0.f && c.v[1] == 0.f && c.v[2] == 0.f; }; // decided here only for constants
    auto remap = [&](float r) { // TrowbridgeReitzDistribution::RoughnessToAlpha, Enigma/MicrofacetDist.cpp:323-328
        if (!m.remap_roughness) return r;
        r = std::max(r, (float)1e-3);
        float x = std::log(r);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    };
    const float one[3] = {1.f, 1.f, 1.f};
    // a lobe whose colour is factor * c (and factor2 * c2); rule: how the lobe's existence follows from textured colours
    auto add = [&](uint32_t kind, const Col& c, const float* factor, uint32_t rule, float ax = 0.f, float ay = 0.f, float etaI = 1.f, float etaT = 1.f,
                   const Col* c2 = nullptr) {
        DevLobe l;
        memset(&l, 0, sizeof(l));
        l.kind = kind;
        l.ax = std::max(float(0.001), ax); l.ay = std::max(float(0.001), ay); // TrowbridgeReitzDistribution ctor, Enigma/MicrofacetDist.cpp:256-263
        for (int k = 0; k < 3; ++k) {
            l.col[k] = c.tex ? factor[k] : factor[k] * c.v[k];
            l.col2[k] = c2 ? c2->v[k] : 0.f;
            if (c2 && c2->tex) l.col2[k] = 1.f;
            l.scale[k] = 1.f;
        }
        l.etaI = etaI; l.etaT = etaT;
        const bool textured = c.tex || (c2 && c2->tex);
        l.tex = textured ? rule : 0u; // slots are filled in once the material's textures are known
        out.lobes.push_back(l);
        out.tex.push_back({c.tex, c2 ? c2->tex : 0u});
    };
    using namespace edev;
    *eta = 1.f;
    const Col kd = colour(m.kd, m.tex_kd), ks = colour(m.ks, m.tex_ks), kr = colour(m.kr, m.tex_kr), kt = colour(m.kt, m.tex_kt);
    switch (m.type) {
    case ENIGMA_MAT_MATTE: { // :76-93
        const float sig = std::min(std::max(m.sigma, 0.f), 90.f);
        if (black(kd)) break;
        if (sig == 0.f) { add(LOBE_LAMBERT, kd, one, kLobeOwn); break; }
        const float sigma = sig * (3.1415926535897932384f / 180.0f), sigma2 = sigma * sigma;
        add(LOBE_OREN_NAYAR, kd, one, kLobeOwn);
        out.lobes.back().ax = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f))); // A and B travel in the alpha slots
        out.lobes.back().ay = (0.45f * sigma2) / (sigma2 + 0.09f);
        break;
    }
    case ENIGMA_MAT_METAL: { // :115-136: eta and k are not clamped, the lobe is always there
        const Col e = colour(m.eta, m.tex_eta, false), k = colour(m.k, m.tex_k, false);
        add(LOBE_MICRO_REFL_CONDUCTOR, e, one, kLobeNoClamp, remap(m.uroughness), remap(m.vroughness), 1.f, 1.f, &k);
        break;
    }
    case ENIGMA_MAT_GLASS: { // :176-220
        *eta = m.index;
        if (black(kr) && black(kt)) break;
        const bool isSpecular = m.uroughness == 0.f && m.vroughness == 0.f;
        if (isSpecular && multipleLobes) { add(LOBE_FRESNEL_SPECULAR, kr, one, kLobeEither, 0.f, 0.f, 1.f, m.index, &kt); break; }
        const float ax = isSpecular ? 0.f : remap(m.uroughness), ay = isSpecular ? 0.f : remap(m.vroughness);
        if (!black(kr)) add(isSpecular ? LOBE_SPEC_REFL : LOBE_MICRO_REFL_DIELECTRIC, kr, one, kLobeOwn, ax, ay, 1.f, m.index);
        if (!black(kt)) add(isSpecular ? LOBE_SPEC_TRANS : LOBE_MICRO_TRANS, kt, one, kLobeOwn, ax, ay, 1.f, m.index);
        break;
    }
    case ENIGMA_MAT_PLASTIC: // :232-255
        if (!black(kd)) add(LOBE_LAMBERT, kd, one, kLobeOwn);
        if (!black(ks)) add(LOBE_MICRO_REFL_DIELECTRIC, ks, one, kLobeOwn, remap(m.uroughness), remap(m.uroughness), 1.5f, 1.f);
        break;
    case ENIGMA_MAT_MIRROR: // :318-328
        if (!black(kr)) add(LOBE_SPEC_REFL_NOOP, kr, one, kLobeOwn);
        break;
    case ENIGMA_MAT_TRANSLUCENT: { // :276-309: Kd and Ks are tested for black by themselves, then multiplied by reflect / transmit
        *eta = 1.5f;
        const Col r = colour(m.reflect, 0), t = colour(m.transmit, 0);
        if (black(r) && black(t)) break;
        if (!black(kd)) {
            if (!black(r)) add(LOBE_LAMBERT, kd, r.v, kLobeOwn | kLobeRaw);
            if (!black(t)) add(LOBE_LAMBERT_TRANS, kd, t.v, kLobeOwn | kLobeRaw);
        }
        if (!black(ks)) {
            const float a = remap(m.uroughness);
            if (!black(r)) add(LOBE_MICRO_REFL_DIELECTRIC, ks, r.v, kLobeOwn | kLobeRaw, a, a, 1.f, 1.5f);
            if (!black(t)) add(LOBE_MICRO_TRANS, ks, t.v, kLobeOwn | kLobeRaw, a, a, 1.f, 1.5f);
        }
        break;
    }
    case ENIGMA_MAT_SUBSTRATE: // :424-448
        if (!black(kd) || !black(ks)) add(LOBE_FRESNEL_BLEND, kd, one, kLobeEither, remap(m.uroughness), remap(m.vroughness), 1.f, 1.f, &ks);
        break;
    case ENIGMA_MAT_UBER: { // :536-592: every colour is multiplied by the opacity first and tested for black then
        const float e = m.index;
        const Col op = colour(m.opacity, 0);
        Col t = op;
        for (int k = 0; k < 3; ++k) t.v[k] = std::max(-op.v[k] + 1.f, 0.f);
        if (!black(t)) add(LOBE_SPEC_TRANS, t, one, 0u, 0.f, 0.f, 1.f, 1.f);
        else *eta = e;
        auto lit = [&](const Col& c) { // op * c is not black (a textured c may be anything)
            if (c.tex) return !black(op);
            return op.v[0] * c.v[0] != 0.f || op.v[1] * c.v[1] != 0.f || op.v[2] * c.v[2] != 0.f;
        };
        if (lit(kd)) add(LOBE_LAMBERT, kd, op.v, kLobeOwn);
        if (lit(ks)) add(LOBE_MICRO_REFL_DIELECTRIC, ks, op.v, kLobeOwn, remap(m.uroughness), remap(m.vroughness), 1.f, e);
        if (lit(kr)) add(LOBE_SPEC_REFL, kr, op.v, kLobeOwn, 0.f, 0.f, 1.f, e);
        if (lit(kt)) add(LOBE_SPEC_TRANS, kt, op.v, kLobeOwn, 0.f, 0.f, 1.f, e);
        break;
    }
    case ENIGMA_MAT_MIX: { // :146-163: the first material's BSDF, its lobes scaled; then the second material's lobes, scaled
        if (m.mix_a >= idx || m.mix_b >= idx) { why = "a mix material names materials that come before it in the array"; return false; }
        const Col s1 = colour(m.amount, 0);
        float s2[3], etaB;
        for (int k = 0; k < 3; ++k) s2[k] = std::max(1.f - s1.v[k], 0.f);
        const size_t first = out.lobes.size();
        if (!materialLobes(table, m.mix_a, multipleLobes, out, eta, why)) return false;
        const size_t second = out.lobes.size();
        if (!materialLobes(table, m.mix_b, multipleLobes, out, &etaB, why)) return false;
        for (size_t i = first; i < out.lobes.size(); ++i) {
            for (int k = 0; k < 3; ++k) out.lobes[i].scale[k] *= (i < second ? s1.v : s2)[k];
            out.lobes[i].scaled = 1u;
        }
        break;
    }
    default:
        why = "unknown material type";
        return false;
    }
    if (out.lobes.size() > (size_t)edev::kMaxLobes) { why = "a material with more than 8 lobes: the reference's BSDF holds MAX_BxDFS = 8 (Enigma/BxDF.h:8)"; return false; }
    return true;
}
// the two lists (allowMultipleLobes true, then false) of material idx appended to lobes; d describes them
static bool generalMaterial(const EnigmaMaterial* table, uint32_t idx, std::vector<DevLobe>& lobes, DevMaterial& d, std::string& why) {
    LobeList multi, single;
    float eta = 1.f, eta2 = 1.f;
    if (!materialLobes(table, idx, true, multi, &eta, why) || !materialLobes(table, idx, false, single, &eta2, why)) return false;
    memset(&d, 0, sizeof(d));
    d.kind = MK_GENERAL;
    d.eta = eta;
    d.flags = multi.lobes.empty() ? 0u : 1u;
    d.lobeOff = (uint32_t)lobes.size();
    d.lobeInfo = (uint32_t)multi.lobes.size() | ((uint32_t)single.lobes.size() << 8);
    // the material's trilinear textures take the two slots k_texture fills per hit
    uint32_t slots[2] = {0u, 0u};
    for (LobeList* list : {&multi, &single})
        for (size_t i = 0; i < list->lobes.size(); ++i) {
            uint32_t bits = 0u;
            const uint32_t ids[2] = {list->tex[i].first, list->tex[i].second};
            for (int k = 0; k < 2; ++k) {
                if (!ids[k]) continue;
                int slot = -1;
                for (int q = 0; q < 2 && slot < 0; ++q)
                    if (slots[q] == ids[k] || slots[q] == 0u) { slots[q] = ids[k]; slot = q; }
                if (slot < 0) { why = "more than two trilinear textures on a mirror, translucent, substrate, uber or mix material (a mix counts those of the materials it names)"; return false; }
                bits |= (uint32_t)(slot + 1) << (8 * k);
            }
            list->lobes[i].tex = bits ? (list->lobes[i].tex | bits) : 0u;
        }
    d.tex0 = slots[0]; d.tex1 = slots[1];
    lobes.insert(lobes.end(), multi.lobes.begin(), multi.lobes.end());
    lobes.insert(lobes.end(), single.lobes.begin(), single.lobes.end());
    return true;
}
static bool isGeneralMaterial(uint32_t type) { return type >= ENIGMA_MAT_MIRROR && type <= ENIGMA_MAT_MIX; }

static int sceneUpload(EnigmaCtx* h, const EnigmaSceneDesc* s) {
    HIPCHK(h, hipSetDevice(h->device));
    if (s->n_triangles && (!s->vertices || !s->indices || !s->tri_material || !s->tri_light)) return h->fail(ENIGMA_E_INVALID, "null geometry array");
    if (s->n_triangles >= (1u << 28)) return h->fail(ENIGMA_E_INVALID, "too many triangles");
    if (s->n_materials == 0 && s->n_triangles) {
        bool anyMaterial = false;
        for (uint32_t t = 0; t < s->n_triangles && s->tri_material; ++t) anyMaterial = anyMaterial || s->tri_material[t] != ENIGMA_NO_MATERIAL;
        if (anyMaterial) return h->fail(ENIGMA_E_INVALID, "no materials");
    }
    if (s->n_materials >= (1u << 24)) return h->fail(ENIGMA_E_UNSUPPORTED, "more than 2^24 materials");
    for (size_t i = 0; i < (size_t)s->n_triangles * 3; ++i)
        if (s->indices[i] >= s->n_vertices) return h->fail(ENIGMA_E_INVALID, "vertex index %u out of range", s->indices[i]);
    // Coordinates must be finite and at most 2^60 in size: products of two of them (surface areas in the tree builder, the
    // edge functions of the triangle test) then stay finite in single precision.  Beyond that the reference's own
    // arithmetic overflows too, and nothing is ever hit.
    if (s->n_triangles)
        for (size_t i = 0; i < (size_t)s->n_vertices * 3; ++i)
            if (!(std::fabs(s->vertices[i]) <= 0x1p60f))
                return h->fail(ENIGMA_E_INVALID, "vertex %zu: coordinate %g is not finite or larger than 2^60", i / 3, (double)s->vertices[i]);
    // a scene has media when it lists any; triangles without the per-triangle arrays have no interface (-1 on both sides),
    // the spheres carry their own fields
    const bool haveMedia = s->n_media > 0;
    const bool triMedia = s->tri_medium_inside && s->tri_medium_outside;
    if ((s->tri_medium_inside != nullptr) != (s->tri_medium_outside != nullptr)) return h->fail(ENIGMA_E_INVALID, "tri_medium_inside and tri_medium_outside go together");
    if (triMedia && !haveMedia) return h->fail(ENIGMA_E_INVALID, "per-triangle medium arrays without media");
    if (haveMedia && s->n_media && !s->media) return h->fail(ENIGMA_E_INVALID, "null media array");
    if (s->n_media >= 0xfffu) return h->fail(ENIGMA_E_UNSUPPORTED, "more than 4094 media");
    // From here on the context's scene state is rewritten piece by piece.  It holds no scene until this upload has
    // succeeded: a failure part-way leaves host flags and device tables of two different scenes behind, and
    // enigma_render must refuse to run on that mixture.  (A description rejected above leaves the old scene in place.)
    h->haveScene = false;
    h->haveNoMaterial = false;
    for (uint32_t t = 0; t < s->n_triangles; ++t) {
        if (s->tri_material[t] == ENIGMA_NO_MATERIAL) {
            h->haveNoMaterial = true;
            if (s->tri_light[t] >= 0) return h->fail(ENIGMA_E_UNSUPPORTED, "triangle %u: an area light on a surface without a material is not supported", t);
        } else if (s->tri_material[t] >= s->n_materials) return h->fail(ENIGMA_E_INVALID, "material index out of range at triangle %u", t);
        if (s->tri_light[t] >= (int32_t)s->n_lights) return h->fail(ENIGMA_E_INVALID, "light index out of range at triangle %u", t);
        const uint32_t fl = s->tri_flags ? s->tri_flags[t] : 0u;
        if ((fl & ENIGMA_TRI_HAS_NORMALS) && !s->normals) return h->fail(ENIGMA_E_INVALID, "triangle %u is flagged to have normals but the normals array is null", t);
        if ((fl & ENIGMA_TRI_HAS_UVS) && !s->uvs) return h->fail(ENIGMA_E_INVALID, "triangle %u is flagged to have uvs but the uvs array is null", t);
        if ((fl & ENIGMA_TRI_HAS_NORMALS) && s->tri_light[t] >= 0 && emitterNormalSide(s, t) == 0)
            return h->fail(ENIGMA_E_UNSUPPORTED, "triangle %u: an area light whose vertex normals lie on both sides of (or in) its plane is not supported", t);
        if (triMedia && (s->tri_medium_inside[t] < -1 || s->tri_medium_inside[t] >= (int32_t)s->n_media || s->tri_medium_outside[t] < -1 ||
                          s->tri_medium_outside[t] >= (int32_t)s->n_media))
            return h->fail(ENIGMA_E_INVALID, "medium index out of range at triangle %u", t);
    }
    // image textures (ImageTexture::getTexture, Enigma/Texture.h:331-372): flip in y, pow(texel * scale, gamma), MipMap.
    // A texture without "trilinear" evaluates to its coarsest texel everywhere (Enigma/MipMap.h:380-395: the EWA path
    // takes the component count of its differential vectors for their length), so it becomes a constant here.
    if (s->n_textures && !s->textures) return h->fail(ENIGMA_E_INVALID, "null texture array");
    std::vector<enigma::Rgb> texConstant(s->n_textures);
    std::vector<uint32_t> texDevice(s->n_textures, 0u); // device index + 1 of the trilinear ones
    std::vector<DevTexture> devTex;
    std::vector<float4> texels;
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const EnigmaTexture& t = s->textures[i];
        if (!t.rgb || !t.width || !t.height) return h->fail(ENIGMA_E_INVALID, "texture %u has no texels", i);
        if (t.width > 32768u || t.height > 32768u) return h->fail(ENIGMA_E_UNSUPPORTED, "texture %u is %u x %u: sides up to 32768", i, t.width, t.height);
        if (t.wrap > 2u) return h->fail(ENIGMA_E_INVALID, "texture %u: unknown wrap mode %u", i, t.wrap);
        const int w = (int)t.width, hh = (int)t.height;
        std::vector<enigma::Rgb> px((size_t)w * hh);
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < w; ++x) {
                const float* c = &t.rgb[3 * ((size_t)(hh - 1 - y) * w + x)];
                px[(size_t)y * w + x] = enigma::Rgb{std::pow(c[0] * t.scale, t.gamma), std::pow(c[1] * t.scale, t.gamma), std::pow(c[2] * t.scale, t.gamma)};
            }
        enigma::EnvPyramid pyr;
        pyr.wrap = (int)t.wrap;
        pyr.build(w, hh, std::move(px));
        texConstant[i] = pyr.texel(pyr.levels() - 1, 0, 0);
        if (!t.trilinear) continue;
        DevTexture d{};
        d.w = (uint32_t)pyr.lw[0]; d.h = (uint32_t)pyr.lh[0]; d.nLevels = (uint32_t)pyr.levels(); d.wrap = t.wrap;
        d.su = t.su; d.sv = t.sv; d.du = t.du; d.dv = t.dv;
        for (int l = 0; l < pyr.levels(); ++l) {
            d.levelOff[l] = (uint32_t)texels.size();
            for (const enigma::Rgb& c : pyr.level[l]) texels.push_back(make_float4(c.r, c.g, c.b, 0.f));
        }
        devTex.push_back(d);
        texDevice[i] = (uint32_t)devTex.size();
    }
    h->haveTrilinear = !devTex.empty();
    // materials (Enigma/Material.cpp:76-255 parameter evaluation)
    std::vector<DevMaterial> mats(s->n_materials);
    for (int k = 0; k < MK_COUNT; ++k) { h->kindPresent[k] = false; h->kindTextured[k] = false; }
    h->smoothGlass = false;
    h->glassBranches = h->glassSpecular = h->deferredLobeChoice = false;
    std::vector<DevLobe> lobeTable;
    // for the lobe lists: the materials with their textures resolved -- a texture that evaluates to a constant folded into
    // the value, a trilinear one named by its device index + 1
    std::vector<EnigmaMaterial> folded(s->materials, s->materials + s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        EnigmaMaterial& fm = folded[i];
        struct Slot { uint32_t* tex; float* value; };
        const Slot slotsOf[6] = {{&fm.tex_kd, fm.kd}, {&fm.tex_ks, fm.ks}, {&fm.tex_kr, fm.kr}, {&fm.tex_kt, fm.kt}, {&fm.tex_eta, fm.eta}, {&fm.tex_k, fm.k}};
        for (const Slot& sl : slotsOf) {
            if (!*sl.tex) continue;
            if (*sl.tex > s->n_textures) return h->fail(ENIGMA_E_INVALID, "material %u: texture index out of range", i);
            const uint32_t dev = texDevice[*sl.tex - 1];
            if (!dev) { const enigma::Rgb c = texConstant[*sl.tex - 1]; sl.value[0] = c.r; sl.value[1] = c.g; sl.value[2] = c.b; }
            *sl.tex = dev;
        }
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        EnigmaMaterial m = s->materials[i];
        DevMaterial d;
        memset(&d, 0, sizeof(d));
        if (isGeneralMaterial(m.type)) {
            std::string why;
            if (!generalMaterial(folded.data(), i, lobeTable, d, why)) return h->fail(ENIGMA_E_UNSUPPORTED, "material %u: %s", i, why.c_str());
            if (d.tex0 | d.tex1) h->kindTextured[MK_GENERAL] = true;
            // the DirectLightingIntegrator's recursion follows the specular lobes of the allowMultipleLobes = false list
            int specR = 0, specT = 0;
            for (uint32_t k = 0; k < (d.lobeInfo >> 8); ++k) {
                const uint32_t kind = lobeTable[d.lobeOff + (d.lobeInfo & 0xffu) + k].kind;
                specR += (kind == edev::LOBE_SPEC_REFL || kind == edev::LOBE_SPEC_REFL_NOOP) ? 1 : 0;
                specT += kind == edev::LOBE_SPEC_TRANS ? 1 : 0;
            }
            if (specR || specT) h->glassSpecular = h->smoothGlass = true;
            if (specR && specT) h->glassBranches = true;
            if (specR && specT > 1) h->deferredLobeChoice = true;
            h->kindPresent[MK_GENERAL] = true;
            mats[i] = d;
            continue;
        }
        // bound textures: constants for the EWA ones, a device texture for the trilinear ones (first / second colour of the kind)
        struct Bind { uint32_t tex; float* value; int slot; };
        const Bind binds[6] = {{m.tex_kd, m.kd, 0}, {m.tex_ks, m.ks, 1}, {m.tex_kr, m.kr, 0}, {m.tex_kt, m.kt, 1}, {m.tex_eta, m.eta, 0}, {m.tex_k, m.k, 1}};
        const bool used[6] = {m.type == ENIGMA_MAT_MATTE || m.type == ENIGMA_MAT_PLASTIC, m.type == ENIGMA_MAT_PLASTIC, m.type == ENIGMA_MAT_GLASS,
                              m.type == ENIGMA_MAT_GLASS, m.type == ENIGMA_MAT_METAL, m.type == ENIGMA_MAT_METAL};
        for (int b = 0; b < 6; ++b) {
            if (!binds[b].tex || !used[b]) continue;
            if (binds[b].tex > s->n_textures) return h->fail(ENIGMA_E_INVALID, "material %u: texture index out of range", i);
            const enigma::Rgb c = texConstant[binds[b].tex - 1];
            binds[b].value[0] = c.r; binds[b].value[1] = c.g; binds[b].value[2] = c.b;
            (binds[b].slot ? d.tex1 : d.tex0) = texDevice[binds[b].tex - 1];
        }
        if (!convertMaterial(m, d)) return h->fail(ENIGMA_E_UNSUPPORTED, "material type %u is not supported", m.type);
        bool nb0 = d.c0[0] != 0.f || d.c0[1] != 0.f || d.c0[2] != 0.f;
        bool nb1 = d.c1[0] != 0.f || d.c1[1] != 0.f || d.c1[2] != 0.f;
        if (d.tex0) nb0 = true; // decided per hit
        if (d.tex1) nb1 = true;
        d.flags = (nb0 ? 1u : 0u) | (nb1 ? 2u : 0u);
        if (d.kind == MK_GLASS_SMOOTH && (nb0 || nb1)) h->glassSpecular = true;
        if (d.kind == MK_GLASS_SMOOTH && nb0 && nb1) h->glassBranches = true;
        if (d.kind == MK_GLASS_SMOOTH) h->smoothGlass = true;
        if (d.tex0 | d.tex1) h->kindTextured[d.kind] = true;
        h->kindPresent[d.kind] = true;
        mats[i] = d;
    }
    // spheres (Enigma/ShapeImp.cpp:13-27): leaf entries of the BVH beside the triangles
    std::vector<DevSphere> sphereTable;
    if ((uint64_t)s->n_triangles + s->n_spheres >= (1u << 28)) return h->fail(ENIGMA_E_INVALID, "too many primitives");
    if (s->n_spheres && !s->spheres) return h->fail(ENIGMA_E_INVALID, "null sphere array");
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const EnigmaSphere& e = s->spheres[i];
        const bool noMat = e.material == ENIGMA_NO_MATERIAL; // a medium boundary, as for triangles
        if (!noMat && e.material >= s->n_materials) return h->fail(ENIGMA_E_INVALID, "sphere %u: material index out of range", i);
        if (noMat && e.light) return h->fail(ENIGMA_E_UNSUPPORTED, "sphere %u: an area light on a surface without a material is not supported", i);
        if (noMat) h->haveNoMaterial = true;
        if (!(e.radius > 0.f)) return h->fail(ENIGMA_E_INVALID, "sphere %u: radius must be positive", i);
        for (int c = 0; c < 4; ++c)
            if (e.object_to_world[12 + c] != (c == 3 ? 1.f : 0.f) || e.world_to_object[12 + c] != (c == 3 ? 1.f : 0.f))
                return h->fail(ENIGMA_E_UNSUPPORTED, "sphere %u: projective transforms are not supported", i);
        DevSphere d;
        memcpy(d.o2w, e.object_to_world, 48); memcpy(d.w2o, e.world_to_object, 48);
        auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
        d.radius = e.radius;
        d.zMin = clampf(std::min(e.z_min, e.z_max), -e.radius, e.radius);
        d.zMax = clampf(std::max(e.z_min, e.z_max), -e.radius, e.radius);
        d.thetaMin = std::acos(clampf(d.zMin / d.radius, -1.f, 1.f));
        d.thetaMax = std::acos(clampf(d.zMax / d.radius, -1.f, 1.f));
        d.phiMax = (3.14159265358979323846f / 180.f) * clampf(e.phi_max, 0.0f, 360.0f);
        d.mw = (noMat ? (uint32_t)MK_NONE << 24 : e.material | (mats[e.material].kind << 24)) | ((e.flags & ENIGMA_TRI_FLIP_NORMAL) ? 1u << 31 : 0u) | ((e.flags & ENIGMA_TRI_REVERSE_ORIENTATION) ? 1u << 29 : 0u);
        if (e.light > s->n_lights || (e.light && (s->lights[e.light - 1].type != ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE || s->lights[e.light - 1].triangle != i)))
            return h->fail(ENIGMA_E_INVALID, "sphere %u: its light index is out of range or names no light on this sphere", i);
        d.light = e.light;
        sphereTable.push_back(d);
    }
    // Shape::worldBound: the eight corners of objectBound() through m_objectToWorld (Enigma/Transform.h:82-94, transformPoint :36-45)
    auto sphereWorldBound = [](const DevSphere& sp, float* lo, float* hi) {
        const float olo[3] = {-sp.radius, -sp.radius, sp.zMin}, ohi[3] = {sp.radius, sp.radius, sp.zMax};
        for (int k = 0; k < 8; ++k) {
            const float x = (k & 1) ? ohi[0] : olo[0], y = (k & 2) ? ohi[1] : olo[1], z = (k & 4) ? ohi[2] : olo[2];
            for (int c = 0; c < 3; ++c) {
                const float* m = &sp.o2w[4 * c];
                const float q = m[0] * x + m[1] * y + m[2] * z + m[3];
                lo[c] = std::min(lo[c], q); hi[c] = std::max(hi[c], q);
            }
        }
    };
    // BVH + reorder of the primitives.  A sphere enters the tree with its world bound, widened by a few ulps of its size
    // and distance from the origin: the hit points that Sphere::intersect accepts carry rounding errors of that order.
    auto t0 = std::chrono::steady_clock::now();
    std::vector<float> sphereBoxes((size_t)s->n_spheres * 6);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        sphereWorldBound(sphereTable[i], lo, hi);
        for (int c = 0; c < 3; ++c) {
            if (!(std::fabs(lo[c]) <= 0x1p60f) || !(std::fabs(hi[c]) <= 0x1p60f)) // a NaN / infinite radius or matrix entry, or a sphere out of the range vertices may have
                return h->fail(ENIGMA_E_INVALID, "sphere %u: its world bound is not finite or reaches beyond 2^60", i);
            const float pad = 1e-5f * ((hi[c] - lo[c]) + std::max(std::fabs(lo[c]), std::fabs(hi[c])));
            sphereBoxes[6 * (size_t)i + c] = lo[c] - pad; sphereBoxes[6 * (size_t)i + 3 + c] = hi[c] + pad;
        }
    }
    enigma::Bvh8Result bvh;
    enigma::build_bvh8(s->vertices, s->indices, s->n_triangles, sphereBoxes.data(), s->n_spheres, &bvh);
    h->msBvhBuild = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (bvh.maxDepth > 64) return h->fail(ENIGMA_E_UNSUPPORTED, "BVH of %u levels (the traversal stacks hold 64)", bvh.maxDepth);
    h->bvhDepth = bvh.maxDepth;
    h->bvhStack = bvh.maxDepth; // one stack entry per level
    h->bvhNodes = (uint32_t)bvh.nodes.size();
    // the triangle array in tree order: three float4 per slot; a sphere's slot holds its index (sign bit of the first word set)
    const uint32_t nSlots = s->n_triangles + s->n_spheres;
    std::vector<float4> triData((size_t)std::max(1u, nSlots) * 3, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> newPos(s->n_triangles);
    std::vector<float4> shadeData; // filled when some triangle has per-vertex normals or uvs
    for (uint32_t k = 0; k < nSlots; ++k) {
        uint32_t t = bvh.triOrder[k];
        if (t >= s->n_triangles) {
            const uint32_t w[3] = {0x80000000u | (t - s->n_triangles), sphereTable[t - s->n_triangles].mw, 0xffffffffu};
            for (int v = 0; v < 3; ++v) { float wf; memcpy(&wf, &w[v], 4); triData[(size_t)k * 3 + v] = make_float4(0.f, 0.f, 0.f, wf); }
            continue;
        }
        newPos[t] = k;
        const uint32_t fl = s->tri_flags ? s->tri_flags[t] : 0u;
        bool flip = (fl & ENIGMA_TRI_FLIP_NORMAL) != 0;
        if ((fl & ENIGMA_TRI_HAS_NORMALS) && s->tri_light[t] >= 0) // what FaceForward(n, shading.n) leaves, shading.n = -ns under ReverseOrientation;
            flip = (emitterNormalSide(s, t) < 0) != ((fl & ENIGMA_TRI_REVERSE_ORIENTATION) != 0); // the emission tests read it, the shading path turns n itself
        const uint32_t bits = (flip ? 1u << 31 : 0u) | ((fl & ENIGMA_TRI_HAS_NORMALS) ? 1u << 27 : 0u) |
                              ((fl & ENIGMA_TRI_HAS_UVS) ? 1u << 28 : 0u) | ((fl & ENIGMA_TRI_REVERSE_ORIENTATION) ? 1u << 29 : 0u);
        const bool noMat = s->tri_material[t] == ENIGMA_NO_MATERIAL;
        uint32_t w[3] = {t, noMat ? ((uint32_t)MK_NONE << 24) | bits : s->tri_material[t] | (mats[s->tri_material[t]].kind << 24) | bits, (uint32_t)s->tri_light[t]};
        if (fl & (ENIGMA_TRI_HAS_NORMALS | ENIGMA_TRI_HAS_UVS)) {
            if (shadeData.empty()) shadeData.assign((size_t)nSlots * 4, make_float4(0, 0, 0, 0));
            float4* sd = &shadeData[(size_t)k * 4];
            float uvw[6] = {0, 0, 0, 0, 0, 0};
            for (int v = 0; v < 3; ++v) {
                const size_t vi = s->indices[3 * (size_t)t + v];
                if (fl & ENIGMA_TRI_HAS_UVS) { uvw[v] = s->uvs[2 * vi]; uvw[3 + v] = s->uvs[2 * vi + 1]; }
                const float* n = (fl & ENIGMA_TRI_HAS_NORMALS) ? &s->normals[3 * vi] : nullptr;
                sd[v] = make_float4(n ? n[0] : 0.f, n ? n[1] : 0.f, n ? n[2] : 0.f, uvw[v]);
            }
            sd[3] = make_float4(uvw[3], uvw[4], uvw[5], 0.f);
        }
        for (int v = 0; v < 3; ++v) {
            const float* p = &s->vertices[3 * (size_t)s->indices[3 * (size_t)t + v]];
            float wf;
            memcpy(&wf, &w[v], 4);
            triData[(size_t)k * 3 + v] = make_float4(p[0], p[1], p[2], wf);
        }
    }
    std::vector<float4> nodeData(bvh.nodes.size() * kNode8Vec4);
    memcpy(nodeData.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(enigma::Bvh8Node));
    std::vector<uint32_t> prim(std::max(1u, nSlots)); // by the caller's primitive number: triangles, then spheres
    for (uint32_t t = 0; t < s->n_triangles; ++t) prim[t] = s->tri_prim_id ? s->tri_prim_id[t] : t + 1;
    for (uint32_t i = 0; i < s->n_spheres; ++i) prim[s->n_triangles + i] = s->spheres[i].prim_id ? s->spheres[i].prim_id : s->n_triangles + i + 1;
    // lights (Enigma/Lights.cpp:89-129,560-651); Power().y() feeds the light distribution
    h->hostLights.assign(s->n_lights, DevLight());
    h->haveAreaLights = false;
    h->directDirty = true;
    h->lightPower.assign(s->n_lights, 0.f);
    const float PI = 3.1415926535897932384f;
    // Scene::worldBound() = the union of the primitives' bounds; BBox3::getBoundingSphere (Enigma/BBox.h:322-326)
    float worldRadius = 0.f;
    if (s->n_triangles || s->n_spheres) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = 0; k < (size_t)s->n_triangles * 3; ++k) {
            const float* p = &s->vertices[3 * (size_t)s->indices[k]];
            for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], p[c]); hi[c] = std::max(hi[c], p[c]); }
        }
        for (const DevSphere& sp : sphereTable) sphereWorldBound(sp, lo, hi);
        for (int c = 0; c < 3; ++c) { h->worldLo[c] = lo[c]; h->worldHi[c] = hi[c]; }
        float ctr[3], d2 = 0.f;
        bool inside = true;
        for (int c = 0; c < 3; ++c) { ctr[c] = (lo[c] * .5f) + (hi[c] * .5f); inside = inside && ctr[c] >= lo[c] && ctr[c] <= hi[c]; }
        for (int c = 0; c < 3; ++c) d2 += (ctr[c] - hi[c]) * (ctr[c] - hi[c]);
        worldRadius = inside ? std::sqrt(d2) : 0.f;
    }
    for (auto& b : h->envTexels) b->release();
    for (auto& b : h->envFloats) b->release();
    h->envTexels.clear(); h->envFloats.clear();
    std::vector<DevEnvMap> envTable;
    std::vector<uint32_t> infinite;
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const EnigmaLight& l = s->lights[i];
        DevLight d;
        memset(&d, 0, sizeof(d));
        for (int c = 0; c < 3; ++c) { d.L[c] = l.L[c]; d.pos[c] = l.pos[c]; }
        d.nSamples = l.n_samples ? l.n_samples : 1u;
        const float YW[3] = {0.212671f, 0.715160f, 0.072169f};
        if (l.type == ENIGMA_LIGHT_DIFFUSE_AREA) {
            if (l.triangle >= s->n_triangles) return h->fail(ENIGMA_E_INVALID, "light %u: triangle out of range", i);
            if (s->tri_light[l.triangle] != (int32_t)i) return h->fail(ENIGMA_E_INVALID, "light %u: triangle %u does not point back at it", i, l.triangle);
            h->haveAreaLights = true;
            d.type = 1; d.tri = newPos[l.triangle]; d.twoSided = l.two_sided ? 1u : 0u;
            {   // the normal of a sampled point: reversed by the orientation flags, or turned to the vertex normals' side (Enigma/TriangleMesh.cpp:615-625)
                const uint32_t fl = s->tri_flags ? s->tri_flags[l.triangle] : 0u;
                const bool sampleFlip = (fl & ENIGMA_TRI_HAS_NORMALS) ? emitterNormalSide(s, l.triangle) < 0 : (fl & ENIGMA_TRI_FLIP_NORMAL) != 0;
                d.sampleFlip = sampleFlip ? 1.f : 0.f;
            }
            const float* a = &s->vertices[3 * (size_t)s->indices[3 * (size_t)l.triangle]];
            const float* b = &s->vertices[3 * (size_t)s->indices[3 * (size_t)l.triangle + 1]];
            const float* c = &s->vertices[3 * (size_t)s->indices[3 * (size_t)l.triangle + 2]];
            // Triangle::area: Length(Cross(v1 - v0, v2 - v0)) * 0.5f (Enigma/TriangleMesh.cpp:438-446)
            float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float cx = e1[1] * e2[2] - e2[1] * e1[2], cy = e1[2] * e2[0] - e2[2] * e1[0], cz = e1[0] * e2[1] - e2[0] * e1[1];
            d.area = std::sqrt(cx * cx + cy * cy + cz * cz) * 0.5f;
            float scale = d.twoSided ? 2.0f : 1.0f;
            float k = d.area * PI * scale; // DiffuseAreaLight::Power, Enigma/Lights.cpp:647-651
            h->lightPower[i] = YW[0] * (d.L[0] * k) + YW[1] * (d.L[1] * k) + YW[2] * (d.L[2] * k);
        } else if (l.type == ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE) {
            if (l.triangle >= s->n_spheres || s->spheres[l.triangle].light != i + 1) return h->fail(ENIGMA_E_INVALID, "light %u: sphere %u does not point back at it", i, l.triangle);
            h->haveAreaLights = true;
            const DevSphere& sp = sphereTable[l.triangle];
            d.type = 5; d.tri = l.triangle; d.twoSided = l.two_sided ? 1u : 0u;
            d.area = sp.phiMax * sp.radius * (sp.zMax - sp.zMin); // Sphere::area, Enigma/ShapeImp.cpp:205-208
            float scale = d.twoSided ? 2.0f : 1.0f;
            float k = d.area * PI * scale; // DiffuseAreaLight::Power, Enigma/Lights.cpp:647-651
            h->lightPower[i] = YW[0] * (d.L[0] * k) + YW[1] * (d.L[1] * k) + YW[2] * (d.L[2] * k);
        } else if (l.type == ENIGMA_LIGHT_POINT) {
            d.type = 0;
            float k = 4.0f * PI; // PointLight::Power, Enigma/Lights.cpp:126-129
            h->lightPower[i] = YW[0] * (d.L[0] * k) + YW[1] * (d.L[1] * k) + YW[2] * (d.L[2] * k);
        } else if (l.type == ENIGMA_LIGHT_PROJECTION) {
            // no projection map ever exists in the reference (Enigma/Lights.cpp:220-221): a point light to every kernel ...
            d.type = 0;
            float P[3]; // ... with ProjectionLight::Power, Enigma/Lights.cpp:291-296, evaluated left to right
            for (int c = 0; c < 3; ++c) P[c] = 1.f * d.L[c] * 2.0f * PI * (1.f - l.cos_total_width);
            h->lightPower[i] = YW[0] * P[0] + YW[1] * P[1] + YW[2] * P[2];
        } else if (l.type == ENIGMA_LIGHT_GONIOMETRIC) {
            // scale() looks its map up with filter width 1: the texel at the top of the pyramid, for every direction
            // (Enigma/Lights.cpp:319-329, Enigma/MipMap.h:355-361): a point light of intensity I x that texel
            enigma::Rgb top{1.f, 1.f, 1.f}, half{1.f, 1.f, 1.f};
            if (l.map_rgb) {
                if (!l.map_width || !l.map_height) return h->fail(ENIGMA_E_INVALID, "light %u: empty goniometric map", i);
                if ((uint64_t)l.map_width * l.map_height > (1u << 26)) return h->fail(ENIGMA_E_UNSUPPORTED, "light %u: goniometric map too large", i);
                std::vector<enigma::Rgb> tex((size_t)l.map_width * l.map_height);
                for (size_t k = 0; k < tex.size(); ++k) tex[k] = enigma::Rgb{l.map_rgb[3 * k], l.map_rgb[3 * k + 1], l.map_rgb[3 * k + 2]};
                enigma::EnvPyramid pyr;
                pyr.build((int)l.map_width, (int)l.map_height, tex);
                top = pyr.lookup(.5f, .5f, 1.f);
                half = pyr.lookup(.5f, .5f, .5f);
            }
            d.type = 0;
            d.L[0] = l.L[0] * top.r; d.L[1] = l.L[1] * top.g; d.L[2] = l.L[2] * top.b;
            const float k = 4.f * PI * PI; // GonioPhotometricLight::Power, Enigma/Lights.cpp:359-363: the intensity is not in it
            h->lightPower[i] = YW[0] * (k * half.r) + YW[1] * (k * half.g) + YW[2] * (k * half.b);
        } else if (l.type == ENIGMA_LIGHT_SPOT) {
            d.type = 2;
            d.cosTotalWidth = l.cos_total_width; d.cosFalloffStart = l.cos_falloff_start;
            for (int c = 0; c < 3; ++c) { d.w2l0[c] = l.world_to_light[c]; d.w2l1[c] = l.world_to_light[3 + c]; d.w2l2[c] = l.world_to_light[6 + c]; }
            float P[3]; // SpotLight::Power, Enigma/Lights.cpp:174-178, evaluated left to right
            for (int c = 0; c < 3; ++c) P[c] = d.L[c] * 2.f * PI * (1.f - .5f * (d.cosFalloffStart + d.cosTotalWidth));
            h->lightPower[i] = YW[0] * P[0] + YW[1] * P[1] + YW[2] * P[2];
        } else if (l.type == ENIGMA_LIGHT_DISTANT) {
            d.type = 3;
            for (int c = 0; c < 3; ++c) d.pos[c] = l.dir[c];
            d.worldRadius = worldRadius;
            float P[3]; // DistantLight::Power, Enigma/Lights.cpp:705-708: m_L * PI * r * r, evaluated left to right
            for (int c = 0; c < 3; ++c) P[c] = d.L[c] * PI * worldRadius * worldRadius;
            h->lightPower[i] = YW[0] * P[0] + YW[1] * P[1] + YW[2] * P[2];
        } else if (l.type == ENIGMA_LIGHT_INFINITE) {
            if (l.map_rgb && (!l.map_width || !l.map_height)) return h->fail(ENIGMA_E_INVALID, "light %u: empty radiance map", i);
            if (l.map_rgb && (uint64_t)l.map_width * l.map_height > (1u << 26)) return h->fail(ENIGMA_E_UNSUPPORTED, "light %u: radiance map too large", i);
            enigma::HostEnvMap em;
            em.build(l.map_rgb, (int)l.map_width, (int)l.map_height, l.L);
            d.type = 4; d.tri = (uint32_t)envTable.size();
            d.worldRadius = worldRadius;
            for (int c = 0; c < 3; ++c) {
                d.w2l0[c] = l.world_to_light[c]; d.w2l1[c] = l.world_to_light[3 + c]; d.w2l2[c] = l.world_to_light[6 + c];
                d.l2w0[c] = l.light_to_world[c]; d.l2w1[c] = l.light_to_world[3 + c]; d.l2w2[c] = l.light_to_world[6 + c];
            }
            h->haveAreaLights = true; // not a delta light: EstimateDirect traces its BSDF-sampled ray
            DevEnvMap t{};
            t.w = em.w; t.h = em.h; t.nu = em.nu; t.nv = em.nv; t.margInt = em.margInt;
            {
                std::vector<float4> tex((size_t)em.w * em.h);
                memcpy(tex.data(), em.texels.data(), tex.size() * sizeof(float4));
                h->envTexels.emplace_back(new DevBuf<float4>());
                HIPCHK(h, h->envTexels.back()->upload(tex));
                t.texels = h->envTexels.back()->p;
            }
            const std::vector<float>* arrays[] = {&em.condFunc, &em.condCdf, &em.condInt, &em.margFunc, &em.margCdf};
            const float** slots[] = {&t.condFunc, &t.condCdf, &t.condInt, &t.margFunc, &t.margCdf};
            for (int a = 0; a < 5; ++a) {
                h->envFloats.emplace_back(new DevBuf<float>());
                HIPCHK(h, h->envFloats.back()->upload(*arrays[a]));
                *slots[a] = h->envFloats.back()->p;
            }
            envTable.push_back(t);
            infinite.push_back(i);
            // InfiniteAreaLight::Power, Enigma/Lights.cpp:541-546: PI * r * r * lookup(half, half, half), left to right
            float k = PI * worldRadius * worldRadius;
            h->lightPower[i] = YW[0] * (k * em.powerLookup.r) + YW[1] * (k * em.powerLookup.g) + YW[2] * (k * em.powerLookup.b);
        } else return h->fail(ENIGMA_E_UNSUPPORTED, "light type %u is not supported", l.type);
        h->hostLights[i] = d;
    }
    // media (Enigma/Medium.cpp:69-76) and the triangles' medium interfaces in device order
    {
        std::vector<DevMedium> md(haveMedia ? s->n_media : 0);
        for (size_t i = 0; i < md.size(); ++i) {
            const EnigmaMedium& m = s->media[i];
            for (int c = 0; c < 3; ++c) { md[i].sigmaT[c] = m.sigma_s[c] + m.sigma_a[c]; md[i].sigmaS[c] = m.sigma_s[c]; }
            if (m.kind > ENIGMA_MEDIUM_GRID_DENSITY) return h->fail(ENIGMA_E_INVALID, "medium %zu: kind %u is not a medium kind", i, m.kind);
            md[i].g = m.g; md[i].opaque = m.kind == ENIGMA_MEDIUM_GRID_DENSITY ? 1.f : 0.f;
        }
        std::vector<uint32_t> tm(haveMedia ? std::max(1u, nSlots) + s->n_spheres : 0, 0u); // by primitive number on the device: slots, then spheres
        for (uint32_t i = 0; haveMedia && i < s->n_spheres; ++i) { // spheres follow the triangles; their ABI fields already hold index + 1
            const EnigmaSphere& e = s->spheres[i];
            if (e.medium_inside < 0 || e.medium_inside > (int32_t)s->n_media || e.medium_outside < 0 || e.medium_outside > (int32_t)s->n_media)
                return h->fail(ENIGMA_E_INVALID, "sphere %u: medium index out of range", i);
            tm[std::max(1u, nSlots) + i] = (uint32_t)e.medium_inside | ((uint32_t)e.medium_outside << 16);
        }
        for (uint32_t k = 0; k < ((haveMedia && triMedia) ? nSlots : 0u); ++k) { // tm is sized by haveMedia
            uint32_t t = bvh.triOrder[k];
            if (t >= s->n_triangles) continue;
            tm[k] = (uint32_t)(s->tri_medium_inside[t] + 1) | ((uint32_t)(s->tri_medium_outside[t] + 1) << 16);
        }
        HIPCHK(h, h->media.upload(md));
        HIPCHK(h, h->triMedium.upload(tm));
        h->scene.media = h->media.p;
        h->scene.triMedium = haveMedia ? h->triMedium.p : nullptr;
    }
    HIPCHK(h, h->spheres.upload(sphereTable));
    h->scene.spheres = sphereTable.empty() ? nullptr : h->spheres.p;
    h->scene.nSpheres = (uint32_t)sphereTable.size();
    h->scene.sphereGeom = nullptr; // allocated with the path state
    HIPCHK(h, h->texTexels.upload(texels));
    for (DevTexture& d : devTex) d.texels = h->texTexels.p;
    HIPCHK(h, h->textures.upload(devTex));
    h->scene.textures = devTex.empty() ? nullptr : h->textures.p;
    h->scene.texColors = nullptr; // allocated with the path state
    HIPCHK(h, h->triShade.upload(shadeData));
    h->scene.triShade = shadeData.empty() ? nullptr : h->triShade.p;
    HIPCHK(h, h->envMaps.upload(envTable));
    HIPCHK(h, h->infiniteLights.upload(infinite));
    h->scene.envMaps = h->envMaps.p;
    h->scene.infiniteLights = h->infiniteLights.p;
    h->scene.nInfinite = (int)infinite.size();
    HIPCHK(h, h->nodes.upload(nodeData));
    HIPCHK(h, h->tris.upload(triData));
    HIPCHK(h, h->materials.upload(mats));
    if (lobeTable.empty()) lobeTable.resize(1); // never a null table behind an MK_GENERAL material without lobes
    HIPCHK(h, h->lobes.upload(lobeTable));
    h->scene.lobes = h->lobes.p;
    HIPCHK(h, h->lights.upload(h->hostLights));
    HIPCHK(h, h->primId.upload(prim));
    h->scene.nodes = h->nodes.p;
    h->scene.tris = h->tris.p;
    h->scene.materials = h->materials.p;
    h->scene.lights = h->lights.p;
    h->scene.primId = h->primId.p;
    h->scene.nLights = (int)s->n_lights;
    h->scene.nTris = std::max(1u, nSlots);
    h->scene.nNodes = h->bvhNodes;
    h->scene.nMaterials = s->n_materials;
    h->lightDistDirty = true;
    h->haveScene = true;
    return ENIGMA_OK;
}

extern "C" int enigma_scene_upload(EnigmaCtx* h, const EnigmaSceneDesc* s) {
    if (!h || !s) return ENIGMA_E_INVALID;
    return guarded(h, [&] { return sceneUpload(h, s); });
}

extern "C" int enigma_camera_set(EnigmaCtx* h, const EnigmaCameraDesc* c) {
    if (!h || !c) return ENIGMA_E_INVALID;
    memcpy(h->camera.r2c, c->raster_to_camera, 64);
    memcpy(h->camera.c2w, c->camera_to_world, 64);
    if (c->type > ENIGMA_CAMERA_ENVIRONMENT) return h->fail(ENIGMA_E_INVALID, "unknown camera type %u", c->type);
    h->camera.lensRadius = c->type == ENIGMA_CAMERA_ENVIRONMENT ? 0.f : c->lens_radius; // the environment camera has no lens: no lens samples are drawn for it
    h->camera.focalDistance = c->focal_distance;
    h->camera.type = c->type;
    h->haveCamera = true;
    return ENIGMA_OK;
}

extern "C" int enigma_film_clear(EnigmaCtx* h) {
    if (!h || !h->haveFilm) return h ? h->fail(ENIGMA_E_STATE, "film not configured") : ENIGMA_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->filmPixels.p, 0, h->filmPixels.n * sizeof(float4), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ENIGMA_OK;
}

extern "C" int enigma_film_config(EnigmaCtx* h, const EnigmaFilmDesc* f) {
    if (!h || !f) return ENIGMA_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    const int32_t* b = f->cropped_pixel_bounds;
    if (b[2] < b[0] || b[3] < b[1]) return h->fail(ENIGMA_E_INVALID, "inverted cropped pixel bounds");
    if (b[0] < 0 || b[1] < 0 || b[2] > 65535 || b[3] > 65535) return h->fail(ENIGMA_E_INVALID, "pixel bounds must lie in [0, 65535]");
    if (!(f->filter_radius[0] > 0.f) || !(f->filter_radius[1] > 0.f)) return h->fail(ENIGMA_E_INVALID, "filter radius must be positive");
    h->filmDesc = *f;
    size_t nPix = (size_t)(b[2] - b[0]) * (size_t)(b[3] - b[1]);
    HIPCHK(h, h->filmPixels.alloc(nPix));
    h->filmPixels.n = nPix;
    std::vector<float> tab(f->filter_table, f->filter_table + 256);
    HIPCHK(h, h->filterTable.upload(tab));
    h->film.x0 = b[0]; h->film.y0 = b[1]; h->film.x1 = b[2]; h->film.y1 = b[3];
    h->film.radiusX = f->filter_radius[0]; h->film.radiusY = f->filter_radius[1];
    h->film.invRadiusX = 1.0f / f->filter_radius[0]; h->film.invRadiusY = 1.0f / f->filter_radius[1]; // Inverted(), Enigma/Film.cpp:255
    h->film.maxSampleLuminance = f->max_sample_luminance;
    h->film.pixels = h->filmPixels.p;
    h->film.filterTable = h->filterTable.p;
    h->camera.resX = (float)f->full_resolution[0]; h->camera.resY = (float)f->full_resolution[1];
    h->haveFilm = true;
    return enigma_film_clear(h);
}

// Which sampler the integrators configured after this call clone per tile (Enigma/Api.cpp:425-434 selects by name).
extern "C" int enigma_sampler_select(EnigmaCtx* h, uint32_t kind, uint32_t sample_at_pixel_center) {
    if (!h) return ENIGMA_E_INVALID;
    if (kind != ENIGMA_SAMPLER_SOBOL && kind != ENIGMA_SAMPLER_HALTON) return h->fail(ENIGMA_E_INVALID, "unknown sampler kind %u", kind);
    h->samplerSelected = kind;
    h->centerSelected = kind == ENIGMA_SAMPLER_HALTON && sample_at_pixel_center != 0;
    return ENIGMA_OK;
}

// render region: the pixels the reference's tile loop visits (Enigma/Integrator.cpp:233-265).  Tile rows start
// at sampleBounds.m_min.x, not m_min.y (:254): with a crop window whose x and y origins differ, the rows
// [min.y, min.x) are never rendered.  Kept, because the film must match the reference's.
static void setRenderRegion(EnigmaCtx* h, const int32_t* sb, const int32_t* pixel_bounds) {
    int dy = sb[3] - sb[1];
    int nTilesY = (dy + 31) / 32;
    int tileY0 = sb[0], tileY1 = std::min(sb[0] + nTilesY * 32, sb[3]);
    int rx0 = std::max(sb[0], pixel_bounds[0]), ry0 = std::max(tileY0, pixel_bounds[1]);
    int rx1 = std::min(sb[2], pixel_bounds[2]), ry1 = std::min(tileY1, pixel_bounds[3]);
    h->integ.rx0 = rx0; h->integ.ry0 = ry0; h->integ.rw = std::max(0, rx1 - rx0); h->integ.rh = std::max(0, ry1 - ry0);
}

// What the kernels need of the sample numbers in flight: per slot the Sobol' constants of its sample number, or, for the
// HaltonSampler, the sample number itself.
static int uploadSampleSlots(EnigmaCtx* h, const int64_t* sampleNums, int64_t first, size_t count) {
    std::vector<uint32_t> consts;
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON) {
        consts.assign(count + 16, 0u);
        for (size_t s = 0; s < count; ++s) consts[s] = (uint32_t)(sampleNums ? sampleNums[s] : first + (int64_t)s);
    } else {
        consts.assign(count * h->nDims + 16, 0u);
        for (size_t s = 0; s < count; ++s)
            enigma::sample_constants(h->log2Res, (uint64_t)(sampleNums ? sampleNums[s] : first + (int64_t)s), h->nDims, &consts[s * h->nDims]);
    }
    HIPCHK(h, h->sampleConst.upload(consts));
    h->sampler.sampleConst = h->sampleConst.p;
    return ENIGMA_OK;
}

// SobolSampler / HaltonSampler construction + the pixels the tile loop visits; shared by the integrators.
static int configureSampler(EnigmaCtx* h, int64_t samples_per_pixel, const int32_t* sb, const int32_t* pixel_bounds, uint32_t nDims, uint32_t kind, bool center) {
    if (samples_per_pixel <= 0 || samples_per_pixel > (1ll << 30)) return h->fail(ENIGMA_E_INVALID, "samples_per_pixel out of range");
    if (sb[2] <= sb[0] || sb[3] <= sb[1]) return h->fail(ENIGMA_E_INVALID, "empty sample bounds");
    const int64_t dx64 = (int64_t)sb[2] - sb[0], dy64 = (int64_t)sb[3] - sb[1];
    if (dx64 > 65535 || dy64 > 65535) return h->fail(ENIGMA_E_INVALID, "sample bounds wider than 65535");
    int dx = (int)dx64, dy = (int)dy64;
    if (sb[0] < sb[1]) return h->fail(ENIGMA_E_UNSUPPORTED, "sample bounds with min.x < min.y make the reference's tile loop (Integrator.cpp:254) "
                                                          "start above the sample bounds; its result is undefined");
    // Arguments rejected above leave the configured integrator as it was.  From here on the sampler state is rewritten;
    // a failure below leaves it half rebuilt, so the handle has no integrator until the caller's configuration succeeds.
    h->haveInteg = false;
    h->samplerKind = kind; h->centerKind = center;
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON) {
        // HaltonSampler ctor, Enigma/Sampler.cpp:305-334: the sample count is kept as given
        // the VolPathIntegrator asks for every dimension its cursor can address (1023); the reference's prime table ends at
        // dimension 999 and it indexes past it beyond (undefined): those dimensions repeat the last one here
        if (nDims > 1023u) return h->fail(ENIGMA_E_INVALID, "dimension cursor overflow");
        h->spp = samples_per_pixel;
        h->resolution = 0; h->log2Res = 0;
        enigma::build_halton_tables(sb, nDims, center, &h->haltonSampler);
        // the device takes the digits of an index off with 32-bit divisions
        if (((uint64_t)h->spp + 1ull) * h->haltonSampler.stride > (1ull << 32))
            return h->fail(ENIGMA_E_UNSUPPORTED, "%lld samples per pixel at a sample stride of %u exceed the 32-bit Halton index of this build", (long long)h->spp, h->haltonSampler.stride);
        h->nDims = nDims;
        HIPCHK(h, h->tx.upload(h->haltonSampler.offX));
        HIPCHK(h, h->ty.upload(h->haltonSampler.offY));
        HIPCHK(h, h->bitConst.upload(h->haltonSampler.words));
        h->sampler.tx = h->tx.p; h->sampler.ty = h->ty.p; h->sampler.bitConst = h->bitConst.p;
        h->sampler.res = 0u; h->sampler.nDims = nDims; // res == 0 is how the kernels know the sampler
        h->sampler.sbx0 = sb[0]; h->sampler.sby0 = sb[1];
        h->sampler.resolutionF = 0.f;
        setRenderRegion(h, sb, pixel_bounds);
        return ENIGMA_OK;
    }
    // SobolSampler ctor, Enigma/Sampler.cpp:211-222
    h->spp = (int64_t)roundUpPow2((uint32_t)samples_per_pixel);
    h->resolution = roundUpPow2((uint32_t)std::max(dx, dy));
    h->log2Res = (uint32_t)log2Int((float)h->resolution);
    if ((1u << h->log2Res) != h->resolution) return h->fail(ENIGMA_E_INVALID, "Log2Int(resolution) is inexact for %u", h->resolution);
    if (nDims > (uint32_t)enigma::kNumSobolDimensions) return h->fail(ENIGMA_E_INVALID, "the integrator needs %u Sobol dimensions (> 1024)", nDims);
    if (nDims > 1023u) return h->fail(ENIGMA_E_INVALID, "dimension cursor overflow");
    // the Sobol' index of (pixel, sample number) has 2 * log2Res + log2(spp) bits; the matrices have 52 columns
    // (Enigma/SobolMatrices.h:6-11, Enigma/SamplerInclude.h:2627-2637)
    if (h->spp > (1ll << (52 - 2 * (int)h->log2Res)))
        return h->fail(ENIGMA_E_INVALID, "%lld samples per pixel at resolution %u exceed the 52-bit Sobol' index", (long long)h->spp, h->resolution);
    h->nDims = nDims;
    enigma::SobolPixelTables T;
    enigma::build_pixel_tables(h->log2Res, nDims, &T);
    // a kernel reads up to 5 dimensions past its cursor before the cursor is checked against anything: pad the tables
    T.tx.resize(T.tx.size() + (size_t)16 * h->resolution, 0u);
    T.ty.resize(T.ty.size() + (size_t)16 * h->resolution, 0u);
    HIPCHK(h, h->tx.upload(T.tx));
    HIPCHK(h, h->ty.upload(T.ty));
    // constant of sample number 1 << b per dimension; sample numbers stay below 2^(52 - 2 log2Res), the width of the
    // Sobol' matrices (Enigma/SamplerInclude.h:2627-2637 reads one matrix column per index bit)
    std::vector<uint32_t> bits((size_t)nDims * kSampleBits, 0u), tmp(nDims);
    int usable = std::min<int>(kSampleBits, 52 - 2 * (int)h->log2Res);
    for (int b = 0; b < usable; ++b) {
        enigma::sample_constants(h->log2Res, 1ull << b, nDims, tmp.data());
        for (uint32_t d = 0; d < nDims; ++d) bits[(size_t)d * kSampleBits + b] = tmp[d];
    }
    HIPCHK(h, h->bitConst.upload(bits));
    h->sampler.tx = h->tx.p; h->sampler.ty = h->ty.p; h->sampler.bitConst = h->bitConst.p;
    h->sampler.res = h->resolution; h->sampler.nDims = nDims;
    h->sampler.sbx0 = sb[0]; h->sampler.sby0 = sb[1];
    h->sampler.resolutionF = (float)(int)h->resolution;
    setRenderRegion(h, sb, pixel_bounds);
    return ENIGMA_OK;
}

static int integratorConfigImpl(EnigmaCtx* h, const EnigmaIntegratorDesc* d) {
    HIPCHK(h, hipSetDevice(h->device));
    if (d->max_depth < 0 || d->max_depth > 120) return h->fail(ENIGMA_E_INVALID, "max_depth out of range");
    if (d->light_strategy != ENIGMA_LIGHTS_UNIFORM && d->light_strategy != ENIGMA_LIGHTS_POWER && d->light_strategy != ENIGMA_LIGHTS_SPATIAL)
        return h->fail(ENIGMA_E_UNSUPPORTED, "light strategy %d is not supported (uniform, power and spatial are)", d->light_strategy);
    // dimensions a path can consume: 5 (camera) + per bounce 1+2+2 (light) + 2 (BSDF) + 1 (roulette)
    int rc = configureSampler(h, d->samples_per_pixel, d->sample_bounds, d->pixel_bounds, 5u + 8u * (uint32_t)d->max_depth, h->samplerSelected, h->centerSelected);
    if (rc) return rc;
    h->integDesc = *d;
    h->integ.maxDepth = d->max_depth;
    h->integ.rrThreshold = d->rr_threshold;
    h->integ.raysPerPath = 1;
    h->lightDistDirty = true;
    h->useDirect = false;
    h->useVolPath = false;
    h->haveInteg = true;
    return ENIGMA_OK;
}

static int volpathConfigImpl(EnigmaCtx* h, const EnigmaIntegratorDesc* d) {
    HIPCHK(h, hipSetDevice(h->device));
    // every ray segment inside a medium draws two more samples and boundaries do not count as bounces, so a path's
    // dimensions are not bounded by its depth: every dimension the 10-bit cursor can address is tabulated (the reference
    // aborts at dimension 1024)
    if (d->max_depth < 0 || d->max_depth > 60) return h->fail(ENIGMA_E_INVALID, "max_depth out of range");
    if (d->light_strategy != ENIGMA_LIGHTS_UNIFORM && d->light_strategy != ENIGMA_LIGHTS_POWER && d->light_strategy != ENIGMA_LIGHTS_SPATIAL)
        return h->fail(ENIGMA_E_UNSUPPORTED, "light strategy %d is not supported (uniform, power and spatial are)", d->light_strategy);
    int rc = configureSampler(h, d->samples_per_pixel, d->sample_bounds, d->pixel_bounds, 1023u, h->samplerSelected, h->centerSelected);
    if (rc) return rc;
    h->integDesc = *d;
    h->integ.maxDepth = d->max_depth;
    h->integ.rrThreshold = d->rr_threshold;
    h->integ.raysPerPath = 1;
    h->lightDistDirty = true;
    h->useDirect = false;
    h->useVolPath = true;
    h->haveInteg = true;
    return ENIGMA_OK;
}

static int directLightingConfigImpl(EnigmaCtx* h, const EnigmaDirectLightingDesc* d) {
    HIPCHK(h, hipSetDevice(h->device));
    if (d->max_depth < 0 || d->max_depth > 30) return h->fail(ENIGMA_E_INVALID, "max_depth out of range");
    if (d->strategy != ENIGMA_DL_SAMPLE_ALL && d->strategy != ENIGMA_DL_SAMPLE_ONE) return h->fail(ENIGMA_E_INVALID, "unknown strategy %d", d->strategy);
    if (d->samples_per_pixel <= 0 || d->samples_per_pixel > (1ll << 30)) return h->fail(ENIGMA_E_INVALID, "samples_per_pixel out of range");
    if (d->sample_bounds[2] <= d->sample_bounds[0] || d->sample_bounds[3] <= d->sample_bounds[1]) return h->fail(ENIGMA_E_INVALID, "empty sample bounds");
    h->haveInteg = false; // rewritten from here on; true again only if everything below succeeds
    h->directDesc = *d;
    h->integDesc.max_depth = d->max_depth;
    memcpy(h->integDesc.pixel_bounds, d->pixel_bounds, sizeof(d->pixel_bounds));
    memcpy(h->integDesc.sample_bounds, d->sample_bounds, sizeof(d->sample_bounds));
    h->integDesc.samples_per_pixel = d->samples_per_pixel;
    h->useDirect = true;
    h->useVolPath = false;
    h->directDirty = true; // finished against the scene at render time (Preprocess needs the lights)
    // the sampler part does not depend on the scene: report its errors now
    int rc = configureSampler(h, d->samples_per_pixel, d->sample_bounds, d->pixel_bounds, 5u, h->samplerSelected, h->centerSelected);
    h->haveInteg = rc == ENIGMA_OK;
    return rc;
}

extern "C" int enigma_integrator_config(EnigmaCtx* h, const EnigmaIntegratorDesc* d) {
    if (!h || !d) return ENIGMA_E_INVALID;
    return guarded(h, [&] { return integratorConfigImpl(h, d); });
}
extern "C" int enigma_volpath_config(EnigmaCtx* h, const EnigmaIntegratorDesc* d) {
    if (!h || !d) return ENIGMA_E_INVALID;
    return guarded(h, [&] { return volpathConfigImpl(h, d); });
}
extern "C" int enigma_direct_lighting_config(EnigmaCtx* h, const EnigmaDirectLightingDesc* d) {
    if (!h || !d) return ENIGMA_E_INVALID;
    return guarded(h, [&] { return directLightingConfigImpl(h, d); });
}

// DirectLightingIntegrator::Preprocess against the current scene (Enigma/DirectLightingIntegrator.cpp:71-87): the
// 2 x lights x maxDepth sample arrays become the reserved dimensions [5, arrayEndDim).
static int finishDirectLighting(EnigmaCtx* h) {
    const EnigmaDirectLightingDesc& d = h->directDesc;
    const uint32_t nLights = (uint32_t)h->hostLights.size();
    uint32_t nArrays = 0, raysPerPath = 1, maxN = 1;
    if (d.strategy == ENIGMA_DL_SAMPLE_ALL) {
        nArrays = 2u * nLights * (uint32_t)d.max_depth;
        raysPerPath = 0;
        for (const DevLight& l : h->hostLights) { raysPerPath += l.nSamples; maxN = std::max(maxN, l.nSamples); }
        raysPerPath = std::max(1u, raysPerPath);
        // Li evaluations of one camera sample exceed maxDepth only if the recursion branches; the reference then indexes
        // m_sampleArray2D past its end (Get2DArray's guard tests the wrong offset, Enigma/Sampler.cpp:62-69)
        if (h->glassBranches && d.max_depth >= 2 && nLights > 0)
            return h->fail(ENIGMA_E_UNSUPPORTED, "strategy 'all' with a smooth glass that has both Kr and Kt: the reference reads past the sample "
                                                 "arrays it requested (undefined); use strategy 'one'");
    }
    // Li evaluations (nodes) per camera sample: 1 without specular lobes, a chain with one, a binary tree with two
    uint64_t nodes = !h->glassSpecular ? 1 : (!h->glassBranches ? (uint64_t)std::max(1, d.max_depth) : ((1ull << std::min(d.max_depth, 20)) - 1));
    uint64_t nDims = 5ull + 2ull * nArrays + nodes * (d.strategy == ENIGMA_DL_SAMPLE_ONE ? 9ull : 4ull);
    const bool haltonNext = h->samplerKind == ENIGMA_SAMPLER_HALTON; // as configured (enigma_direct_lighting_config ran the sampler's part)
    if (nDims > (uint64_t)(haltonNext ? enigma::kHaltonDimensions : enigma::kNumSobolDimensions))
        return h->fail(ENIGMA_E_UNSUPPORTED, "DirectLightingIntegrator needs %llu sample dimensions here (%u sample arrays); the reference's %s sampler has %d",
                       (unsigned long long)nDims, nArrays, haltonNext ? "halton" : "sobol", haltonNext ? enigma::kHaltonDimensions : enigma::kNumSobolDimensions);
    int rc = configureSampler(h, d.samples_per_pixel, d.sample_bounds, d.pixel_bounds, (uint32_t)nDims, h->samplerKind, h->centerKind); // the sampler chosen when the integrator was configured
    if (rc) { h->haveInteg = false; return rc; } // the sampler tables are half rebuilt
    // entry k of a sample array of n entries is a sample of sample number (current sample number) * n + k (Enigma/Sampler.cpp:170-180)
    if (h->samplerKind == ENIGMA_SAMPLER_SOBOL && (uint64_t)h->spp * maxN > (1ull << std::min(40, 52 - 2 * (int)h->log2Res))) {
        h->haveInteg = false;
        return h->fail(ENIGMA_E_UNSUPPORTED, "sample arrays of %u entries x %lld samples per pixel exceed the Sobol' index width", maxN, (long long)h->spp);
    }
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON && ((uint64_t)h->spp * std::max(1u, maxN) + 1ull) * h->haltonSampler.stride > (1ull << 32)) {
        h->haveInteg = false;
        return h->fail(ENIGMA_E_UNSUPPORTED, "sample arrays of %u entries x %lld samples per pixel exceed the 32-bit Halton index of this build", maxN, (long long)h->spp);
    }
    h->integ.maxDepth = d.max_depth;
    h->integ.rrThreshold = 0.f;
    h->integ.dlStrategy = d.strategy;
    h->integ.arrayEndDim = 5 + 2 * (int)nArrays;
    h->integ.raysPerPath = (int)raysPerPath;
    h->integ.pendingLevels = std::max(1, d.max_depth - 1);
    h->directDirty = false;
    h->haveInteg = true; // configureSampler cleared it while it rebuilt the tables
    return ENIGMA_OK;
}

// CreateLightSampleDistribution + Distribution1D ctor (Enigma/LightDist.cpp:20-71, Enigma/MonteCarlo.h:17-38)
static int buildLightDistribution(EnigmaCtx* h) {
    int n = h->scene.nLights;
    std::vector<float> func(std::max(1, n), 1.0f), cdf((size_t)n + 1, 0.f);
    if (n > 0) {
        bool uniform = h->integDesc.light_strategy == ENIGMA_LIGHTS_UNIFORM || n == 1;
        for (int i = 0; i < n; ++i) func[i] = uniform ? 1.0f : h->lightPower[i];
        cdf[0] = 0.;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
        float funcInt = cdf[n];
        if (funcInt == 0.f) for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
        else for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt;
        h->scene.lightFuncInt = funcInt;
    }
    HIPCHK(h, h->lightCdf.upload(cdf));
    HIPCHK(h, h->lightFunc.upload(func));
    h->scene.lightCdf = h->lightCdf.p;
    h->scene.lightFunc = h->lightFunc.p;
    h->scene.spatialTable = nullptr;
    if (h->integDesc.light_strategy == ENIGMA_LIGHTS_SPATIAL && n > 1) { // one light: uniform (Enigma/LightDist.cpp:22)
        // SpatialLightDistribution's constructor (Enigma/LightDist.cpp:80-103).  As in the reference a voxel's table is made
        // when a path first lands in the voxel (launchSpatialFill); here the rows those tables will take are set aside:
        // one per voxel, or as many as fit the memory allowed for them
        float diag[3], bmax = 0.f;
        for (int c = 0; c < 3; ++c) { diag[c] = h->worldHi[c] - h->worldLo[c]; }
        const int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2); // BBox3::maximumExtent
        bmax = diag[ext];
        size_t total = 1;
        for (int c = 0; c < 3; ++c) {
            h->scene.nVoxels[c] = std::max(1, int(std::round(diag[c] / bmax * 64)));
            h->scene.boundMin[c] = h->worldLo[c]; h->scene.boundMax[c] = h->worldHi[c];
            total *= (size_t)h->scene.nVoxels[c];
        }
        const size_t stride = 2 * (size_t)n + 2;
        size_t budget = (size_t)32 << 30; // a scene whose paths reach more voxels than this holds fails its render with a message
        if (const char* e = getenv("ENIGMA_SPATIAL_TABLE_BYTES")) budget = (size_t)atoll(e);
        // ... and never more than half of what the device has free right now: a smaller GPU gets a smaller table (and the
        // message below if the paths outgrow it), not a bare allocation failure
        size_t freeBytes = 0, totalBytes = 0;
        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) budget = std::min(budget, freeBytes / 2);
        const size_t capacity = std::max<size_t>(1, std::min(total, budget / (stride * sizeof(float))));
        std::vector<float> hal(5 * 128);
        auto radical = [](int base, uint64_t a) { // RadicalInverseSpecialized, Enigma/SamplerInclude.h:49-62
            const float invBase = (float)1 / (float)base;
            uint64_t reversedDigits = 0;
            float invBaseN = 1;
            while (a) { uint64_t next = a / base; uint64_t digit = a - next * base; reversedDigits = reversedDigits * base + digit; invBaseN *= invBase; a = next; }
            return std::min(reversedDigits * invBaseN, 0x1.fffffep-1f);
        };
        for (uint64_t i = 0; i < 128; ++i) {
            uint64_t r = 0;
            for (int b = 0; b < 64; ++b) r |= ((i >> b) & 1ull) << (63 - b); // ReverseBits64
            hal[i] = (float)(r * 5.4210108624275222e-20);                    // Enigma/SamplerInclude.h:2664
            hal[128 + i] = radical(3, i); hal[256 + i] = radical(5, i); hal[384 + i] = radical(7, i); hal[512 + i] = radical(11, i);
        }
        HIPCHK(h, h->halton.upload(hal));
        HIPCHK(h, h->spatialTable.alloc(capacity * stride));
        HIPCHK(h, h->spatialSlot.alloc(total));
        HIPCHK(h, h->spatialNew.alloc(capacity));
        HIPCHK(h, h->spatialCtl.alloc(4));
        HIPCHK(h, hipMemsetAsync(h->spatialSlot.p, 0xff, total * sizeof(uint32_t), h->stream)); // kSpatialEmpty
        HIPCHK(h, hipMemsetAsync(h->spatialCtl.p, 0, 4 * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->scene.spatialSlot = h->spatialSlot.p;
        h->scene.spatialNew = h->spatialNew.p;
        h->scene.spatialCtl = h->spatialCtl.p;
        h->scene.spatialCapacity = (uint32_t)capacity;
        h->scene.spatialTable = h->spatialTable.p;
    }
    h->lightDistDirty = false;
    return ENIGMA_OK;
}

// n paths, each with raysPerPath shadow/MIS ray slots and (DirectLightingIntegrator) pendingLevels stack entries
static int ensureCapacity(EnigmaCtx* h, size_t n, size_t raysPerPath, size_t pendingLevels) {
    if (n * raysPerPath > h->rayCapacity) {
        DevBuf<float4>* r4[] = {&h->shO, &h->shD};
        for (auto* b : r4) HIPCHK(h, b->alloc(n * raysPerPath));
        HIPCHK(h, h->miRec.alloc(n * raysPerPath * 3));
        HIPCHK(h, h->shWz.alloc(n * raysPerPath));
        HIPCHK(h, h->shIdx.alloc(n * raysPerPath));
        HIPCHK(h, h->miIdx.alloc(n * raysPerPath));
        h->rayCapacity = n * raysPerPath;
    }
    if (n * pendingLevels > h->pendCapacity) {
        HIPCHK(h, h->pendO.alloc(n * pendingLevels));
        HIPCHK(h, h->pendD.alloc(n * pendingLevels));
        HIPCHK(h, h->pendB.alloc(n * pendingLevels));
        h->pendCapacity = n * pendingLevels;
    }
    if (h->scene.nSpheres && h->sphereGeom.n < n * 4) HIPCHK(h, h->sphereGeom.alloc(n * 4));
    h->scene.sphereGeom = h->scene.nSpheres ? h->sphereGeom.p : nullptr;
    if (h->haveTrilinear && h->texColors.n < n * 2) HIPCHK(h, h->texColors.alloc(n * 2));
    h->scene.texColors = h->haveTrilinear ? h->texColors.p : nullptr;
    if (n <= h->capacity) return ENIGMA_OK;
    HIPCHK(h, h->pathRec.alloc(n * 4));
    HIPCHK(h, h->L.alloc(n));
    HIPCHK(h, h->arrayOff.alloc(n));
    HIPCHK(h, h->pendCount.alloc(n));
    HIPCHK(h, h->pfilm.alloc(n));
    HIPCHK(h, h->extQ[0].alloc(n));
    HIPCHK(h, h->extQ[1].alloc(n));
    for (auto& q : h->matQ) HIPCHK(h, q.alloc(n));
    h->capacity = n;
    return ENIGMA_OK;
}

// Scenes up to this many bytes of nodes + triangles are traversed out of LDS (see stageSceneToLds).
static constexpr size_t kLdsSceneBytes = 32 * 1024;
static size_t sceneLdsBytes(const EnigmaCtx* h) {
    return ((size_t)h->scene.nNodes * kLdsNodePitch + (size_t)h->scene.nTris * 3) * sizeof(float4);
}

// triangles + materials + lights + light distribution, as k_shade stages them
static size_t shadeLdsBytes(const EnigmaCtx* h) {
    return ((size_t)h->scene.nTris * 3 + (size_t)h->scene.nMaterials * 4 + (size_t)h->scene.nLights * kLightVec4) * sizeof(float4) + (2 * (size_t)h->scene.nLights + 1) * sizeof(float);
}

// Events that bracket the two traversal kernels of a step, on the library's stream.
struct StepEvents { hipEvent_t c0, c1, s0, s1, h0, h1; uint8_t* shadowTimed; uint8_t* shadeTimed; };
constexpr size_t kEventsPerStep = 6; // closest-hit kernel, shadow kernel, the shading kernels of the step: begin / end each
static int acquireStepEvents(EnigmaCtx* h, StepEvents* ev) {
    const size_t i = h->stats.intersect_launches;
    while (h->evPool.size() < (i + 1) * kEventsPerStep) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->evPool.push_back(e);
    }
    if (h->shadowTimed.size() < i + 1) { h->shadowTimed.resize(i + 1); h->shadeTimed.resize(i + 1); }
    h->shadowTimed[i] = 0; h->shadeTimed[i] = 0;
    hipEvent_t* e = &h->evPool[kEventsPerStep * i];
    *ev = StepEvents{e[0], e[1], e[2], e[3], e[4], e[5], &h->shadowTimed[i], &h->shadeTimed[i]};
    return ENIGMA_OK;
}

// Evaluates the trilinear image textures at this bounce's hits (the material queues of the kinds that have a textured
// material), between the closest-hit kernel and the shade kernels.
static void launchTextures(EnigmaCtx* h, const DevPaths& P, int grid, int noneQueue = MK_NONE) {
    if (h->scene.nSpheres) // the surface interactions of this bounce's sphere hits
        for (int k = 0; k < MK_COUNT; ++k)
            if (h->kindPresent[k]) k_sphere_surface<<<grid, kBlock, 0, h->stream>>>(h->scene, P, h->matQ[k].p, k, h->counters.p);
    if (h->scene.nSpheres && h->haveNoMaterial) k_sphere_surface<<<grid, kBlock, 0, h->stream>>>(h->scene, P, h->matQ[noneQueue].p, noneQueue, h->counters.p);
    if (!h->haveTrilinear) return;
    const float diffScale = 1 / std::sqrt((float)h->spp); // RayDifferential::scaleDifferentials, Enigma/Integrator.cpp:276
    for (int k = 0; k < MK_COUNT; ++k)
        if (h->kindTextured[k]) {
            if (h->samplerKind == ENIGMA_SAMPLER_HALTON) k_texture<true><<<grid, kBlock, 0, h->stream>>>(h->scene, P, h->sampler, h->integ, h->camera, diffScale, h->matQ[k].p, k, h->counters.p);
            else k_texture<false><<<grid, kBlock, 0, h->stream>>>(h->scene, P, h->sampler, h->integ, h->camera, diffScale, h->matQ[k].p, k, h->counters.p);
        }
}

// The light-distribution tables of the voxels that this bounce's interactions lie in and that have none yet
// (kernels_tex.h); lastKind: the last queue whose interactions ask for a table.
static void launchSpatialFill(EnigmaCtx* h, const DevPaths& P, int grid, int lastKind) {
    if (!h->scene.spatialTable) return;
    MatQueues mq;
    for (int k = 0; k < kQueueKinds; ++k) mq.q[k] = h->matQ[k].p;
    k_spatial_mark<<<grid, kBlock, 0, h->stream>>>(h->scene, P, mq, 0, lastKind, h->counters.p);
    k_spatial_contrib<<<grid, kBlock, 0, h->stream>>>(h->scene, h->halton.p);
    k_spatial_finish<<<grid, kBlock, 0, h->stream>>>(h->scene);
    k_spatial_filled<<<1, 1, 0, h->stream>>>(h->scene);
}

// The one launch site of the closest-hit kernel: the three integrators' steps and the primary-hit tap all go through it,
// so the tap runs the very instantiation and launch shape that renders.
template <int LDSN, int OVF, bool LDS_SCENE, int MODE, bool SPH>
static void launchClosest(EnigmaCtx* h, const DevPaths& P, uint32_t nPaths, int iter, int gridTrace, const uint32_t* pendCount) {
    DevCounters* ctr = h->counters.p;
    const uint32_t cur = (uint32_t)iter & 1u;
    MatQueues mq;
    for (int k = 0; k < kQueueKinds; ++k) mq.q[k] = h->matQ[k].p;
    const uint32_t* queue = iter == 0 ? nullptr : h->extQ[cur].p;
    const uint32_t* countPtr = iter == 0 ? nullptr : &ctr->extCount[cur].v;
    // the queues a finished ray can land in: the material kinds of the scene; surfaces without a material; under the
    // DirectLightingIntegrator the paths that resume a branch, under the VolPathIntegrator the medium interactions
    uint32_t queueMask = 0;
    for (int k = 0; k < MK_COUNT; ++k) queueMask |= h->kindPresent[k] ? 1u << k : 0u;
    if (MODE == 0) queueMask |= h->haveNoMaterial ? 1u << MK_NONE : 0u;
    else queueMask |= (1u << MK_COUNT) | (1u << kQueueMedium);
    // only the VolPathIntegrator's build draws samples (the medium's free path): it alone exists per sampler
    if (MODE == 2 && h->samplerKind == ENIGMA_SAMPLER_HALTON)
        k_intersect_closest<LDSN, OVF, LDS_SCENE, MODE, SPH, MODE == 2><<<gridTrace, kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, h->stream>>>(
            h->scene, P, h->integ, h->sampler, queue, countPtr, nPaths, mq, ctr, h->refillIdle, pendCount, queueMask);
    else
        k_intersect_closest<LDSN, OVF, LDS_SCENE, MODE, SPH, false><<<gridTrace, kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, h->stream>>>(
            h->scene, P, h->integ, h->sampler, queue, countPtr, nPaths, mq, ctr, h->refillIdle, pendCount, queueMask);
}

// The shade kernels of one bounce, one per material kind present; compiled per sampler (HAL: the HaltonSampler).
template <bool LDS_SCENE, bool VOL, bool SPH, bool HAL>
static void launchShadeKinds(EnigmaCtx* h, const DevPaths& P, const DevRayQueue& shq, const DevRayQueue& misq, uint32_t* nextQ, uint32_t next, int gridShade) {
    DevCounters* ctr = h->counters.p;
    hipStream_t s = h->stream;
    const size_t lds = LDS_SCENE ? shadeLdsBytes(h) : 0;
#define ENIGMA_SHADE_KIND(MK) \
    if (h->kindPresent[MK]) k_shade<MK, LDS_SCENE, VOL, SPH, HAL><<<gridShade, kBlock, lds, s>>>(h->scene, P, h->sampler, h->integ, shq, misq, h->matQ[MK].p, nextQ, next, ctr)
    ENIGMA_SHADE_KIND(MK_MATTE); ENIGMA_SHADE_KIND(MK_METAL); ENIGMA_SHADE_KIND(MK_GLASS_SMOOTH); ENIGMA_SHADE_KIND(MK_GLASS_ROUGH);
    ENIGMA_SHADE_KIND(MK_PLASTIC); ENIGMA_SHADE_KIND(MK_MATTE_ROUGH); ENIGMA_SHADE_KIND(MK_GENERAL);
#undef ENIGMA_SHADE_KIND
    if constexpr (VOL)
        if (h->scene.triMedium) k_shade_medium<LDS_SCENE, SPH, HAL><<<gridShade, kBlock, lds, s>>>(h->scene, P, h->sampler, h->integ, shq, misq, h->matQ[kQueueMedium].p, nextQ, next, ctr);
}
template <bool LDS_SCENE, bool SPH, bool HAL>
static void launchShadeKindsDirect(EnigmaCtx* h, const DevPaths& P, const DevDirect& D, const DevRayQueue& shq, const DevRayQueue& misq, uint32_t* nextQ, uint32_t next, int gridShade) {
    DevCounters* ctr = h->counters.p;
    hipStream_t s = h->stream;
    const size_t lds = LDS_SCENE ? shadeLdsBytes(h) : 0;
#define ENIGMA_SHADE_KIND(MK) \
    if (h->kindPresent[MK]) k_shade_direct<MK, LDS_SCENE, SPH, HAL><<<gridShade, kBlock, lds, s>>>(h->scene, P, h->sampler, h->integ, D, shq, misq, h->matQ[MK].p, nextQ, next, ctr)
    ENIGMA_SHADE_KIND(MK_MATTE); ENIGMA_SHADE_KIND(MK_METAL); ENIGMA_SHADE_KIND(MK_GLASS_SMOOTH); ENIGMA_SHADE_KIND(MK_GLASS_ROUGH);
    ENIGMA_SHADE_KIND(MK_PLASTIC); ENIGMA_SHADE_KIND(MK_MATTE_ROUGH); ENIGMA_SHADE_KIND(MK_GENERAL);
#undef ENIGMA_SHADE_KIND
}

template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
static void launchBounce(EnigmaCtx* h, const DevPaths& P, const DevRayQueue& shq, const DevRayQueue& misq, uint32_t nPaths, int bounce, int gridTrace,
                         int gridShadow, int gridShade, const StepEvents& ev) {
    DevCounters* ctr = h->counters.p;
    uint32_t cur = (uint32_t)bounce & 1u;
    hipStream_t s = h->stream;
    k_stage<<<1, 1, 0, s>>>(ctr, 0, cur, nPaths, bounce == 0 ? 1 : 0);
    (void)hipEventRecord(ev.c0, s);
    launchClosest<LDSN, OVF, LDS_SCENE, 0, SPH>(h, P, nPaths, bounce, gridTrace, nullptr);
    (void)hipEventRecord(ev.c1, s);
    if (bounce >= h->integ.maxDepth && !h->haveNoMaterial) return; // nothing survives the depth test: no shading, no shadow rays
    uint32_t* nextQ = h->extQ[cur ^ 1u].p;
    (void)hipEventRecord(ev.h0, s);
    launchTextures(h, P, gridShade);
    launchSpatialFill(h, P, gridShade, MK_COUNT - 1);
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON) launchShadeKinds<LDS_SCENE, false, SPH, true>(h, P, shq, misq, nextQ, cur ^ 1u, gridShade);
    else launchShadeKinds<LDS_SCENE, false, SPH, false>(h, P, shq, misq, nextQ, cur ^ 1u, gridShade);
    // surfaces without a material: the ray goes on behind them and the vertex does not count as a bounce (Enigma/PathIntegrator.cpp:87-93)
    if (h->haveNoMaterial) k_interface<<<gridShadow, kBlock, 0, s>>>(h->scene, P, h->matQ[MK_NONE].p, MK_NONE, nextQ, cur ^ 1u, ctr);
    (void)hipEventRecord(ev.h1, s);
    *ev.shadeTimed = 1;
    k_stage<<<1, 1, 0, s>>>(ctr, 1, cur, 0, 0);
    if (h->scene.nLights > 0) {
        (void)hipEventRecord(ev.s0, s);
        if constexpr (LDS_SCENE) k_shadow<LDSN, OVF, true, SPH><<<gridTrace, kBlock, sceneLdsBytes(h), s>>>(h->scene, P, shq, ctr, h->refillIdle);
        else k_shadow_stream<LDSN, OVF, SPH><<<std::min(gridTrace, h->residentTraceBlocks), kBlock, 0, s>>>(h->scene, P, shq, ctr, h->refillIdle);
        (void)hipEventRecord(ev.s1, s);
        *ev.shadowTimed = 1;
        if (h->haveAreaLights) k_mis<LDSN, OVF, LDS_SCENE, SPH><<<gridShadow, kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, s>>>(h->scene, P, misq, ctr);
    }
}

// One iteration of VolPathIntegrator::Li's loop for every path in flight (Enigma/VolPathIntegrator.cpp:63-174):
// trace + medium sampling, then the surface, medium and boundary branches, then the shadow and MIS rays.
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
static void launchVolBounce(EnigmaCtx* h, const DevPaths& P, const DevRayQueue& shq, const DevRayQueue& misq, uint32_t nPaths, int iter, int gridTrace,
                            int gridShadow, int gridShade, const StepEvents& ev) {
    DevCounters* ctr = h->counters.p;
    uint32_t cur = (uint32_t)iter & 1u;
    hipStream_t s = h->stream;
    k_stage<<<1, 1, 0, s>>>(ctr, 0, cur, nPaths, iter == 0 ? 1 : 0);
    (void)hipEventRecord(ev.c0, s);
    launchClosest<LDSN, OVF, LDS_SCENE, 2, SPH>(h, P, nPaths, iter, gridTrace, nullptr);
    (void)hipEventRecord(ev.c1, s);
    uint32_t* nextQ = h->extQ[cur ^ 1u].p;
    (void)hipEventRecord(ev.h0, s);
    launchTextures(h, P, gridShade);
    launchSpatialFill(h, P, gridShade, kQueueMedium);
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON) launchShadeKinds<LDS_SCENE, true, SPH, true>(h, P, shq, misq, nextQ, cur ^ 1u, gridShade);
    else launchShadeKinds<LDS_SCENE, true, SPH, false>(h, P, shq, misq, nextQ, cur ^ 1u, gridShade);
    if (h->haveNoMaterial) k_interface<<<gridShadow, kBlock, 0, s>>>(h->scene, P, h->matQ[MK_NONE].p, MK_NONE, nextQ, cur ^ 1u, ctr);
    (void)hipEventRecord(ev.h1, s);
    *ev.shadeTimed = 1;
    if (h->scene.nLights > 0) {
        (void)hipEventRecord(ev.s0, s);
        k_stage<<<1, 1, 0, s>>>(ctr, 3, cur, 0, 0);
        k_shadow_tr<LDSN, OVF, LDS_SCENE, SPH><<<std::min(gridShade, h->residentTraceBlocks), kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, s>>>(h->scene, P, shq, ctr, h->refillIdle);
        (void)hipEventRecord(ev.s1, s);
        *ev.shadowTimed = 1;
        if (h->haveAreaLights) k_mis_tr<LDSN, OVF, LDS_SCENE, SPH><<<gridShadow, kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, s>>>(h->scene, P, misq, ctr);
    }
}

// One step of the depth-first walk of DirectLightingIntegrator::Li's recursion for every path in flight:
// trace the current rays, shade the hits (direct lighting + the specular children), resume pending branches
// of the paths that missed, then the shadow and MIS rays of this step.
template <int LDSN, int OVF, bool LDS_SCENE, bool SPH = false>
static void launchDirectStep(EnigmaCtx* h, const DevPaths& P, const DevDirect& D, const DevRayQueue& shq, const DevRayQueue& misq, uint32_t nPaths,
                             int step, int gridTrace, int gridShadow, int gridShade, const StepEvents& ev) {
    DevCounters* ctr = h->counters.p;
    uint32_t cur = (uint32_t)step & 1u;
    hipStream_t s = h->stream;
    k_stage<<<1, 1, 0, s>>>(ctr, 0, cur, nPaths, step == 0 ? 1 : 0);
    (void)hipEventRecord(ev.c0, s);
    launchClosest<LDSN, OVF, LDS_SCENE, 1, SPH>(h, P, nPaths, step, gridTrace, D.pendCount);
    (void)hipEventRecord(ev.c1, s);
    uint32_t* nextQ = h->extQ[cur ^ 1u].p;
    (void)hipEventRecord(ev.h0, s);
    launchTextures(h, P, gridShade, kQueueMedium);
    if (h->samplerKind == ENIGMA_SAMPLER_HALTON) launchShadeKindsDirect<LDS_SCENE, SPH, true>(h, P, D, shq, misq, nextQ, cur ^ 1u, gridShade);
    else launchShadeKindsDirect<LDS_SCENE, SPH, false>(h, P, D, shq, misq, nextQ, cur ^ 1u, gridShade);
    if (h->glassSpecular) k_resume_pending<<<gridShadow, kBlock, 0, s>>>(P, h->integ, D, h->matQ[MK_COUNT].p, nextQ, cur ^ 1u, ctr);
    // surfaces without a material: Li goes on behind them at the same depth (Enigma/DirectLightingIntegrator.cpp:51-54)
    if (h->haveNoMaterial) k_interface<<<gridShadow, kBlock, 0, s>>>(h->scene, P, h->matQ[kQueueMedium].p, kQueueMedium, nextQ, cur ^ 1u, ctr);
    (void)hipEventRecord(ev.h1, s);
    *ev.shadeTimed = 1;
    k_stage<<<1, 1, 0, s>>>(ctr, 1, cur, 0, 0);
    if (h->scene.nLights > 0) {
        (void)hipEventRecord(ev.s0, s);
        if constexpr (LDS_SCENE) k_shadow<LDSN, OVF, true, SPH><<<gridTrace, kBlock, sceneLdsBytes(h), s>>>(h->scene, P, shq, ctr, h->refillIdle);
        else k_shadow_stream<LDSN, OVF, SPH><<<std::min(gridTrace, h->residentTraceBlocks), kBlock, 0, s>>>(h->scene, P, shq, ctr, h->refillIdle);
        (void)hipEventRecord(ev.s1, s);
        *ev.shadowTimed = 1;
        if (h->haveAreaLights) k_mis<LDSN, OVF, LDS_SCENE, SPH><<<gridShadow, kBlock, LDS_SCENE ? sceneLdsBytes(h) : 0, s>>>(h->scene, P, misq, ctr);
    }
}

// traversal flavour: scene in LDS (small scene, shallow tree) | all stack levels in LDS | 11 levels in LDS + a private spill array;
// scenes with spheres: the last one, with the sphere test compiled in
static bool sceneRunsFromLds(const EnigmaCtx* h) {
    return h->scene.nSpheres == 0 && !(h->integDesc.light_strategy == ENIGMA_LIGHTS_SPATIAL && !h->useDirect && h->scene.nLights > 1) && h->bvhStack <= kLdsSceneLevels &&
           sceneLdsBytes(h) <= kLdsSceneBytes && shadeLdsBytes(h) <= kLdsSceneBytes && !getenv("ENIGMA_NO_LDS_SCENE");
}
// Paths per chunk of the traversal kernels' grids
static int traceGrid(uint32_t nPaths, int cus) { return std::min((int)((nPaths + kTraceChunk - 1) / kTraceChunk), cus * 32); }
static int deviceCus(const EnigmaCtx* h) {
    hipDeviceProp_t prop;
    return hipGetDeviceProperties(&prop, h->device) == hipSuccess ? prop.multiProcessorCount : 256;
}

static int renderImpl(EnigmaCtx* h, int64_t sample_begin, int64_t sample_end) {
    if (!h->haveScene || !h->haveCamera || !h->haveFilm || !h->haveInteg) return h->fail(ENIGMA_E_STATE, "scene, camera, film and integrator must be configured before render");
    if (sample_begin < 0 || sample_end < sample_begin || sample_end > h->spp) return h->fail(ENIGMA_E_INVALID, "sample range [%lld, %lld) outside [0, %lld)", (long long)sample_begin, (long long)sample_end, (long long)h->spp);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->useDirect && h->directDirty) { int rc = finishDirectLighting(h); if (rc) return rc; }
    if (h->lightDistDirty) { int rc = buildLightDistribution(h); if (rc) return rc; }
    auto wall0 = std::chrono::steady_clock::now();
    const uint32_t nPixels = (uint32_t)h->integ.rw * (uint32_t)h->integ.rh;
    h->stats.ms_render = 0; h->stats.ms_intersect = 0; h->stats.intersect_launches = 0;
    h->stats.ms_shadow = 0; h->stats.shadow_launches = 0;
    h->stats.ms_shade = 0; h->stats.shade_launches = 0;
    if (nPixels == 0 || sample_begin == sample_end) return ENIGMA_OK;
    // passes of k sample numbers over all pixels, ~32M paths in flight (capped by the 12-bit sample slot)
    const bool direct = h->useDirect, vol = h->useVolPath && !h->useDirect;
    if (direct && h->haveTrilinear && h->smoothGlass)
        return h->fail(ENIGMA_E_UNSUPPORTED, "trilinear textures with a smooth glass under the DirectLightingIntegrator: its specular rays carry ray differentials on "
                                             "(Enigma/Integrator.cpp:329-444), which is not supported");
    if (direct && h->deferredLobeChoice)
        return h->fail(ENIGMA_E_UNSUPPORTED, "a material with a specular reflection lobe and two specular transmission lobes (an uber with Kr, Kt and opacity < 1, "
                                             "or a mix that adds up to that) under the DirectLightingIntegrator: SpecularTransmit picks its lobe with a sample "
                                             "drawn after the whole reflection subtree (Enigma/Integrator.cpp:329-444), which is not supported");
    const size_t raysPerPath = (size_t)std::max(1, h->integ.raysPerPath), pendingLevels = direct ? (size_t)h->integ.pendingLevels : 0;
    // fewer paths in flight when each carries many ray slots / pending-branch entries (~4 GB of ray slots at most)
    size_t basePaths = (size_t)32 << 20; // ~11 GB of path state at 330 B per path; larger launches run closer to the hardware's rates
    if (const char* e = getenv("ENIGMA_PATHS_IN_FLIGHT")) basePaths = (size_t)atoll(e);
    const size_t targetPaths = std::max<size_t>((size_t)1 << 16, basePaths / std::max<size_t>(1, (raysPerPath + pendingLevels) / 4));
    uint32_t k = (uint32_t)std::max<size_t>(1, std::min<size_t>(targetPaths / nPixels, 64));
    int64_t nSamples = sample_end - sample_begin;
    const int64_t chunkMax = 4096; // sample slots addressable per constant table
    if (const char* e = getenv("ENIGMA_REFILL_IDLE")) h->refillIdle = (uint32_t)atoi(e);
    if (const char* e = getenv("ENIGMA_SHADE_BLOCKS_PER_CU")) h->shadeBlocksPerCu = std::max(1, atoi(e));
    const bool ldsScene = sceneRunsFromLds(h);
    h->stats.scene_in_lds = ldsScene ? 1 : 0;
    const int cus = deviceCus(h);
    h->residentTraceBlocks = cus * 6;
    for (int64_t c0 = 0; c0 < nSamples; c0 += chunkMax) {
        int64_t cn = std::min(chunkMax, nSamples - c0);
        { int rc = uploadSampleSlots(h, nullptr, sample_begin + c0, (size_t)cn); if (rc) return rc; }
        h->sampler.sampleBase = (unsigned long long)(sample_begin + c0);
        size_t passPathsMax = (size_t)nPixels * std::min<int64_t>(k, cn);
        { int rc = ensureCapacity(h, passPathsMax, raysPerPath, pendingLevels); if (rc) return rc; }
        DevPaths P{{h->pathRec.p}, {h->pathRec.p + 1}, h->L.p, {h->pathRec.p + 2}, {h->pathRec.p + 3}, h->pfilm.p};
        const RayField none{nullptr, 1u};
        DevRayQueue shq{{h->shO.p, 1u}, {h->shD.p, 1u}, none, {h->shWz.p, 1u}, h->shIdx.p, (uint32_t)raysPerPath, (!direct && !vol) ? 1u : 0u, none, none, none};
        // a MIS ray is one 48-byte record per slot (origin, direction + light, weight)
        DevRayQueue misq{{h->miRec.p, 3u}, {h->miRec.p + 1, 3u}, {h->miRec.p + 2, 3u}, {nullptr, 1u}, h->miIdx.p, (uint32_t)raysPerPath, 0u, none, none, none};
        if (vol) { // one 96-byte record per shadow-ray slot: a, b, {wz}, p1.p, p1.pError, p1.n
            if (passPathsMax > h->volCapacity) {
                HIPCHK(h, h->shRec.alloc(passPathsMax * 6));
                h->volCapacity = passPathsMax;
            }
            float4* rec = h->shRec.p;
            shq.o = {rec, 6u}; shq.d = {rec + 1, 6u}; shq.wz = {reinterpret_cast<float*>(rec + 2), 24u};
            shq.tgtP = {rec + 3, 6u}; shq.tgtE = {rec + 4, 6u}; shq.tgtN = {rec + 5, 6u};
        }
        DevDirect D{nullptr, nullptr, nullptr, nullptr, nullptr};
        if (direct) D = DevDirect{h->arrayOff.p, h->pendCount.p, h->pendO.p, h->pendD.p, h->pendB.p};
        for (int64_t s0 = 0; s0 < cn; s0 += k) {
            uint32_t ks = (uint32_t)std::min<int64_t>(k, cn - s0);
            uint32_t nPaths = nPixels * ks;
            h->integ.nPixels = nPixels;
            h->integ.slotBase = (uint32_t)s0; // kernels take the integrator by value: set before the pass's launches
            int gridAll = (int)((nPaths + kBlock - 1) / kBlock);
            // traversal grids: enough blocks that the hardware dispatcher balances rays of very different cost,
            // few enough that launches late in a pass (short queues) do not drown in empty blocks
            int gridTrace = traceGrid(nPaths, cus);
            int gridShadow = std::min(gridAll, cus * 4); // MIS rays: few per bounce
            int gridShade = std::min(gridAll, cus * h->shadeBlocksPerCu);
            if (h->samplerKind == ENIGMA_SAMPLER_HALTON) k_generate<true><<<gridAll, kBlock, 0, h->stream>>>(P, h->sampler, h->camera, h->integ, nPaths, nPixels, (uint32_t)s0, h->counters.p,
                                                          direct ? (uint32_t)h->integ.arrayEndDim : 5u, D);
            else k_generate<false><<<gridAll, kBlock, 0, h->stream>>>(P, h->sampler, h->camera, h->integ, nPaths, nPixels, (uint32_t)s0, h->counters.p,
                                                          direct ? (uint32_t)h->integ.arrayEndDim : 5u, D);
            // DirectLightingIntegrator: one step per level of the recursion still alive; the host reads the number of
            // continuing paths after each step (a camera sample is a chain of <= maxDepth steps, or a tree with glass)
            if (!ENIGMA_FULL_BUILD && (direct || vol || h->scene.nSpheres || (!ldsScene && h->bvhStack > 11))) return h->fail(ENIGMA_E_UNSUPPORTED, "not compiled into this experiment build");
#if ENIGMA_FULL_BUILD
            for (int step = 0; direct; ++step) {
                StepEvents ev;
                { int rc = acquireStepEvents(h, &ev); if (rc) return rc; }
                if (h->scene.nSpheres) launchDirectStep<11, 53, false, true>(h, P, D, shq, misq, nPaths, step, gridTrace, gridShadow, gridShade, ev);
                else if (ldsScene) {
                    if (h->bvhStack <= kLdsShallowLevels) launchDirectStep<kLdsShallowLevels, 0, true>(h, P, D, shq, misq, nPaths, step, gridTrace, gridShadow, gridShade, ev);
                    else launchDirectStep<kLdsSceneLevels, 0, true>(h, P, D, shq, misq, nPaths, step, gridTrace, gridShadow, gridShade, ev);
                }
                else if (h->bvhStack <= 11) launchDirectStep<11, 0, false>(h, P, D, shq, misq, nPaths, step, gridTrace, gridShadow, gridShade, ev);
                else launchDirectStep<11, 53, false>(h, P, D, shq, misq, nPaths, step, gridTrace, gridShadow, gridShade, ev);
                h->stats.intersect_launches++;
                uint32_t alive = 0;
                HIPCHK(h, hipMemcpyAsync(&alive, &h->counters.p->extCount[(step & 1) ^ 1].v, sizeof(alive), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if (alive == 0) break;
                if (step > (1 << 20)) return h->fail(ENIGMA_E_DEVICE, "DirectLightingIntegrator walk did not terminate");
            }
            // VolPathIntegrator: crossing a medium boundary is an iteration of Li's loop that is not a bounce, so the number
            // of iterations is not known beforehand; the host reads the number of continuing paths after each one
            for (int iter = 0; vol; ++iter) {
                StepEvents ev;
                { int rc = acquireStepEvents(h, &ev); if (rc) return rc; }
                if (h->scene.nSpheres) launchVolBounce<11, 53, false, true>(h, P, shq, misq, nPaths, iter, gridTrace, gridShadow, gridShade, ev);
                else if (ldsScene) {
                    if (h->bvhStack <= kLdsShallowLevels) launchVolBounce<kLdsShallowLevels, 0, true>(h, P, shq, misq, nPaths, iter, gridTrace, gridShadow, gridShade, ev);
                    else launchVolBounce<kLdsSceneLevels, 0, true>(h, P, shq, misq, nPaths, iter, gridTrace, gridShadow, gridShade, ev);
                }
                else if (h->bvhStack <= 11) launchVolBounce<11, 0, false>(h, P, shq, misq, nPaths, iter, gridTrace, gridShadow, gridShade, ev);
                else launchVolBounce<11, 53, false>(h, P, shq, misq, nPaths, iter, gridTrace, gridShadow, gridShade, ev);
                h->stats.intersect_launches++;
                uint32_t alive = 0;
                HIPCHK(h, hipMemcpyAsync(&alive, &h->counters.p->extCount[(iter & 1) ^ 1].v, sizeof(alive), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if (alive == 0) break;
                if (iter > 4096) return h->fail(ENIGMA_E_DEVICE, "VolPathIntegrator: paths still alive after 4096 iterations");
            }
#endif
            // PathIntegrator: one launch set per bounce.  With surfaces without a material in the scene a pass through one of
            // them is an iteration that is no bounce, so the iterations go on until no path is left (the host reads the count)
            for (int b = 0; !direct && !vol && (h->haveNoMaterial || b <= h->integ.maxDepth); ++b) {
                StepEvents ev;
                { int rc = acquireStepEvents(h, &ev); if (rc) return rc; }
#if ENIGMA_FULL_BUILD
                if (h->scene.nSpheres) launchBounce<11, 53, false, true>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                else if (ldsScene) {
                    if (h->bvhStack <= kLdsShallowLevels) launchBounce<kLdsShallowLevels, 0, true>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                    else launchBounce<kLdsSceneLevels, 0, true>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                }
                else if (h->bvhStack <= 11) launchBounce<11, 0, false>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                else launchBounce<11, 53, false>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
#else
                if (ldsScene) {
                    if (h->bvhStack <= kLdsShallowLevels) launchBounce<kLdsShallowLevels, 0, true>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                    else launchBounce<kLdsSceneLevels, 0, true>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
                }
                else launchBounce<11, 0, false>(h, P, shq, misq, nPaths, b, gridTrace, gridShadow, gridShade, ev);
#endif
                h->stats.intersect_launches++;
                if (h->haveNoMaterial) {
                    uint32_t alive = 0;
                    HIPCHK(h, hipMemcpyAsync(&alive, &h->counters.p->extCount[(b & 1) ^ 1].v, sizeof(alive), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    if (alive == 0) break;
                    if (b > 4096) return h->fail(ENIGMA_E_DEVICE, "PathIntegrator: paths still alive after 4096 iterations");
                }
            }
            if (h->scene.spatialTable) {
                // The pass's samples reach the film only if every one of its light picks read the table of its own voxel: a
                // full table (ctl[2]) or a lookup without a row (ctl[3]) fails the render HERE, before the splat.  The film
                // then holds the passes completed so far -- whole sample numbers, correctly lit -- and nothing of this one.
                uint32_t ctl[4] = {0, 0, 0, 0};
                HIPCHK(h, hipMemcpyAsync(ctl, h->spatialCtl.p, sizeof(ctl), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                if (ctl[2] || ctl[3]) {
                    h->lightDistDirty = true; // the next render starts from an empty table
                    const long long done = (long long)(c0 + s0);
                    if (ctl[2]) return h->fail(ENIGMA_E_UNSUPPORTED, "the paths reached %u voxels of the spatial light distribution, the table holds %u (%d lights; ENIGMA_SPATIAL_TABLE_BYTES "
                                                                   "and free device memory set the limit): sample numbers [%lld, %lld) were deposited, nothing after them -- use the power strategy",
                                               ctl[0], h->scene.spatialCapacity, h->scene.nLights, (long long)sample_begin, (long long)sample_begin + done);
                    return h->fail(ENIGMA_E_DEVICE, "%u light picks found no table for their voxel of the spatial light distribution (marking and lookup disagree): sample numbers "
                                                  "[%lld, %lld) were deposited, nothing after them", ctl[3], (long long)sample_begin, (long long)sample_begin + done);
                }
            }
            k_film_splat<<<(nPixels + kBlock - 1) / kBlock, kBlock, 0, h->stream>>>(P, h->film, h->integ, nPixels, ks);
            HIPCHK(h, hipGetLastError());
        }
        HIPCHK(h, hipStreamSynchronize(h->stream)); // the constant table is re-uploaded per chunk
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (uint64_t i = 0; i < h->stats.intersect_launches; ++i) {
        float ms = 0.f;
        hipEvent_t* e = &h->evPool[i * kEventsPerStep];
        if (hipEventElapsedTime(&ms, e[0], e[1]) == hipSuccess) h->stats.ms_intersect += ms;
        if (h->shadowTimed[i] && hipEventElapsedTime(&ms, e[2], e[3]) == hipSuccess) { h->stats.ms_shadow += ms; h->stats.shadow_launches++; }
        if (h->shadeTimed[i] && hipEventElapsedTime(&ms, e[4], e[5]) == hipSuccess) { h->stats.ms_shade += ms; h->stats.shade_launches++; }
    }
    h->stats.ms_render = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return ENIGMA_OK;
}

extern "C" int enigma_render(EnigmaCtx* h, int64_t sample_begin, int64_t sample_end) {
    if (!h) return ENIGMA_E_INVALID;
    return guarded(h, [&] { return renderImpl(h, sample_begin, sample_end); });
}

extern "C" int enigma_film_read(EnigmaCtx* h, float* rgbw, uint64_t n_floats) {
    if (!h || !rgbw) return ENIGMA_E_INVALID;
    if (!h->haveFilm) return h->fail(ENIGMA_E_STATE, "film not configured");
    if (n_floats != h->filmPixels.n * 4) return h->fail(ENIGMA_E_INVALID, "film has %zu floats, caller asked for %llu", h->filmPixels.n * 4, (unsigned long long)n_floats);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(rgbw, h->filmPixels.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return ENIGMA_OK;
}

extern "C" int enigma_film_device_ptr(EnigmaCtx* h, void** dev_ptr, uint64_t* n_floats) {
    if (!h || !dev_ptr || !n_floats) return ENIGMA_E_INVALID;
    if (!h->haveFilm) return h->fail(ENIGMA_E_STATE, "film not configured");
    *dev_ptr = h->filmPixels.p;
    *n_floats = h->filmPixels.n * 4;
    return ENIGMA_OK;
}


// ---------------------------------------------------------------- multi-GPU: one film reduce over RCCL
// Ranks render disjoint sample numbers of every pixel into their own films (one process per GPU, a full scene replica
// each); the films are float4 sums, so one ncclReduce(sum) over xGMI onto the root rank is the whole exchange.  RCCL is
// bound at run time (dlopen) the first time a communicator call is made: a single-GPU user never loads it, and a
// process that has already loaded an RCCL (PyTorch's) shares that copy.
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*commCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*commUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*errorString)(ncclResult_t) = nullptr;
    std::string error;
};
RcclApi* rccl() {
    static RcclApi api;
    if (api.lib || !api.error.empty()) return &api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    for (const char* n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
        const char* e = dlerror(); // one call: it hands the message out once and then forgets it
        why += std::string(why.empty() ? "" : "; ") + (e ? e : n);
    }
    if (!api.lib) { api.error = "cannot load librccl: " + why; return &api; }
    struct { const char* name; void** slot; } syms[] = {
        {"ncclGetUniqueId", (void**)&api.getUniqueId}, {"ncclCommInitRank", (void**)&api.commInitRank}, {"ncclCommDestroy", (void**)&api.commDestroy},
        {"ncclCommCount", (void**)&api.commCount}, {"ncclCommUserRank", (void**)&api.commUserRank}, {"ncclReduce", (void**)&api.reduce},
        {"ncclGetErrorString", (void**)&api.errorString}};
    for (auto& sym : syms) {
        *sym.slot = dlsym(api.lib, sym.name);
        if (!*sym.slot) { api.error = std::string("librccl lacks ") + sym.name; dlclose(api.lib); api.lib = nullptr; return &api; }
    }
    return &api;
}
} // namespace

#define RCCLCHK(h, api, expr)                                                                              \
    do {                                                                                                   \
        ncclResult_t r__ = (expr);                                                                         \
        if (r__ != ncclSuccess) return (h)->fail(ENIGMA_E_DEVICE, "%s: %s", #expr, (api)->errorString(r__)); \
    } while (0)

extern "C" int enigma_comm_unique_id(void* id_out) {
    if (!id_out) return ENIGMA_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == ENIGMA_COMM_ID_BYTES, "the ABI hands the id over as 128 opaque bytes");
    RcclApi* api = rccl();
    if (!api->lib) return ENIGMA_E_DEVICE;
    ncclUniqueId id;
    if (api->getUniqueId(&id) != ncclSuccess) return ENIGMA_E_DEVICE;
    memcpy(id_out, &id, sizeof(id));
    return ENIGMA_OK;
}

extern "C" int enigma_comm_init(EnigmaCtx* h, int rank, int world, const void* unique_id) {
    if (!h || !unique_id) return ENIGMA_E_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return h->fail(ENIGMA_E_INVALID, "rank %d of %d", rank, world);
    if (h->comm) return h->fail(ENIGMA_E_STATE, "the handle already belongs to a communicator");
    RcclApi* api = rccl();
    if (!api->lib) return h->fail(ENIGMA_E_DEVICE, "%s", api->error.c_str());
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RCCLCHK(h, api, api->commInitRank(&h->comm, world, id, rank));
    // what the communicator itself says, not what the caller passed
    RCCLCHK(h, api, api->commCount(h->comm, &h->commWorld));
    RCCLCHK(h, api, api->commUserRank(h->comm, &h->commRank));
    return ENIGMA_OK;
}

extern "C" int enigma_comm_info(EnigmaCtx* h, int* rank_out, int* world_out) {
    if (!h || !rank_out || !world_out) return ENIGMA_E_INVALID;
    *rank_out = h->commRank;
    *world_out = h->comm ? h->commWorld : 1;
    return ENIGMA_OK;
}

extern "C" int enigma_film_reduce(EnigmaCtx* h, int root) {
    if (!h) return ENIGMA_E_INVALID;
    if (!h->haveFilm) return h->fail(ENIGMA_E_STATE, "film not configured");
    if (!h->comm) { // one rank: the film is already the sum
        if (root != 0) return h->fail(ENIGMA_E_INVALID, "root %d of a single rank", root);
        return ENIGMA_OK;
    }
    if (root < 0 || root >= h->commWorld) return h->fail(ENIGMA_E_INVALID, "root %d of %d ranks", root, h->commWorld);
    RcclApi* api = rccl();
    HIPCHK(h, hipSetDevice(h->device));
    // in place on every rank; only the root's film holds the sum afterwards (the others keep their partial films)
    RCCLCHK(h, api, api->reduce(h->filmPixels.p, h->filmPixels.p, h->filmPixels.n * 4, ncclFloat, ncclSum, root, h->comm, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ENIGMA_OK;
}

extern "C" int enigma_comm_destroy(EnigmaCtx* h) {
    if (!h) return ENIGMA_E_INVALID;
    if (!h->comm) return ENIGMA_OK;
    RcclApi* api = rccl();
    HIPCHK(h, hipSetDevice(h->device));
    (void)hipStreamSynchronize(h->stream);
    ncclResult_t r = api->commDestroy(h->comm);
    h->comm = nullptr; h->commRank = 0; h->commWorld = 1;
    if (r != ncclSuccess) return h->fail(ENIGMA_E_DEVICE, "ncclCommDestroy: %s", api->errorString(r));
    return ENIGMA_OK;
}

extern "C" int enigma_film_resolve(const float* rgbw, uint64_t n_pixels, float scale, float* rgb_out) {
    if (!rgbw || !rgb_out) return ENIGMA_E_INVALID;
    for (uint64_t i = 0; i < n_pixels; ++i) {
        const float* c = &rgbw[4 * i];
        float xyz[3], rgb[3];
        // FilmTile RGB -> Film XYZ on merge (Enigma/Film.cpp:115-118, Enigma/Spectrum.h:28-32)
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
        // Film::WriteImage (Enigma/Film.cpp:174-196, Enigma/Spectrum.h:22-26)
        rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
        rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
        rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
        float w = c[3];
        if (w != 0) {
            float invWt = (float)1 / w;
            rgb[0] = std::max((float)0, rgb[0] * invWt);
            rgb[1] = std::max((float)0, rgb[1] * invWt);
            rgb[2] = std::max((float)0, rgb[2] * invWt);
        }
        rgb_out[3 * i] = rgb[0] * scale; rgb_out[3 * i + 1] = rgb[1] * scale; rgb_out[3 * i + 2] = rgb[2] * scale;
    }
    return ENIGMA_OK;
}

extern "C" int enigma_default_metal_spectra(float eta_rgb[3], float k_rgb[3]) {
    if (!eta_rgb || !k_rgb) return ENIGMA_E_INVALID;
    enigma::default_metal_spectra(eta_rgb, k_rgb);
    return ENIGMA_OK;
}

extern "C" int enigma_get_stats(EnigmaCtx* h, EnigmaStats* out) {
    if (!h || !out) return ENIGMA_E_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    DevCounters c;
    HIPCHK(h, hipMemcpy(&c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost));
    h->stats.camera_samples = c.samples;
    h->stats.rays_closest = c.raysClosest;
    h->stats.rays_shadow = c.raysShadow;
    h->stats.rays_mis = c.raysMis;
    h->stats.shaded_vertices = c.shaded;
    for (int k = 0; k < kQueueKinds; ++k) h->stats.shaded_vertices += c.matCount[k].v; // the last step's queues are summed in when the next step resets them
    {   // traversal tallies: one slot per block of a launch, summed here
        std::vector<unsigned long long> v(h->visits.n);
        HIPCHK(h, hipMemcpy(v.data(), h->visits.p, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        uint64_t unused[2], full[kVisitKinds];
        uint64_t* outs[kVisitKinds * kVisitTallies] = {&h->stats.nodes_closest, &h->stats.tris_closest, &h->stats.node_steps_closest, &h->stats.tri_steps_closest, &full[0],
                                                       &h->stats.nodes_shadow, &h->stats.tris_shadow, &h->stats.node_steps_shadow, &h->stats.tri_steps_shadow, &full[1],
                                                       &h->stats.nodes_mis, &h->stats.tris_mis, &unused[0], &unused[1], &full[2]};
        for (uint32_t k = 0; k < kVisitKinds * kVisitTallies; ++k) {
            uint64_t sum = 0;
            for (uint32_t i = 0; i < kVisitSlots; ++i) sum += v[(size_t)k * kVisitSlots + i];
            *outs[k] = sum;
        }
        h->stats.stack_overflows = full[0] + full[1] + full[2];
    }
    h->stats.bvh_nodes = h->bvhNodes;
    h->stats.bvh_depth = h->bvhDepth;
    h->stats.bvh_node_bytes = sizeof(enigma::Bvh8Node);
    h->stats.ms_bvh_build = h->msBvhBuild;
    *out = h->stats;
    return ENIGMA_OK;
}

// ---------------------------------------------------------------- debug taps
static int debugSlots(EnigmaCtx* h, uint32_t n, const int32_t* pixels, const int64_t* sampleNums, DevBuf<int32_t>& dPix, DevBuf<uint32_t>& dSlots) {
    if (!h->haveInteg) return h->fail(ENIGMA_E_STATE, "integrator not configured");
    const int32_t* sb = h->integDesc.sample_bounds;
    std::vector<uint32_t> slots(n);
    std::vector<int32_t> pix(pixels, pixels + 2 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        if (pixels[2 * i] < sb[0] || pixels[2 * i] >= sb[2] || pixels[2 * i + 1] < sb[1] || pixels[2 * i + 1] >= sb[3])
            return h->fail(ENIGMA_E_INVALID, "pixel %u outside the sample bounds", i);
        if (sampleNums[i] < 0) return h->fail(ENIGMA_E_INVALID, "negative sample number");
        if (h->samplerKind == ENIGMA_SAMPLER_HALTON && ((uint64_t)sampleNums[i] + 1ull) * h->haltonSampler.stride > (1ull << 32))
            return h->fail(ENIGMA_E_UNSUPPORTED, "sample number %lld exceeds the 32-bit Halton index of this build", (long long)sampleNums[i]);
        slots[i] = i;
    }
    { int rc = uploadSampleSlots(h, sampleNums, 0, n); if (rc) return rc; }
    HIPCHK(h, dPix.upload(pix));
    HIPCHK(h, dSlots.upload(slots));
    return ENIGMA_OK;
}

extern "C" int enigma_debug_sobol(EnigmaCtx* h, uint32_t n, const int32_t* pixels_xy, const int64_t* sample_nums, uint32_t n_dims, uint64_t* index_out, float* dims_out) {
    if (!h || !pixels_xy || !sample_nums || !dims_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    if (n == 0) return ENIGMA_OK;
    if (n_dims > h->nDims) return h->fail(ENIGMA_E_INVALID, "n_dims %u exceeds the %u dimensions tabulated for this max_depth", n_dims, h->nDims);
    DevBuf<int32_t> dPix; DevBuf<uint32_t> dSlots; DevBuf<float> dOut;
    int rc = debugSlots(h, n, pixels_xy, sample_nums, dPix, dSlots);
    if (rc == ENIGMA_OK) {
        hipError_t e = dOut.alloc((size_t)n * n_dims);
        if (e == hipSuccess) {
            if (h->samplerKind == ENIGMA_SAMPLER_HALTON) k_debug_sobol<true><<<(n + 255) / 256, 256, 0, h->stream>>>(h->sampler, n, dPix.p, dSlots.p, n_dims, dOut.p);
            else k_debug_sobol<false><<<(n + 255) / 256, 256, 0, h->stream>>>(h->sampler, n, dPix.p, dSlots.p, n_dims, dOut.p);
            e = hipStreamSynchronize(h->stream);
            if (e == hipSuccess) e = hipMemcpy(dims_out, dOut.p, (size_t)n * n_dims * sizeof(float), hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) rc = h->fail(ENIGMA_E_DEVICE, "debug_sobol: %s", hipGetErrorString(e));
    }
    if (rc == ENIGMA_OK && index_out) {
        const int32_t* sb = h->integDesc.sample_bounds;
        for (uint32_t i = 0; i < n; ++i)
            index_out[i] = h->samplerKind == ENIGMA_SAMPLER_HALTON
                         ? enigma::halton_index(h->haltonSampler, (uint32_t)(pixels_xy[2 * i] - sb[0]), (uint32_t)(pixels_xy[2 * i + 1] - sb[1]), (uint64_t)sample_nums[i])
                         : enigma::sobol_interval_to_index(h->log2Res, (uint64_t)sample_nums[i], (uint32_t)(pixels_xy[2 * i] - sb[0]), (uint32_t)(pixels_xy[2 * i + 1] - sb[1]));
    }
    dPix.release(); dSlots.release(); dOut.release();
    return rc;
    });
}

extern "C" int enigma_debug_camera_rays(EnigmaCtx* h, uint32_t n, const int32_t* pixels_xy, const int64_t* sample_nums, float* out_ray, float* out_pfilm) {
    if (!h || !pixels_xy || !sample_nums || !out_ray || !out_pfilm) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->haveCamera) return h->fail(ENIGMA_E_STATE, "camera not configured");
    if (n == 0) return ENIGMA_OK;
    DevBuf<int32_t> dPix; DevBuf<uint32_t> dSlots; DevBuf<float> dRay, dPf;
    int rc = debugSlots(h, n, pixels_xy, sample_nums, dPix, dSlots);
    if (rc == ENIGMA_OK) {
        hipError_t e = dRay.alloc((size_t)n * 6);
        if (e == hipSuccess) e = dPf.alloc((size_t)n * 2);
        if (e == hipSuccess) {
            if (h->samplerKind == ENIGMA_SAMPLER_HALTON) k_debug_camera<true><<<(n + 255) / 256, 256, 0, h->stream>>>(h->sampler, h->camera, n, dPix.p, dSlots.p, dRay.p, dPf.p);
            else k_debug_camera<false><<<(n + 255) / 256, 256, 0, h->stream>>>(h->sampler, h->camera, n, dPix.p, dSlots.p, dRay.p, dPf.p);
            e = hipStreamSynchronize(h->stream);
            if (e == hipSuccess) e = hipMemcpy(out_ray, dRay.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(out_pfilm, dPf.p, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) rc = h->fail(ENIGMA_E_DEVICE, "debug_camera_rays: %s", hipGetErrorString(e));
    }
    dPix.release(); dSlots.release(); dRay.release(); dPf.release();
    return rc;
    });
}

template <bool SPH>
static void launchDebugIntersect(EnigmaCtx* h, int grid, uint32_t n, const float* dRays, int anyHit, int32_t* dTri, uint32_t* dPrim, float* dT, float* dB) {
    // ENIGMA_DEBUG_TINY_STACK keeps only two levels in LDS so that tests drive the private spill array hard
#if !ENIGMA_FULL_BUILD
    if (SPH || h->bvhStack > 12) return;
    k_debug_intersect<12, 0, false><<<grid, kBlock, 0, h->stream>>>(h->scene, n, dRays, anyHit, dTri, dPrim, dT, dB);
    return;
#else
    if (getenv("ENIGMA_DEBUG_TINY_STACK")) k_debug_intersect<2, 62, SPH><<<grid, kBlock, 0, h->stream>>>(h->scene, n, dRays, anyHit, dTri, dPrim, dT, dB);
    else if (h->bvhStack <= 12) k_debug_intersect<12, 0, SPH><<<grid, kBlock, 0, h->stream>>>(h->scene, n, dRays, anyHit, dTri, dPrim, dT, dB);
    else k_debug_intersect<12, 52, SPH><<<grid, kBlock, 0, h->stream>>>(h->scene, n, dRays, anyHit, dTri, dPrim, dT, dB);
#endif
}

// Launches the intersection tap over rays already on the device and brings the results back; the staging buffers
// are the caller's (device memory of the call, or the fixed scratch array of the entry point that allocates nothing).
static hipError_t debugIntersectRun(EnigmaCtx* h, uint32_t n, const float* dRays, int anyHit, int32_t* dTri, uint32_t* dPrim, float* dT, float* dB,
                                    int32_t* triOut, uint32_t* primOut, float* tOut, float* baryOut) {
    int grid = (int)((n + kBlock - 1) / kBlock);
    if (h->scene.nSpheres) launchDebugIntersect<true>(h, grid, n, dRays, anyHit, dTri, dPrim, dT, dB);
    else launchDebugIntersect<false>(h, grid, n, dRays, anyHit, dTri, dPrim, dT, dB);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(triOut, dTri, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && !anyHit) {
        e = hipMemcpyAsync(primOut, dPrim, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(tOut, dT, n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(baryOut, dB, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    }
    hipError_t es = hipStreamSynchronize(h->stream);
    return e != hipSuccess ? e : es;
}

static int debugIntersect(EnigmaCtx* h, uint32_t n, const float* rays7, int anyHit, int32_t* triOut, uint32_t* primOut, float* tOut, float* baryOut) {
    if (!h->haveScene) return h->fail(ENIGMA_E_STATE, "scene not uploaded");
    if (n == 0) return ENIGMA_OK;
    DevBuf<float> dRays, dT, dB; DevBuf<int32_t> dTri; DevBuf<uint32_t> dPrim;
    std::vector<float> r(rays7, rays7 + 7 * (size_t)n);
    hipError_t e = dRays.upload(r);
    if (e == hipSuccess) e = dT.alloc(n);
    if (e == hipSuccess) e = dB.alloc((size_t)n * 3);
    if (e == hipSuccess) e = dTri.alloc(n);
    if (e == hipSuccess) e = dPrim.alloc(n);
    if (e == hipSuccess) e = debugIntersectRun(h, n, dRays.p, anyHit, dTri.p, dPrim.p, dT.p, dB.p, triOut, primOut, tOut, baryOut);
    dRays.release(); dT.release(); dB.release(); dTri.release(); dPrim.release();
    if (e != hipSuccess) return h->fail(ENIGMA_E_DEVICE, "debug_intersect: %s", hipGetErrorString(e));
    return ENIGMA_OK;
}

// ---- primary hits out of the production kernel ----
template <int MODE>
static void launchClosestFlavour(EnigmaCtx* h, const DevPaths& P, uint32_t nPaths, int gridTrace, const uint32_t* pendCount) {
#if ENIGMA_FULL_BUILD
    if (h->scene.nSpheres) launchClosest<11, 53, false, MODE, true>(h, P, nPaths, 0, gridTrace, pendCount);
    else if (sceneRunsFromLds(h)) {
        if (h->bvhStack <= kLdsShallowLevels) launchClosest<kLdsShallowLevels, 0, true, MODE, false>(h, P, nPaths, 0, gridTrace, pendCount);
        else launchClosest<kLdsSceneLevels, 0, true, MODE, false>(h, P, nPaths, 0, gridTrace, pendCount);
    }
    else if (h->bvhStack <= 11) launchClosest<11, 0, false, MODE, false>(h, P, nPaths, 0, gridTrace, pendCount);
    else launchClosest<11, 53, false, MODE, false>(h, P, nPaths, 0, gridTrace, pendCount);
#else
    if (MODE == 0 && sceneRunsFromLds(h)) {
        if (h->bvhStack <= kLdsShallowLevels) launchClosest<kLdsShallowLevels, 0, true, 0, false>(h, P, nPaths, 0, gridTrace, pendCount);
        else launchClosest<kLdsSceneLevels, 0, true, 0, false>(h, P, nPaths, 0, gridTrace, pendCount);
    }
    else if (MODE == 0) launchClosest<11, 0, false, 0, false>(h, P, nPaths, 0, gridTrace, pendCount);
#endif
}

// The first two kernels of a real pass for ONE sample number of every pixel of the render region: k_generate, then the
// closest-hit kernel exactly as enigma_render launches it for bounce 0 (ray pool, voted steps, the scene staged into LDS
// when it is small, the material binning).  The hit records it leaves in the path state are what the shade kernels would
// consume; they are read back instead.
static int primaryHitsImpl(EnigmaCtx* h, int64_t sample_num, int32_t* triOut, uint32_t* primOut, float* tOut, float* baryOut) {
    if (!h->haveScene || !h->haveCamera || !h->haveFilm || !h->haveInteg) return h->fail(ENIGMA_E_STATE, "scene, camera, film and integrator must be configured before the primary-hit tap");
    if (sample_num < 0 || sample_num >= h->spp) return h->fail(ENIGMA_E_INVALID, "sample number %lld outside [0, %lld)", (long long)sample_num, (long long)h->spp);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->useDirect && h->directDirty) { int rc = finishDirectLighting(h); if (rc) return rc; }
    const bool direct = h->useDirect, vol = h->useVolPath && !h->useDirect;
    if (!direct && h->integ.maxDepth < 1) return h->fail(ENIGMA_E_INVALID, "the primary-hit tap needs max_depth >= 1 (at depth 0 the kernel writes no hit record)");
    const uint32_t nPixels = (uint32_t)h->integ.rw * (uint32_t)h->integ.rh;
    if (nPixels == 0) return ENIGMA_OK;
    { int rc = uploadSampleSlots(h, nullptr, sample_num, 1); if (rc) return rc; }
    h->sampler.sampleBase = (unsigned long long)sample_num;
    if (direct && h->deferredLobeChoice)
        return h->fail(ENIGMA_E_UNSUPPORTED, "a material with a specular reflection lobe and two specular transmission lobes (an uber with Kr, Kt and opacity < 1, "
                                             "or a mix that adds up to that) under the DirectLightingIntegrator: SpecularTransmit picks its lobe with a sample "
                                             "drawn after the whole reflection subtree (Enigma/Integrator.cpp:329-444), which is not supported");
    const size_t raysPerPath = (size_t)std::max(1, h->integ.raysPerPath), pendingLevels = direct ? (size_t)h->integ.pendingLevels : 0;
    { int rc = ensureCapacity(h, nPixels, raysPerPath, pendingLevels); if (rc) return rc; }
    DevPaths P{{h->pathRec.p}, {h->pathRec.p + 1}, h->L.p, {h->pathRec.p + 2}, {h->pathRec.p + 3}, h->pfilm.p};
    DevDirect D{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (direct) D = DevDirect{h->arrayOff.p, h->pendCount.p, h->pendO.p, h->pendD.p, h->pendB.p};
    h->integ.nPixels = nPixels;
    h->integ.slotBase = 0;
    DevBuf<float> dT, dB; DevBuf<int32_t> dTri; DevBuf<uint32_t> dPrim;
    hipError_t e = dT.alloc(nPixels);
    if (e == hipSuccess) e = dB.alloc((size_t)nPixels * 3);
    if (e == hipSuccess) e = dTri.alloc(nPixels);
    if (e == hipSuccess) e = dPrim.alloc(nPixels);
    DevBuf<float> hitT;
    if (e == hipSuccess) e = hitT.alloc(nPixels);
    if (e == hipSuccess) {
        const int cus = deviceCus(h);
        const int gridAll = (int)((nPixels + kBlock - 1) / kBlock);
        if (const char* env = getenv("ENIGMA_REFILL_IDLE")) h->refillIdle = (uint32_t)atoi(env);
        h->scene.hitT = hitT.p;
        if (h->samplerKind == ENIGMA_SAMPLER_HALTON) k_generate<true><<<gridAll, kBlock, 0, h->stream>>>(P, h->sampler, h->camera, h->integ, nPixels, nPixels, 0u, h->counters.p, direct ? (uint32_t)h->integ.arrayEndDim : 5u, D);
        else k_generate<false><<<gridAll, kBlock, 0, h->stream>>>(P, h->sampler, h->camera, h->integ, nPixels, nPixels, 0u, h->counters.p, direct ? (uint32_t)h->integ.arrayEndDim : 5u, D);
        k_stage<<<1, 1, 0, h->stream>>>(h->counters.p, 0, 0u, nPixels, 1);
        const int gridTrace = traceGrid(nPixels, cus);
        if (direct) launchClosestFlavour<1>(h, P, nPixels, gridTrace, D.pendCount);
        else if (vol) launchClosestFlavour<2>(h, P, nPixels, gridTrace, nullptr);
        else launchClosestFlavour<0>(h, P, nPixels, gridTrace, nullptr);
        k_debug_gather_primary<<<gridAll, kBlock, 0, h->stream>>>(h->scene, P, h->integ, nPixels, dTri.p, dPrim.p, dT.p, dB.p);
        h->scene.hitT = nullptr;
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMemcpy(triOut, dTri.p, (size_t)nPixels * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(primOut, dPrim.p, (size_t)nPixels * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(tOut, dT.p, (size_t)nPixels * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(baryOut, dB.p, (size_t)nPixels * 3 * sizeof(float), hipMemcpyDeviceToHost);
    }
    h->scene.hitT = nullptr;
    dT.release(); dB.release(); dTri.release(); dPrim.release(); hitT.release();
    if (e != hipSuccess) return h->fail(ENIGMA_E_DEVICE, "debug_primary_hits: %s", hipGetErrorString(e));
    return ENIGMA_OK;
}

extern "C" int enigma_debug_primary_hits(EnigmaCtx* h, int64_t sample_num, uint64_t n_pixels, int32_t* tri_out, uint32_t* prim_id_out, float* t_out, float* bary_out) {
    if (!h || !tri_out || !prim_id_out || !t_out || !bary_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
        const uint64_t want = (uint64_t)std::max(0, h->integ.rw) * (uint64_t)std::max(0, h->integ.rh);
        if (h->haveInteg && n_pixels != want) return h->fail(ENIGMA_E_INVALID, "the render region has %llu pixels, the caller's arrays hold %llu", (unsigned long long)want, (unsigned long long)n_pixels);
        return primaryHitsImpl(h, sample_num, tri_out, prim_id_out, t_out, bary_out);
    });
}

// ---- the entry point that allocates nothing ----
// Staging for enigma_debug_intersect_noalloc: per ray 7 words in (o, d, tMax) and 6 words out (triangle, primitive id,
// t, barycentrics).  One copy of the array lives on every GPU the library runs on, for the life of the process.
constexpr uint32_t kNoAllocWordsPerRay = 13;
static __device__ __align__(256) uint32_t scratch[(size_t)ENIGMA_NOALLOC_MAX_RAYS * kNoAllocWordsPerRay];
// Device address of `scratch` per GPU.  In host code the array's name is only a handle for hipGetSymbolAddress; every
// use of the memory below goes through the address resolved here, with that GPU current.
constexpr int kMaxDevices = 64;
static std::atomic<uint32_t*> g_scratchPtr[kMaxDevices]; // zero-initialised; written once per device (contexts may be created on several threads)

static hipError_t resolveScratch(int device) { // caller has made `device` current
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (g_scratchPtr[device].load(std::memory_order_acquire)) return hipSuccess;
    uint32_t* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratchPtr[device].store(ptr, std::memory_order_release); // two threads racing here store the same address
    return hipSuccess;
}

extern "C" int enigma_debug_intersect_noalloc(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* tri_out, uint32_t* prim_id_out, float* t_out, float* bary_out) {
    if (!h || !rays7 || !tri_out || !prim_id_out || !t_out || !bary_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->haveScene) return h->fail(ENIGMA_E_STATE, "scene not uploaded");
    if (n > ENIGMA_NOALLOC_MAX_RAYS) return h->fail(ENIGMA_E_INVALID, "%u rays exceed the %u the fixed staging array holds", n, (unsigned)ENIGMA_NOALLOC_MAX_RAYS);
    if (n == 0) return ENIGMA_OK;
    uint32_t* base = (h->device >= 0 && h->device < kMaxDevices) ? g_scratchPtr[h->device].load(std::memory_order_acquire) : nullptr;
    if (!base) return h->fail(ENIGMA_E_STATE, "the staging array of device %d was not resolved when the context was created", h->device);
    // carve the array: every region starts on a 256-byte boundary for any n (the ray region is sized for the maximum)
    float* dRays = reinterpret_cast<float*>(base);
    int32_t* dTri = reinterpret_cast<int32_t*>(base + (size_t)ENIGMA_NOALLOC_MAX_RAYS * 7);
    uint32_t* dPrim = base + (size_t)ENIGMA_NOALLOC_MAX_RAYS * 8;
    float* dT = reinterpret_cast<float*>(base + (size_t)ENIGMA_NOALLOC_MAX_RAYS * 9);
    float* dB = reinterpret_cast<float*>(base + (size_t)ENIGMA_NOALLOC_MAX_RAYS * 10);
    // every output word of rays [0, n) is written by the kernel, so nothing of the previous call's contents is read
    hipError_t e = hipMemcpyAsync(dRays, rays7, (size_t)n * 7 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = debugIntersectRun(h, n, dRays, 0, dTri, dPrim, dT, dB, tri_out, prim_id_out, t_out, bary_out);
    if (e != hipSuccess) return h->fail(ENIGMA_E_DEVICE, "debug_intersect_noalloc: %s", hipGetErrorString(e));
    return ENIGMA_OK;
    });
}

extern "C" int enigma_debug_intersect(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* tri_out, uint32_t* prim_id_out, float* t_out, float* bary_out) {
    if (!h || !rays7 || !tri_out || !prim_id_out || !t_out || !bary_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    return debugIntersect(h, n, rays7, 0, tri_out, prim_id_out, t_out, bary_out);
    });
}

extern "C" int enigma_debug_intersect_any(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* hit_out) {
    if (!h || !rays7 || !hit_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    return debugIntersect(h, n, rays7, 1, hit_out, nullptr, nullptr, nullptr);
    });
}

extern "C" int enigma_debug_bsdf_of(EnigmaCtx* h, const EnigmaMaterial* materials, uint32_t n_materials, uint32_t which, uint32_t multiple_lobes, uint32_t n,
                                    const float* wo3, const float* wi3, const float* u2, float* out12) {
    if (!h || !materials || !wo3 || !wi3 || !u2 || !out12 || which >= n_materials) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    if (n == 0) return ENIGMA_OK;
    const EnigmaMaterial& m = materials[which];
    DevMaterial d;
    memset(&d, 0, sizeof(d));
    std::vector<DevLobe> lobeTable;
    if (isGeneralMaterial(m.type)) {
        for (uint32_t k = 0; k <= which; ++k)
            if (materials[k].tex_kd | materials[k].tex_ks | materials[k].tex_kr | materials[k].tex_kt | materials[k].tex_eta | materials[k].tex_k)
                return h->fail(ENIGMA_E_UNSUPPORTED, "the BSDF tap takes constant parameters only");
        std::string why;
        if (!generalMaterial(materials, which, lobeTable, d, why)) return h->fail(ENIGMA_E_UNSUPPORTED, "material %u: %s", which, why.c_str());
    } else {
        if (m.tex_kd | m.tex_ks | m.tex_kr | m.tex_kt | m.tex_eta | m.tex_k) return h->fail(ENIGMA_E_UNSUPPORTED, "the BSDF tap takes constant parameters only");
        if (!convertMaterial(m, d)) return h->fail(ENIGMA_E_UNSUPPORTED, "material type %u is not supported", m.type);
        const bool nb0 = d.c0[0] != 0.f || d.c0[1] != 0.f || d.c0[2] != 0.f, nb1 = d.c1[0] != 0.f || d.c1[1] != 0.f || d.c1[2] != 0.f;
        d.flags = (nb0 ? 1u : 0u) | (nb1 ? 2u : 0u);
    }
    if (lobeTable.empty()) lobeTable.resize(1);
    const bool multi = multiple_lobes != 0;
    DevBuf<float> dWo, dWi, dU, dOut;
    DevBuf<DevLobe> dLobes;
    hipError_t e = dWo.upload(std::vector<float>(wo3, wo3 + 3 * (size_t)n));
    if (e == hipSuccess) e = dWi.upload(std::vector<float>(wi3, wi3 + 3 * (size_t)n));
    if (e == hipSuccess) e = dU.upload(std::vector<float>(u2, u2 + 2 * (size_t)n));
    if (e == hipSuccess) e = dOut.alloc((size_t)n * 12);
    if (e == hipSuccess) e = dLobes.upload(lobeTable);
    if (e == hipSuccess) {
        const unsigned grid = (n + 255) / 256;
#if ENIGMA_FULL_BUILD
        switch (d.kind) {
        case MK_MATTE: k_debug_bsdf<MK_MATTE><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p); break;
        case MK_MATTE_ROUGH: k_debug_bsdf<MK_MATTE_ROUGH><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p); break;
        case MK_METAL: k_debug_bsdf<MK_METAL><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p); break;
        case MK_GLASS_SMOOTH: // without allowMultipleLobes a smooth glass is two specular lobes (Enigma/Material.cpp:204-218)
            if (multi) k_debug_bsdf<MK_GLASS_SMOOTH><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p);
            else k_debug_bsdf<MK_DL_GLASS><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p);
            break;
        case MK_GLASS_ROUGH: k_debug_bsdf<MK_GLASS_ROUGH><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p); break;
        case MK_GENERAL: k_debug_bsdf<MK_GENERAL><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p, dLobes.p, multi); break;
        default: k_debug_bsdf<MK_PLASTIC><<<grid, 256, 0, h->stream>>>(d, n, dWo.p, dWi.p, dU.p, dOut.p); break;
        }
#endif
        e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMemcpy(out12, dOut.p, (size_t)n * 12 * sizeof(float), hipMemcpyDeviceToHost);
    }
    dWo.release(); dWi.release(); dU.release(); dOut.release(); dLobes.release();
    if (e != hipSuccess) return h->fail(ENIGMA_E_DEVICE, "debug_bsdf: %s", hipGetErrorString(e));
    return ENIGMA_OK;
    });
}

extern "C" int enigma_debug_bsdf(EnigmaCtx* h, const EnigmaMaterial* material, uint32_t n, const float* wo3, const float* wi3, const float* u2, float* out12) {
    if (!h || !material) return ENIGMA_E_INVALID;
    if (material->type == ENIGMA_MAT_MIX) return guarded(h, [&]() -> int { return h->fail(ENIGMA_E_INVALID, "a mix material names entries of its array: use enigma_debug_bsdf_of"); });
    return enigma_debug_bsdf_of(h, material, 1, 0, 1, n, wo3, wi3, u2, out12);
}

extern "C" int enigma_debug_film_footprint(EnigmaCtx* h, uint32_t n, const float* pfilm_xy, uint32_t max_per, int32_t* pixels_xy_out, float* weights_out, uint32_t* counts_out) {
    if (!h || !pfilm_xy || !pixels_xy_out || !weights_out || !counts_out) return ENIGMA_E_INVALID;
    return guarded(h, [&]() -> int {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->haveFilm) return h->fail(ENIGMA_E_STATE, "film not configured");
    if (n == 0) return ENIGMA_OK;
    DevBuf<float> dPf, dW; DevBuf<int32_t> dPix; DevBuf<uint32_t> dCnt;
    std::vector<float> pf(pfilm_xy, pfilm_xy + 2 * (size_t)n);
    hipError_t e = dPf.upload(pf);
    if (e == hipSuccess) e = dW.alloc((size_t)n * max_per);
    if (e == hipSuccess) e = dPix.alloc((size_t)n * max_per * 2);
    if (e == hipSuccess) e = dCnt.alloc(n);
    if (e == hipSuccess) {
        k_debug_footprint<<<(n + 255) / 256, 256, 0, h->stream>>>(h->film, n, dPf.p, max_per, dPix.p, dW.p, dCnt.p);
        e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) e = hipMemcpy(pixels_xy_out, dPix.p, (size_t)n * max_per * 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(weights_out, dW.p, (size_t)n * max_per * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(counts_out, dCnt.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    dPf.release(); dW.release(); dPix.release(); dCnt.release();
    if (e != hipSuccess) return h->fail(ENIGMA_E_DEVICE, "debug_film_footprint: %s", hipGetErrorString(e));
    return ENIGMA_OK;
    });
}

