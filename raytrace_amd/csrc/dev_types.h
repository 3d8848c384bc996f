// Device data layout of the wavefront path tracer (all arrays live in HBM for a whole render).
#pragma once
#include "dev_common.h"

namespace edev {

// Material kinds the shade kernels are sorted by.  ENIGMA_MAT_GLASS splits by roughness at
// upload: a smooth glass is one FresnelSpecular lobe (allowMultipleLobes is always true on the
// PathIntegrator path, Enigma/PathIntegrator.cpp:89, Enigma/Material.cpp:190-194).
enum MatKind : uint32_t { MK_MATTE = 0, MK_METAL = 1, MK_GLASS_SMOOTH = 2, MK_GLASS_ROUGH = 3, MK_PLASTIC = 4,
                          MK_MATTE_ROUGH = 5 /* matte with sigma != 0: Oren-Nayar instead of Lambert */,
                          MK_GENERAL = 6 /* mirror, translucent, substrate, uber, mix: a list of lobes (DevLobe) walked at run time */,
                          MK_COUNT = 7,
                          MK_NONE = 7 /* a triangle without a material: a medium boundary (VolPathIntegrator) */,
                          MK_DL_GLASS = 8 /* lobe set of smooth glass under DirectLightingIntegrator; never stored in a scene */ };
// One queue per material kind, then queue MK_COUNT: DirectLightingIntegrator paths that resume a pending branch /
// VolPathIntegrator paths at a material-less surface; queue MK_COUNT + 1: VolPathIntegrator medium interactions.
constexpr int kQueueKinds = MK_COUNT + 2;
constexpr int kQueueMedium = MK_COUNT + 1;

// HomogeneousMedium (Enigma/Medium.cpp:69-108): sigma_t = sigma_s + sigma_a, Henyey-Greenstein g
struct DevMedium { float sigmaT[3]; float g; float sigmaS[3]; float opaque; }; // opaque != 0: the reference's GridDensityMedium, whose Tr and Sample return black (Enigma/Medium.cpp:51-59)

struct DevMaterial {        // 64 bytes
    uint32_t kind;          // MatKind
    uint32_t flags;         // bit0: first colour non-black, bit1: second colour non-black
    float alphax, alphay;   // TrowbridgeReitz alphas after remap + clamp (Enigma/MicrofacetDist.cpp:256-263,323-328)
    float c0[3]; float eta; // matte/plastic: Kd.  metal: eta rgb.  glass: Kr; eta = index
    float c1[3]; float pad; // plastic: Ks.  metal: k rgb.  glass: Kt
    uint32_t tex0, tex1;    // device texture index + 1 bound to c0 / c1 (trilinear image textures), 0 = the constant
    uint32_t lobeOff;       // MK_GENERAL: first entry in DevScene::lobes of the list built with allowMultipleLobes = true, ...
    uint32_t lobeInfo;      // ... its length | length of the allowMultipleLobes = false list << 8 (it follows the first); eta = BSDF::m_eta
};

// One BxDF of an MK_GENERAL material, in the order Material::computeScatteringFunctions adds them (Enigma/Material.cpp).
// kind is a LobeKind (dev_bsdf.h).  scaled: the lobe sits inside a ScaledBxDF (MixMaterial, Enigma/Material.cpp:158-162):
// its values are multiplied by scale, and because ScaledBxDF does not override pdf() (Enigma/BxDF.h:157-170) the lobe
// answers density queries with the base class's cosine-hemisphere density (Enigma/BxDF.cpp:568-571) whatever it wraps.
struct DevLobe {            // 64 bytes
    uint32_t kind, scaled;
    float ax, ay;           // the lobe's TrowbridgeReitz alphas (Oren-Nayar: A and B)
    float col[3]; float etaI;  // R / T / Kd; conductor: eta.  Dielectric Fresnel / transmission interface: etaI, etaT
    float col2[3]; float etaT; // conductor: k; FresnelSpecular: T; FresnelBlend: Rs
    float scale[3];
    uint32_t tex;           // 0: col / col2 are the colours.  Else a trilinear texture feeds a colour: bits 0-7 / 8-15 the slot + 1
                            // (DevMaterial::tex0 / tex1, DevScene::texColors) behind col / col2, which then hold the factor the
                            // texture's value is multiplied by; see kLobe* below
};
// DevLobe::tex, upper bits.  Which lobes exist at a hit is decided by the IsBlack tests of computeScatteringFunctions on
// the values the textures have THERE: kLobeOwn -- the lobe exists if its colour is not black (tested on the texture's
// value alone with kLobeRaw: TranslucentMaterial tests Kd and multiplies by reflect afterwards, Enigma/Material.cpp:288-293;
// on the product otherwise: UberMaterial tests op * Kd, :553-554); kLobeEither -- if either of its two colours is not black
// (FresnelBlend, FresnelSpecular: :439, :188).  kLobeNoClamp: the metal's eta and k are not clamped (:122-131).
constexpr uint32_t kLobeRaw = 1u << 16, kLobeOwn = 1u << 17, kLobeEither = 1u << 18, kLobeNoClamp = 1u << 19;
static_assert(sizeof(DevLobe) == 64, "four float4 per lobe");
constexpr int kMaxLobes = 8; // MAX_BxDFS, Enigma/BxDF.h:8

static_assert(sizeof(DevMaterial) == 64, "k_shade copies materials as four float4");

struct DevLight {           // 144 bytes
    uint32_t type;          // 0 point, 1 diffuse area, 2 spot, 3 distant, 4 infinite area, 5 diffuse area on a sphere
    uint32_t tri;           // area: triangle position in the device (BVH) order; type 5: index into DevScene::spheres
    uint32_t twoSided;
    float area;             // Triangle::area() (Enigma/TriangleMesh.cpp:438-446)
    float L[3]; uint32_t nSamples;  // Light::m_nSamples (DirectLightingIntegrator, strategy all)
    float pos[3];           // point / spot: m_pos.  distant: m_lightDir (not normalised, as in the reference)
    float cosTotalWidth;    // spot
    float w2l0[3]; float cosFalloffStart; // spot: rows of the upper 3x3 of m_worldToLight
    float w2l1[3]; float worldRadius;     // distant: radius of the scene's bounding sphere (Light::Preprocess)
    float w2l2[3]; float sampleFlip;      // area: 1 when a sampled point's normal is -Cross(p1 - p0, p2 - p0) (Enigma/TriangleMesh.cpp:615-625)
    float l2w0[3]; float pad2;             // infinite: rows of the upper 3x3 of m_lightToWorld (its m_worldToLight sits in w2l0..2,
    float l2w1[3]; float pad3;             // its radiance map and sampling distribution in DevScene::envMaps[tri])
    float l2w2[3]; float pad4;
};
constexpr uint32_t kLightVec4 = 9;

static_assert(sizeof(DevLight) == kLightVec4 * 16, "k_shade copies lights as float4");

// InfiniteAreaLight data (Enigma/Lights.cpp:395-442): level 0 of the radiance MipMap (texels already times L) and the
// Distribution2D over the 2w x 2h luminance image (Enigma/MonteCarlo.cpp:236-271), flattened row by row.
struct DevEnvMap {
    const float4* texels;   // [h][w]
    const float* condFunc;  // [nv][nu]
    const float* condCdf;   // [nv][nu + 1]
    const float* condInt;   // [nv] funcInt of each row
    const float* margFunc;  // [nv]
    const float* margCdf;   // [nv + 1]
    float margInt;
    int w, h, nu, nv;
};

// One triangle, 48 bytes, three float4 loads:
//   v0 = {p0.xyz, bits(original triangle index)}
//   v1 = {p1.xyz, bits(material index [0:24) | material kind [24:27) | flip << 31)}
//   v2 = {p2.xyz, bits(light index or -1)}
struct DevTexture {         // a trilinear ImageTexture: the MipMap pyramid (level 0 first) and its UVMapping2D
    const float4* texels;
    uint32_t w, h, nLevels, wrap; // wrap: 0 repeat, 1 black, 2 clamp
    float su, sv, du, dv;
    uint32_t levelOff[16];  // first texel of each level
};

struct DevSphere {          // 128 bytes: a Sphere shape and its primitive
    float o2w[12], w2o[12]; // rows 0..2 of Shape::m_objectToWorld / m_worldToObject (affine)
    float radius, zMin, zMax, thetaMin, thetaMax, phiMax; // Sphere's members (Enigma/ShapeImp.cpp:13-27)
    uint32_t mw;            // the second triangle word's layout: material | kind << 24 | reverseOrientation << 29 | flip-normal << 31
    uint32_t light;         // index + 1 of its DiffuseAreaLight (DevLight type 5), 0 = not emissive
};

constexpr uint32_t kSpatialEmpty = 0xffffffffu, kSpatialClaimed = 0xfffffffeu;

struct DevScene {
    const float4* nodes;        // 8-wide BVH nodes, 5 float4 each
    const float4* tris;         // triangles in BVH order, 3 float4 each (a sphere owns a slot too, see dev_trace8.h)
    const DevMaterial* materials;
    const DevLight* lights;
    // SpatialLightDistribution (Enigma/LightDist.cpp:80-267): per filled voxel nLights func values, nLights + 1 cdf values
    // and funcInt; nullptr = one table for the whole scene (lightCdf / lightFunc below).  As in the reference a voxel's
    // table is made when a path first lands in the voxel: spatialSlot[voxel] = its row of the table, or kSpatialEmpty.
    float* spatialTable;
    uint32_t* spatialSlot;
    uint32_t* spatialCtl;       // [0] rows in use, [1] rows claimed since the last fill, [2] voxels that found the table full,
                                // [3] lookups that found no row for their voxel (spatialLookup)
    uint32_t* spatialNew;       // the voxels claimed since the last fill
    uint32_t spatialCapacity;   // rows of the table
    int nVoxels[3];
    float boundMin[3], boundMax[3]; // Scene::worldBound()
    const float* lightCdf;      // Distribution1D::cdf, nLights + 1
    const float* lightFunc;     // Distribution1D::func, nLights
    const uint32_t* primId;     // reference m_primitiveId by original triangle index
    const DevEnvMap* envMaps;   // one per infinite area light (DevLight::tri indexes it)
    const uint32_t* infiniteLights; // Scene::m_infiniteLights: indices of the infinite area lights
    int nInfinite;
    const DevMedium* media;     // VolPathIntegrator only
    const float4* triShade;     // 4 float4 per triangle (device order): (n0, u0) (n1, u1) (n2, u2) (v0, v1, v2, -), read for triangles whose
                                // second word flags per-vertex normals or uvs; nullptr = no mesh has either
    const DevSphere* spheres;   // primitive index nTris + i in hit records
    uint32_t nSpheres;
    float4* sphereGeom;         // four float4 per path: the surface interaction k_sphere_surface worked out for a sphere hit
                                // (p | pError.x) (n | pError.y) (dpdu | pError.z) (wo | material index)
    const DevTexture* textures; // trilinear image textures; nullptr = none
    float4* texColors;          // two float4 per path: the textured colours k_texture evaluated at the path's hit
    unsigned long long* visits; // tallies of nodes fetched / triangles tested per kind of ray query (dev_trace.h), nullptr = not kept
    float* hitT;                // debug tap (enigma_debug_primary_hits): when set, the closest-hit kernel leaves the distance of every
                                // ray that finishes its walk at the path's index (+inf for a miss); nullptr in every render
    const uint32_t* triMedium;  // per triangle (device order): (inside medium + 1) | (outside medium + 1) << 16, 0 = none; nullptr = no media
    float lightFuncInt;
    int nLights;
    uint32_t nTris;
    uint32_t nNodes;
    uint32_t nMaterials;
    const DevLobe* lobes;       // lobe lists of the MK_GENERAL materials (last: the kernels that never read it keep the argument
                                // offsets, and with them the scalar registers, they had without it)
};

struct DevCamera {
    float r2c[16];
    float c2w[16];
    float lensRadius, focalDistance;
    uint32_t type;              // ENIGMA_CAMERA_*
    float resX, resY;           // Film::m_fullResolution (the environment camera's angles are fractions of it)
};

struct DevFilm {
    int x0, y0, x1, y1;         // cropped pixel bounds
    float radiusX, radiusY, invRadiusX, invRadiusY;
    float maxSampleLuminance;
    float4* pixels;             // {sum r, g, b, filter weight sum} per pixel
    const float* filterTable;   // 16 x 16
};

// SobolSampler; with res == 0 the HaltonSampler, which keeps other data behind the same pointers: tx / ty = the two
// terms of the pixel's index offset per column / row, sampleConst = the sample number of each slot, bitConst = its
// parameters, primes and digit permutations (haltonDimension in kernels.h).
struct DevSampler {
    const uint32_t* tx;         // [nDims][res] pixel-x XOR table
    const uint32_t* ty;         // [nDims][res]
    const uint32_t* sampleConst;// [samples in flight][nDims]
    const uint32_t* bitConst;   // [nDims][kSampleBits]: constant of sample number 1 << b (the constant is linear in the bits)
    unsigned long long sampleBase; // sample number of slot 0
    uint32_t res, nDims;
    int sbx0, sby0;             // sample bounds origin
    float resolutionF;          // (float)m_resolution
};

constexpr int kSampleBits = 40;

struct DevIntegrator {
    int maxDepth;
    float rrThreshold;
    int rx0, ry0, rw, rh;       // render region = sample bounds clipped to pixel bounds
    uint32_t nPixels, slotBase; // rw * rh; sample slot of the pass's first sample number
    // DirectLightingIntegrator
    int dlStrategy;             // ENIGMA_DL_SAMPLE_*
    int arrayEndDim;            // GlobalSampler::m_arrayEndDim: first dimension after the sample arrays
    int raysPerPath;            // shadow (and MIS) ray slots per path: sum of the lights' sample counts, or 1
    int pendingLevels;          // pending-branch stack entries per path (maxDepth - 1, at least 1)
};

// DirectLightingIntegrator per-path state: the recursion Li -> SpecularReflect -> Li ... -> SpecularTransmit
// (Enigma/Integrator.cpp:329-444) is walked depth first; a node's transmission branch waits on this stack
// while its reflection subtree runs, because the sampler dimensions it consumes follow that subtree's.
struct DevDirect {
    uint32_t* arrayOff;   // Sampler::m_array2DOffset
    uint32_t* pendCount;
    float4* pendO;        // [path][level] o.xyz, valid (1) / placeholder that only consumes its Get2D (0)
    float4* pendD;        // d.xyz, depth of the child
    float4* pendB;        // throughput of the child
};

// Per-path state, structure of arrays of float4 so that a kernel touches one 16-byte slot per field group:
// the sampler cursor travels in the spare word of the ray direction, the hit triangle in the spare word of the
// hit record, and the pixel / sample slot of a path follow from its index (path = slot * nPixels + pixel).
// rayO, rayD, beta and hit of a path are the four float4 of one 64-byte record: a kernel that visits paths in queue
// order touches one 64-byte piece of memory per path instead of one in each of four arrays.  L and pfilm have other
// readers (shadow / MIS kernels, film splat) and stay arrays of their own.
struct PathField { // field k of the records: element i is at base[4 * i]
    float4* base;
    DEV float4& operator[](size_t i) const { return base[i * 4]; }
};
struct DevPaths {
    PathField rayO;   // o.xyz, unused
    PathField rayD;   // d.xyz, bits(state): medium + 1 [0:12) (VolPathIntegrator) | dim[12:22) | bounces or depth[22:30) | specularBounce[30]
    float4* L;        // radiance accumulated so far, unused
    PathField beta;   // throughput rgb, etaScale
    PathField hit;    // bits(triangle, device order), b0, b1, b2
    float2* pfilm;    // CameraSample::m_image
};

// Shadow rays and BSDF-sampled MIS rays.  The ray itself is stored at the slot of its path (written the
// moment the shade kernel has it, so nothing stays live in registers); idx is the compact list of
// paths that have one this bounce.
// A shadow ray (origin, direction to just short of the light sample, contribution) is nine floats; they are
// packed into two float4 and one float instead of three float4: a = {o.xyz, d.x}, b = {d.y, d.z, w.x, w.y},
// wz = w.z.  Its tMax is the constant 1 - ShadowEpsilon (Enigma/Interaction.cpp:58-64).
// A field of the ray slots: element i at base[i * stride].  Stride 1 = an array of its own; the VolPathIntegrator's
// shadow rays keep all their fields in one 96-byte record per slot (stride 6 float4), two 64-byte pieces of memory
// instead of six.
struct RayField { float4* base; uint32_t stride; DEV float4& operator[](size_t i) const { return base[i * stride]; } };
struct RayFieldF { float* base; uint32_t stride; DEV float& operator[](size_t i) const { return base[i * stride]; } };
struct DevRayQueue {
    RayField o;       // MIS: [slot] o.xyz, bits(medium + 1) under the VolPathIntegrator.  shadow: a
    RayField d;       // MIS: [slot] d.xyz, bits(light index).      shadow: b
    RayField w;       // MIS: [slot] weight of the light's Le.      shadow: unused
    RayFieldF wz;     // shadow only
    uint32_t* idx;    // [0, count) ray slots (= path index * perPath + ray of the path)
    uint32_t perPath; // 1 for PathIntegrator
    uint32_t dense;   // shadow rays: 1 = the ray of queue position i is stored at i (PathIntegrator: the shadow kernel then
                      // streams o/d/wz instead of gathering 16-byte pieces at path slots), 0 = at the slot idx[i] names
    // VolPathIntegrator shadow rays: the light-side end point of VisibilityTester::Tr, which respawns the ray
    // towards it after every medium boundary (Enigma/Lights.cpp:718-740)
    RayField tgtP;    // p1.p, bits(medium + 1 of the first segment)
    RayField tgtE;    // p1.pError
    RayField tgtN;    // p1.n
};

// Queue tails are bumped with device-scope atomics by every block of a launch.  Atomics on one
// 128-byte line are served one after the other by a single L2 channel, so each tail gets a line of
// its own: tails that share a line queue up behind each other even though they are different words.
struct alignas(128) DevTail { uint32_t v; uint32_t pad[31]; };
struct DevCounters {
    DevTail extCount[2];        // extension-ray queue sizes (current, next)
    DevTail matCount[kQueueKinds];
    DevTail shadowCount, misCount;
    DevTail shadowCursor;       // next unreserved position of the shadow queue while k_shadow_stream runs (its waves reserve runs of it)
    unsigned long long raysClosest, raysShadow, raysMis, samples;
    unsigned long long shaded;  // path vertices handed to the shading kernels (the material queues' sizes, summed when they are reset)
};

DEV uint32_t packState(uint32_t medium, uint32_t dim, uint32_t bounces, uint32_t spec) {
    return medium | ((dim & 0x3ffu) << 12) | (bounces << 22) | (spec << 30); // the reference aborts past dimension 1023; here the cursor wraps
}
DEV uint32_t stMedium(uint32_t s) { return s & 0xfffu; } // medium + 1, 0 = none
DEV uint32_t stDim(uint32_t s) { return (s >> 12) & 0x3ffu; }
DEV uint32_t stBounces(uint32_t s) { return (s >> 22) & 0xffu; }
DEV uint32_t stSpec(uint32_t s) { return (s >> 30) & 1u; }
// bit 31: the ray is no longer the camera's own ray although no bounce has been counted (a VolPath ray continued
// through a material-less surface): it carries no ray differentials (Enigma/VolPathIntegrator.cpp:108, Enigma/Ray.cpp:44-47)
constexpr uint32_t kStNoDifferentials = 1u << 31;
DEV bool stNoDifferentials(uint32_t s) { return (s >> 31) != 0; }

} // namespace edev
