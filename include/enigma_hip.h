/*
 * enigma_hip.h -- C ABI of the MI355X (gfx950) PathIntegrator back end.
 *
 * The reference renderer ("Enigma", namespace RayTrace) has no FFI: its hot path
 * sits behind the C++ virtual `Integrator::Render(const Scene&)`
 * (Enigma/Integrator.h:35-41, Enigma/Integrator.cpp:225-322) and is reached from
 * `renderOptions->m_integrator->Render(*renderOptions->m_scene)`
 * (Enigma/Main.h:245).  This header is the boundary a `GpuPathIntegrator :
 * SamplerIntegrator` binds instead: plain pointers and sizes, POD structs,
 * caller-owned host memory, device memory behind an opaque handle.
 *
 * Conventions
 *  - every call returns 0 on success and a negative ENIGMA_E_* code on failure;
 *    enigma_last_error(h) returns a human readable reason (never NULL).
 *    (The reference only prints and continues, Enigma/Error.h:5-10.)
 *  - one host thread per handle; calls on one handle are serialised by the caller.
 *  - all matrices are 16 floats, row-major in the sense of the reference's
 *    `m_mat[r][c]` as used by Transform::transformPoint (Enigma/Transform.h:36-45).
 *  - geometry is world space (TriangleMesh stores world-space m_p,
 *    Enigma/TriangleMesh.cpp:35-37); no transform composition happens here.
 */
#ifndef ENIGMA_HIP_H
#define ENIGMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENIGMA_OK 0
#define ENIGMA_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define ENIGMA_E_STATE (-2)       /* call order violated (e.g. render before upload) */
#define ENIGMA_E_DEVICE (-3)      /* HIP runtime error */
#define ENIGMA_E_UNSUPPORTED (-4) /* feature of the reference that is out of scope */

/* Material kinds: Enigma/Material.cpp:76-93 (matte), :115-136 (metal),
 * :176-220 (glass), :232-255 (plastic), :318-328 (mirror), :276-309 (translucent),
 * :424-448 (substrate), :536-592 (uber), :146-163 (mix).  Parameters are the values the
 * reference's ConstantTexture::Evaluate would return (Enigma/Texture.h:119-131). */
enum { ENIGMA_MAT_MATTE = 0, ENIGMA_MAT_METAL = 1, ENIGMA_MAT_GLASS = 2, ENIGMA_MAT_PLASTIC = 3,
       ENIGMA_MAT_MIRROR = 4, ENIGMA_MAT_TRANSLUCENT = 5, ENIGMA_MAT_SUBSTRATE = 6, ENIGMA_MAT_UBER = 7,
       ENIGMA_MAT_MIX = 8 };

typedef struct EnigmaMaterial {
    uint32_t type;            /* ENIGMA_MAT_* */
    uint32_t remap_roughness; /* remaproughness flag (metal / glass / plastic) */
    float kd[3];              /* matte, plastic, translucent, substrate, uber: Kd */
    float ks[3];              /* plastic, translucent, substrate, uber: Ks */
    float kr[3];              /* glass, mirror, uber: Kr */
    float kt[3];              /* glass, uber: Kt */
    float eta[3];             /* metal: eta (RGB) */
    float k[3];               /* metal: k (RGB) */
    float sigma;              /* matte: sigma in degrees, clamped to [0, 90]; 0 = LambertianReflection, else OrenNayer
                                 (Enigma/Material.cpp:86-91) */
    float uroughness;         /* metal / glass / substrate: u roughness; plastic / translucent: roughness; uber: "uroughness" if
                                 the scene gives it, else "roughness" (Enigma/Material.cpp:563-566) */
    float vroughness;         /* metal / glass / substrate: v roughness; plastic / translucent: roughness; uber: "vroughness" if
                                 the scene gives it, else the u value (Enigma/Material.cpp:567-570) */
    float index;              /* glass: index of refraction; uber: "eta", else "index" (Enigma/Material.cpp:729-731) */
    /* Image textures bound to the spectrum parameters: index + 1 into EnigmaSceneDesc.textures, 0 = the constant
     * above (a ConstantTexture, Enigma/Texture.h:119-131).  The float parameters stay constants. */
    uint32_t tex_kd, tex_ks, tex_kr, tex_kt, tex_eta, tex_k;
    /* Parameters of the mirror, translucent, substrate, uber and mix materials.  Their Kd / Ks / Kr / Kt take image
     * textures like those of the first four (tex_*); the parameters below are constants.  One such material -- a mix
     * together with the materials it names -- may use two trilinear textures at most (ENIGMA_E_UNSUPPORTED beyond). */
    float reflect[3];         /* translucent: reflect */
    float transmit[3];        /* translucent: transmit */
    float opacity[3];         /* uber: opacity */
    float amount[3];          /* mix: amount (the weight of the first material; the second gets 1 - amount, both clamped) */
    uint32_t mix_a, mix_b;    /* mix: "namedmaterial1" / "namedmaterial2" as indices into the array this material is in;
                                 both must be lower than this material's own index (a named material exists before it is
                                 used, Enigma/Api.cpp:157-173) */
} EnigmaMaterial;

/* ImageTexture<RGBSpectrum, Spectrum> over a UVMapping2D (Enigma/Texture.h:313-415, Enigma/Texture.cpp:22-29,412-453).
 * Two properties of the reference are kept because they decide the picture:
 *  - without "trilinear" the EWA path takes the component count of its differential vectors for their length
 *    (Enigma/MipMap.h:380-381), which puts every lookup on the coarsest MIP level: the texture evaluates to the
 *    image's average colour wherever it is looked up;
 *  - "gamma" reaches TexInfo::m_gamma as a bool turned float and is used as the EXPONENT of the conversion
 *    pow(texel * scale, m_gamma) (Enigma/Texture.h:375-380, Enigma/Texture.cpp:449-451): 1 keeps the texels, 0 makes
 *    every texel 1.
 * With "trilinear", camera rays are filtered with their ray differentials (Enigma/Interaction.cpp:161-209,
 * Enigma/MipMap.h:354-376) and every later ray reads level 0 bilinearly (SpawnRay drops the differentials). */
enum { ENIGMA_WRAP_REPEAT = 0, ENIGMA_WRAP_BLACK = 1, ENIGMA_WRAP_CLAMP = 2 };
typedef struct EnigmaTexture {
    uint32_t width, height; /* other than power-of-two sides are zoomed first, as MipMap's constructor does (MipMap.h:121-127,155-237) */
    const float* rgb;       /* [height*width*3] as ReadImage returns them: top row first, before scale and gamma */
    float scale;            /* TexInfo::m_scale */
    float gamma;            /* TexInfo::m_gamma, the exponent (see above) */
    uint32_t trilinear;     /* TexInfo::m_doTrilinear */
    uint32_t wrap;          /* ENIGMA_WRAP_* */
    float su, sv, du, dv;   /* UVMapping2D: st = (su * u + du, sv * v + dv) */
} EnigmaTexture;

/* Light kinds: PointLight (Enigma/Lights.cpp:89-129), DiffuseAreaLight (:560-651), SpotLight (:133-190),
 * DistantLight (:656-708).  One DiffuseAreaLight exists per emissive triangle (Enigma/Api.cpp:863-874). */
enum { ENIGMA_LIGHT_POINT = 0, ENIGMA_LIGHT_DIFFUSE_AREA = 1, ENIGMA_LIGHT_SPOT = 2, ENIGMA_LIGHT_DISTANT = 3,
       ENIGMA_LIGHT_INFINITE = 4 /* InfiniteAreaLight, Enigma/Lights.cpp:395-546 */,
       ENIGMA_LIGHT_PROJECTION = 6 /* ProjectionLight as the reference has it (Enigma/Lights.cpp:194-296): its constructor
                                      clears the image it has read BEFORE it tests whether there is one (:220-221), so no
                                      projection map ever exists and projection() answers 1 for every direction (:240-241).
                                      What is left is a point light -- Sample_Li = I / d^2 all around, no cone, no image
                                      (:282-289) -- whose Power() is the cone's, I * 2 pi (1 - cos_total_width) (:291-296):
                                      that number decides how often the power and spatial light distributions pick it.
                                      Fields: pos, L = I * scale, cos_total_width = m_cosTotalWidth =
                                      cosf(atanf(tanf(radians(fov) / 2) * sqrtf(2))) (aspect 1 without a map, :225-234). */,
       ENIGMA_LIGHT_GONIOMETRIC = 7 /* GonioPhotometricLight as the reference has it (Enigma/Lights.cpp:301-363).  scale() hands
                                       the illuminant tag (the integer 1) to MipMap::lookup where the filter width belongs
                                       (:327-328); width 1 selects the single texel at the top of the image pyramid
                                       (Enigma/MipMap.h:355-361) whatever the direction.  The light is therefore a point light of
                                       intensity I x that texel, and Power() = 4 pi^2 x lookup((.5, .5), width .5) without the
                                       intensity (:359-363).  Fields: pos, L = I * scale, map_rgb / map_width / map_height = the
                                       texels ReadImage returned for "mapname" (NULL: no map, scale() = 1). */,
       ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE = 5 /* a DiffuseAreaLight whose shape is a Sphere (Sphere::sample / pdf,
                                               Enigma/ShapeImp.cpp:205-310): `triangle` is the index into spheres */ };

typedef struct EnigmaLight {
    uint32_t type;      /* ENIGMA_LIGHT_* */
    uint32_t triangle;  /* area light: index of its triangle (of its sphere for ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE) */
    uint32_t two_sided; /* area light: twosided flag */
    uint32_t n_samples; /* Light::m_nSamples ("samples"/"nsamples", default 1; 0 means 1): read only by
                           DirectLightingIntegrator's UniformSampleAll strategy (Enigma/DirectLightingIntegrator.cpp:76-77) */
    float L[3];         /* area: emitted radiance Lemission; point / spot: intensity I; distant: radiance L */
    float pos[3];       /* point / spot light: world position m_pos */
    float dir[3];       /* distant: m_lightDir = lightToWorld.transformVector(from - to); the reference does NOT
                           normalise it (Enigma/Lights.cpp:656-663) and neither does this library */
    float cos_total_width;    /* spot: cosf(radians(coneangle)) (Enigma/Lights.cpp:138); projection: see ENIGMA_LIGHT_PROJECTION */
    float cos_falloff_start;  /* spot: cosf(radians(coneangle - conedeltaangle)) (Enigma/Lights.cpp:139) */
    float world_to_light[9];  /* spot, infinite: upper-left 3x3 of m_worldToLight, row major (Falloff, Enigma/Lights.cpp:180-190) */
    float light_to_world[9];  /* infinite: upper-left 3x3 of m_lightToWorld */
    /* goniometric: its map (see ENIGMA_LIGHT_GONIOMETRIC).
     * infinite: the latitude-longitude radiance map the constructor reads (ReadImage -> RGB texels, row 0 = theta 0),
     * multiplied by L; NULL = the 1 x 1 map of colour L (no "mapname").  Sides that are not powers of two are zoomed with the
     * reference's Lanczos filter first (MipMap::resizeImage, Enigma/MipMap.h:155-237). */
    uint32_t map_width, map_height;
    const float* map_rgb;     /* [map_height * map_width * 3] */
} EnigmaLight;

#define ENIGMA_TRI_FLIP_NORMAL 1u /* reverseOrientation ^ transformSwapsHandedness */
/* The triangle's mesh has per-vertex normals (TriangleMesh::m_n) / uvs (m_uv): Triangle::intersect then builds the
 * shading geometry from them (Enigma/TriangleMesh.cpp:238-262,290-371).  Meshes without them use the default uvs
 * (0,0), (1,0), (1,1) and shade with the geometric normal. */
#define ENIGMA_TRI_HAS_NORMALS 2u
#define ENIGMA_TRI_HAS_UVS 4u
#define ENIGMA_TRI_REVERSE_ORIENTATION 8u /* m_reverseOrientation alone: flips the shading bitangent (:369) */

/* HomogeneousMedium (Enigma/Medium.cpp:69-108) with its HenyeyGreenstein phase function
 * (Enigma/MediumInterface.cpp:159-200).  Read by the VolPathIntegrator only.
 * kind ENIGMA_MEDIUM_GRID_DENSITY: the reference's GridDensityMedium (MakeMedium "heterogeneous", Enigma/Api.cpp:303-326)
 * exactly as it stands: its Tr and Sample return a default-constructed Spectrum -- black -- without reading the density
 * grid or drawing a sample (Enigma/Medium.cpp:51-59).  A path segment inside such a medium ends the path with nothing
 * added (VolPathIntegrator::Li multiplies the throughput by Sample's value and stops on black,
 * Enigma/VolPathIntegrator.cpp:68-71), and a shadow or MIS ray through it carries nothing (VisibilityTester::Tr,
 * Scene::intersectTr).  The grid itself never reaches the boundary: nothing on this path reads it. */
#define ENIGMA_MEDIUM_HOMOGENEOUS 0u
#define ENIGMA_MEDIUM_GRID_DENSITY 1u
typedef struct EnigmaMedium {
    float sigma_a[3];
    float sigma_s[3];
    float g;
    uint32_t kind;
} EnigmaMedium;

/* tri_material value of a GeometricPrimitive without a material (MakeMaterial "" / "none", Enigma/Api.cpp:135):
 * a medium boundary that the VolPathIntegrator and the PathIntegrator pass through without counting a bounce
 * (Enigma/VolPathIntegrator.cpp:105-111, Enigma/PathIntegrator.cpp:87-93); the DirectLightingIntegrator's recursion
 * through such surfaces is not built: enigma_render returns ENIGMA_E_UNSUPPORTED for it. */
#define ENIGMA_NO_MATERIAL 0xffffffffu

/* A Sphere shape with its GeometricPrimitive (Enigma/ShapeImp.cpp:13-203, Enigma/Primitive.cpp:146-162): full or
 * partial (z range, phi sweep), any object-to-world transform.  Spheres are leaf entries of the BVH beside the
 * triangles; triangles and spheres together number fewer than 2^28.  A sphere may carry a DiffuseAreaLight
 * (`light`); its sampling routines are the reference's: they cover the whole sphere whatever the z range and sweep,
 * and compare the object-space radius with world-space distances (Enigma/ShapeImp.cpp:210-310). */
typedef struct EnigmaSphere {
    float object_to_world[16]; /* Shape::m_objectToWorld->m_mat, row-major m[r][c] */
    float world_to_object[16]; /* Shape::m_worldToObject->m_mat */
    float radius, z_min, z_max, phi_max; /* the constructor's arguments (phi_max in degrees) */
    uint32_t material;         /* index into materials, or ENIGMA_NO_MATERIAL: a medium boundary (VolPathIntegrator) */
    uint32_t flags;            /* ENIGMA_TRI_FLIP_NORMAL: m_reverseOrientation ^ m_transformSwapsHandedness;
                                  ENIGMA_TRI_REVERSE_ORIENTATION: m_reverseOrientation alone (read by Sphere::sample) */
    uint32_t prim_id;          /* Primitive::m_primitiveId; 0 = n_triangles + index + 1 */
    int32_t medium_inside;     /* the primitive's MediumInterface as indices into media + 1, 0 = none (so that a zeroed */
    int32_t medium_outside;    /* struct has no interface); read when the scene has media */
    uint32_t light;            /* index into lights + 1 of its ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE, 0 = not emissive */
} EnigmaSphere;

typedef struct EnigmaSceneDesc {
    uint32_t n_vertices;
    uint32_t n_triangles;
    const float* vertices;        /* [n_vertices*3] world space */
    const uint32_t* indices;      /* [n_triangles*3] */
    const uint32_t* tri_material; /* [n_triangles] index into materials */
    const int32_t* tri_light;     /* [n_triangles] index into lights or -1 */
    const uint32_t* tri_flags;    /* [n_triangles] ENIGMA_TRI_* (NULL = all 0) */
    const uint32_t* tri_prim_id;  /* [n_triangles] reference m_primitiveId
                                     (Enigma/Primitive.cpp:19-26); NULL = i+1 */
    const float* normals;         /* [n_vertices*3] world-space per-vertex normals (as the TriangleMesh constructor leaves
                                     them), read for triangles flagged ENIGMA_TRI_HAS_NORMALS; may be NULL otherwise.
                                     Per-vertex tangents (m_s) are not supported.  The three normals of an emissive
                                     triangle must lie on one side of its plane (Triangle::sample turns the sampled
                                     point's normal to them, Enigma/TriangleMesh.cpp:618-623); otherwise
                                     ENIGMA_E_UNSUPPORTED. */
    const float* uvs;             /* [n_vertices*2] per-vertex uvs for triangles flagged ENIGMA_TRI_HAS_UVS */
    uint32_t n_materials;
    uint32_t n_lights;
    const EnigmaMaterial* materials;
    const EnigmaLight* lights;    /* in the order of Scene::m_lights */
    /* GeometricPrimitive::m_mi per triangle (Enigma/Primitive.cpp:155-160): indices into media or -1 (no medium).
     * Both NULL = no media anywhere.  Camera rays start outside every medium: the reference's
     * PerspectiveCamera never hands its medium to the ray (Enigma/Camera.cpp:144-192). */
    uint32_t n_media;
    const EnigmaMedium* media;
    const int32_t* tri_medium_inside;
    const int32_t* tri_medium_outside;
    uint32_t n_textures;            /* image textures the materials' tex_* fields index */
    const EnigmaTexture* textures;
    uint32_t n_spheres;
    const EnigmaSphere* spheres;
} EnigmaSceneDesc;

/* PerspectiveCamera state read by GenerateRayDifferential (Enigma/Camera.cpp:144-192):
 * ProjectiveCamera::m_rasterToCamera, Camera::m_cameraToWorld (start transform),
 * m_lensRadius, m_focalDistance (Enigma/Camera.h:40-56). */
#define ENIGMA_CAMERA_PERSPECTIVE 0u
/* EnvironmentCamera (Enigma/Camera.cpp:364-383): the ray through film point (x, y) leaves the camera's origin along
 * SphericalDirection(pi * y / yres, 2 pi * x / xres) -- the camera's z axis at the top row -- in camera space; its ray
 * differentials are the rays one pixel further in x and in y (Camera::GenerateRayDifferential, Enigma/Camera.cpp:35-57).
 * Reads camera_to_world only; the resolution is the film's full_resolution. */
#define ENIGMA_CAMERA_ENVIRONMENT 1u
typedef struct EnigmaCameraDesc {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius;
    float focal_distance;
    uint32_t type;            /* ENIGMA_CAMERA_* */
} EnigmaCameraDesc;

/* Film state read by GetSampleBounds/GetFilmTile/AddSample (Enigma/Film.cpp:73-99,
 * 263-300): cropped pixel bounds, filter radius and the 16x16 filter table the
 * Film constructor tabulates (Enigma/Film.cpp:61-70). */
typedef struct EnigmaFilmDesc {
    int32_t full_resolution[2];
    int32_t cropped_pixel_bounds[4]; /* x0, y0, x1, y1 (max exclusive) */
    float filter_radius[2];
    float filter_table[256];
    float max_sample_luminance;      /* +inf by default */
} EnigmaFilmDesc;

enum { ENIGMA_LIGHTS_UNIFORM = 0, ENIGMA_LIGHTS_POWER = 1,
       ENIGMA_LIGHTS_SPATIAL = 2 /* SpatialLightDistribution (Enigma/LightDist.cpp:80-267), the reference's default: one table per voxel
                                    of a grid over the scene bounds; as in the reference a voxel's table is made when a path first
                                    lands in the voxel (between the closest-hit kernel of a bounce and its shading kernels) */ };

/* PathIntegrator + SobolSampler parameters (Enigma/PathIntegrator.cpp:30-44,
 * 182-201; Enigma/Sampler.cpp:211-222). */
typedef struct EnigmaIntegratorDesc {
    int32_t max_depth;
    float rr_threshold;
    int32_t light_strategy;     /* ENIGMA_LIGHTS_*; one light always means uniform
                                   (Enigma/LightDist.cpp:22) */
    int32_t pixel_bounds[4];    /* SamplerIntegrator::m_pixelBounds, x0,y0,x1,y1 */
    int64_t samples_per_pixel;  /* rounded up to a power of two like SobolSampler */
    int32_t sample_bounds[4];   /* SobolSampler::m_sampleBounds = Film::GetSampleBounds() */
} EnigmaIntegratorDesc;

enum { ENIGMA_DL_SAMPLE_ALL = 0, ENIGMA_DL_SAMPLE_ONE = 1 };

/* DirectLightingIntegrator + SobolSampler parameters (Enigma/DirectLightingIntegrator.cpp:30-35,
 * 89-114; Enigma/Sampler.cpp:211-222). */
typedef struct EnigmaDirectLightingDesc {
    int32_t max_depth;          /* "maxdepth", default 5: depth limit of the specular recursion */
    int32_t strategy;           /* ENIGMA_DL_SAMPLE_*: "strategy" all (default) | one */
    int32_t pixel_bounds[4];
    int64_t samples_per_pixel;
    int32_t sample_bounds[4];
} EnigmaDirectLightingDesc;

typedef struct EnigmaStats {
    uint64_t camera_samples;  /* camera samples started */
    uint64_t rays_closest;    /* Scene::intersect equivalents: path rays */
    uint64_t rays_shadow;     /* Scene::intersectP equivalents: shadow rays */
    uint64_t rays_mis;        /* Scene::intersect equivalents: BSDF-sampled MIS rays */
    double ms_render;         /* wall time inside enigma_render, last call */
    double ms_intersect;      /* device time of the closest-hit kernel launches, last call */
    uint64_t intersect_launches;
    uint64_t bvh_nodes;       /* device BVH size */
    uint64_t bvh_depth;
    double ms_bvh_build;
    /* Work the traversal kernels did since the handle was created, per kind of query: tree nodes fetched and
     * triangles tested (Enigma/BVH.cpp:639-722 visits a node / calls Primitive::intersect once per such count). */
    uint64_t nodes_closest, tris_closest; /* path rays */
    uint64_t nodes_shadow, tris_shadow;   /* shadow rays */
    uint64_t nodes_mis, tris_mis;         /* BSDF-sampled MIS rays */
    uint64_t bvh_node_bytes;              /* size of one device tree node */
    uint64_t scene_in_lds;                /* last call: 1 when the traversal kernels ran on a copy of the scene in LDS */
    double ms_shadow;                     /* device time of the shadow-ray kernel launches, last call */
    uint64_t shadow_launches;
    /* Wave-wide steps of the pooled traversal kernels: a step either fetches a node in every lane that is at one or
     * tests triangles in every lane that holds some.  nodes / (64 * node steps) = lanes busy in a node step. */
    uint64_t node_steps_closest, tri_steps_closest, node_steps_shadow, tri_steps_shadow;
    /* Times a traversal found its per-lane stack full and dropped a subtree.  Always 0: the builder caps the tree at
     * the 64 levels the stacks hold; anything else is a defect worth failing a test on. */
    uint64_t stack_overflows;
    /* Device time of the shading kernels of the last call (texture evaluation, light-table fills, one shade kernel per
     * material kind present, medium interactions, boundary crossings) and the number of steps that ran them. */
    double ms_shade;
    uint64_t shade_launches;
    uint64_t shaded_vertices; /* path vertices the shading kernels have worked on since the handle was created */
} EnigmaStats;

typedef struct EnigmaCtx EnigmaCtx;

/* Replaces construction of the integrator object graph (Enigma/Api.cpp:1170-1254). */
int enigma_ctx_create(int device_id, EnigmaCtx** out);
void enigma_ctx_destroy(EnigmaCtx* h);
const char* enigma_last_error(const EnigmaCtx* h);

/* Replaces RenderOptions::MakeScene + BVHAccel construction (Enigma/BVH.cpp:148-189,
 * Enigma/Scene.cpp:19-30): flattens the scene, builds the device BVH, uploads. */
int enigma_scene_upload(EnigmaCtx* h, const EnigmaSceneDesc* scene);
/* Replaces RenderOptions::MakeCamera (Enigma/Camera.cpp:80-118). */
int enigma_camera_set(EnigmaCtx* h, const EnigmaCameraDesc* cam);
/* Replaces Film construction (Enigma/Film.cpp:39-71); clears the device film. */
int enigma_film_config(EnigmaCtx* h, const EnigmaFilmDesc* film);
/* Replaces RenderOptions::MakeSampler's choice by name (Enigma/Api.cpp:425-434) for the two GlobalSamplers:
 * "sobol" (CreateSobolSampler, Enigma/Sampler.cpp:554-559) and "halton" (CreateHaltonSampler, :521-528; the
 * reference's default when a scene names none, RenderOptions::SamplerName, Enigma/Api.h:82).  The choice is taken up by
 * the next enigma_integrator_config / enigma_direct_lighting_config / enigma_volpath_config, whose samples_per_pixel
 * and sample_bounds then construct that sampler; a new context starts with sobol.  For halton, samples_per_pixel is
 * kept as given (not rounded to a power of two), sample_at_pixel_center is "samplepixelcenter", the digit
 * permutations are the ones the reference's process makes (a default-seeded std::mt19937 behind RNG,
 * Enigma/SamplerInclude.h:2639-2655, Enigma/RNG.cpp:18-38), and a configuration whose largest sample index
 * ((samples_per_pixel x largest sample array + 1) x sample stride, the stride at most 128 x 243) does not fit 32 bits
 * returns ENIGMA_E_UNSUPPORTED. */
enum { ENIGMA_SAMPLER_SOBOL = 0, ENIGMA_SAMPLER_HALTON = 1 };
int enigma_sampler_select(EnigmaCtx* h, uint32_t kind, uint32_t sample_at_pixel_center);
/* Replaces CreatePathIntegrator + SobolSampler construction + PathIntegrator::Preprocess
 * (Enigma/PathIntegrator.cpp:41-44,182-201; Enigma/Sampler.cpp:211-222). */
int enigma_integrator_config(EnigmaCtx* h, const EnigmaIntegratorDesc* integ);
/* Replaces CreateDirectLightingIntegrator + SobolSampler construction +
 * DirectLightingIntegrator::Preprocess (Enigma/DirectLightingIntegrator.cpp:71-114): the sample arrays
 * Preprocess requests become reserved Sobol' dimensions.  enigma_render then runs
 * DirectLightingIntegrator::Li (:37-69) with UniformSampleAllLights / UniformSampleOneLight
 * (Enigma/Integrator.cpp:42-92) and SpecularReflect / SpecularTransmit (:329-444).  The integrator
 * configured last is the one enigma_render runs.  ENIGMA_E_UNSUPPORTED where the reference itself is
 * undefined: strategy "all" reading past the sample arrays it requested (a smooth glass with both Kr and
 * Kt branches the recursion, Enigma/Sampler.cpp:62-69), or more than 1024 Sobol' dimensions. */
int enigma_direct_lighting_config(EnigmaCtx* h, const EnigmaDirectLightingDesc* integ);
/* Replaces CreateVolPathIntegrator + SobolSampler construction + VolPathIntegrator::Preprocess
 * (Enigma/VolPathIntegrator.cpp:27-43,179-200).  enigma_render then runs VolPathIntegrator::Li (:45-177):
 * HomogeneousMedium::Sample / Tr, Henyey-Greenstein scattering, EstimateDirect with handleMedia = true
 * (VisibilityTester::Tr, Enigma/Lights.cpp:718-740; Scene::intersectTr, Enigma/Scene.cpp:46-64).
 * Same parameters as the PathIntegrator.  The BSSRDF branch (:138-162) is not built: no material in scope
 * has one. */
int enigma_volpath_config(EnigmaCtx* h, const EnigmaIntegratorDesc* integ);

/* Replaces the tile loop of SamplerIntegrator::Render (Enigma/Integrator.cpp:239-320)
 * for sample numbers [sample_begin, sample_end) of every pixel; accumulates into the
 * device film (FilmTile::AddSample, Enigma/Film.cpp:263-300).  Blocking. */
int enigma_render(EnigmaCtx* h, int64_t sample_begin, int64_t sample_end);

/* Device film as one FilmTile covering the cropped pixel bounds: per pixel
 * {contribSum.r, contribSum.g, contribSum.b, filterWeightSum} (Enigma/Film.h:24-27),
 * row-major.  What Film::MergeFilmTile consumes (Enigma/Film.cpp:101-125). */
int enigma_film_read(EnigmaCtx* h, float* rgbw, uint64_t n_floats);
int enigma_film_clear(EnigmaCtx* h);
/* Device pointer of the film accumulators (float4 per pixel) and their float count,
 * for in-place reductions across ranks (e.g. an RCCL reduce by the caller). */
int enigma_film_device_ptr(EnigmaCtx* h, void** dev_ptr, uint64_t* n_floats);
/* ---- multi-GPU: samples-per-pixel sharding, one film reduce ----
 * The reference renders on one host (tile tasks on a thread pool, Enigma/Integrator.cpp:310-320; films merged under a
 * mutex, Enigma/Film.cpp:101-125) and has no multi-process path.  Here every rank (one process per GPU, a full scene
 * replica each) renders a disjoint range of sample numbers with enigma_render into its own film, and one
 * ncclReduce(sum, float32) over RCCL/xGMI on the library's stream adds the films onto the root's: the device-side
 * counterpart of MergeFilmTile across ranks.  No other exchange exists on the path.
 *   rank 0:  enigma_comm_unique_id(id)  -> the caller ships the 128 bytes to every rank (file, socket, MPI, ...)
 *   all:     enigma_comm_init(h, rank, world, id); enigma_render(h, my_begin, my_end); enigma_film_reduce(h, 0);
 *   rank 0:  enigma_film_read(h, ...)
 * RCCL is loaded when the first of these calls is made; without it they return ENIGMA_E_DEVICE.
 * enigma_comm_init and enigma_film_reduce are collectives WITHOUT a timeout: every rank of the communicator must reach
 * them, and a rank that fails before it does leaves its peers blocked inside RCCL.  A launcher must therefore watch its
 * ranks and end all of them when one dies (enigma_render --gpus N and bench.py --gpus N do). */
#define ENIGMA_COMM_ID_BYTES 128
int enigma_comm_unique_id(void* id_out /* ENIGMA_COMM_ID_BYTES */);
int enigma_comm_init(EnigmaCtx* h, int rank, int world, const void* unique_id);
/* rank and size as the communicator reports them (0 and 1 for a handle outside any communicator) */
int enigma_comm_info(EnigmaCtx* h, int* rank_out, int* world_out);
/* Sums the films of all ranks onto the film of `root`; blocking; collective (every rank calls it). */
int enigma_film_reduce(EnigmaCtx* h, int root);
int enigma_comm_destroy(EnigmaCtx* h);

/* Film::WriteImage normalisation (Enigma/Film.cpp:157-209) on a host rgbw buffer:
 * rgb = max(0, XYZToRGB(RGBToXYZ(sum)) / weight) * scale.  Host code, no device work. */
int enigma_film_resolve(const float* rgbw, uint64_t n_pixels, float scale, float* rgb_out);

/* The "eta" and "k" of a MetalMaterial that names neither (Enigma/Material.cpp:604-608): copper's measured index and
 * extinction spectra turned into RGB by RGBSpectrum::FromSampled (Enigma/Spectrum.cpp:1489-1511) over the CIE 1931
 * colour matching functions.  Host code; no handle, no device work. */
int enigma_default_metal_spectra(float eta_rgb[3], float k_rgb[3]);

int enigma_get_stats(EnigmaCtx* h, EnigmaStats* out);

/* ---- debug taps for bit-exact parity checks (run the device code paths) ---- */

/* SobolSampler::GetIndexForSample + SampleDimension (Enigma/Sampler.cpp:224-240):
 * for n (pixel, sampleNum) pairs return the 64-bit Sobol index and dimensions
 * [0, n_dims) as floats (dims 0,1 are the pixel-relative offsets). */
int enigma_debug_sobol(EnigmaCtx* h, uint32_t n, const int32_t* pixels_xy, const int64_t* sample_nums,
                       uint32_t n_dims, uint64_t* index_out, float* dims_out);
/* GetCameraSample + PerspectiveCamera::GenerateRayDifferential main ray
 * (Enigma/Sampler.cpp:30-37, Enigma/Camera.cpp:144-192): out_ray = o.xyz,d.xyz;
 * out_pfilm = CameraSample::m_image. */
int enigma_debug_camera_rays(EnigmaCtx* h, uint32_t n, const int32_t* pixels_xy, const int64_t* sample_nums,
                             float* out_ray, float* out_pfilm);
/* Scene::intersect / intersectP (Enigma/Scene.cpp:33-44) for n rays given as
 * o.xyz, d.xyz, tMax: closest-hit returns the triangle index (or -1; a sphere
 * reports n_triangles + its index), its reference primitive id (0 on miss), t
 * and barycentrics (for a sphere: the object-space hit point of
 * Sphere::intersect, Enigma/ShapeImp.cpp:38-92); any-hit returns 0/1. */
int enigma_debug_intersect(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* tri_out,
                           uint32_t* prim_id_out, float* t_out, float* bary_out);
int enigma_debug_intersect_any(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* hit_out);
/* Scene::intersect for the camera rays of ONE sample number of every pixel the tile loop visits
 * (Enigma/Integrator.cpp:262-283: GetCameraSample, GenerateRayDifferential, the first scene.intersect of Li,
 * Enigma/PathIntegrator.cpp:62-66; Enigma/BVH.cpp:639-680, Enigma/TriangleMesh.cpp:491-602), out of the kernels that
 * render: the call launches the ray-generation kernel and then the closest-hit kernel of a pass exactly as enigma_render
 * does for bounce 0 (same instantiation for the configured integrator and scene, same grid, ray pool and LDS staging) and
 * reads back the hit records it left in the path state.  Arrays are row-major over the render region (sample bounds
 * clipped to the integrator's pixel bounds, n_pixels = its width x height); values as enigma_debug_intersect reports
 * them.  The rays it traces are counted in EnigmaStats like any others; the film is not touched.  max_depth >= 1. */
int enigma_debug_primary_hits(EnigmaCtx* h, int64_t sample_num, uint64_t n_pixels, int32_t* tri_out,
                              uint32_t* prim_id_out, float* t_out, float* bary_out);
/* enigma_debug_intersect for callers that cannot allocate device memory (harnesses that forbid hipMalloc, loops that
 * must not fragment or grow the device heap).  No counterpart in the reference.  Same arguments and results; the rays
 * and results are staged in a fixed array inside the library (13 words per ray, ENIGMA_NOALLOC_MAX_RAYS rays, 6.5 MiB
 * held on every GPU the library runs on until the process ends) instead of buffers allocated per call: no device
 * allocation or release happens here.  It is synchronous like the other taps -- it copies from and to the caller's
 * pageable host arrays and waits for the library's stream -- and so cannot itself be captured into a HIP graph.  More
 * rays than ENIGMA_NOALLOC_MAX_RAYS: ENIGMA_E_INVALID.  All calls on one GPU share that GPU's array: calls that overlap
 * in time on one GPU (other handles, other threads) race. */
#define ENIGMA_NOALLOC_MAX_RAYS 131072u
int enigma_debug_intersect_noalloc(EnigmaCtx* h, uint32_t n, const float* rays7, int32_t* tri_out,
                                   uint32_t* prim_id_out, float* t_out, float* bary_out);
/* BSDF::f, BSDF::Pdf and BSDF::sample_f with BSDF_ALL (Enigma/BxDF.cpp:885-985) of the BSDF that
 * Material::computeScatteringFunctions builds for `material` (constant parameters; allowMultipleLobes = true as under the
 * PathIntegrator) at a surface whose geometric and shading normal are (0,0,1) and whose dpdu is (1,0,0).  For n triples
 * (wo, wi, u): out12 = f.rgb, Pdf, sampled f.rgb, sampled pdf, sampled wi.xyz, sampled BxDFType as a float. */
int enigma_debug_bsdf(EnigmaCtx* h, const EnigmaMaterial* material, uint32_t n, const float* wo3, const float* wi3,
                      const float* u2, float* out12);
/* The same for material `which` of an array of n_materials (a mix names other entries of its array), with
 * allowMultipleLobes = multiple_lobes != 0: the PathIntegrator and VolPathIntegrator pass true, the
 * DirectLightingIntegrator false, which turns smooth glass into two specular lobes (Enigma/Material.cpp:191-218). */
int enigma_debug_bsdf_of(EnigmaCtx* h, const EnigmaMaterial* materials, uint32_t n_materials, uint32_t which,
                         uint32_t multiple_lobes, uint32_t n, const float* wo3, const float* wi3, const float* u2,
                         float* out12);
/* FilmTile::AddSample footprint (Enigma/Film.cpp:269-292) for n film positions:
 * up to max_per (pixel x, pixel y) pairs and filter weights each, count in counts. */
int enigma_debug_film_footprint(EnigmaCtx* h, uint32_t n, const float* pfilm_xy, uint32_t max_per,
                                int32_t* pixels_xy_out, float* weights_out, uint32_t* counts_out);

#ifdef __cplusplus
}
#endif
#endif /* ENIGMA_HIP_H */
