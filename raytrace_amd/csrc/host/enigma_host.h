// Host-side mirror of the reference's interface for the PathIntegrator path (C++17, header only).
//
// Same class and method names, argument meaning and error behaviour as the reference's public
// interface for this path, so host code written against Enigma reads the same:
//   Integrator / SamplerIntegrator        Enigma/Integrator.h:35-70
//   PathIntegrator ctor arguments         Enigma/PathIntegrator.cpp:30-39, CreatePathIntegrator :182-201
//   Scene (m_lights, worldBound)          Enigma/Scene.h:20-33
//   Camera / ProjectiveCamera / PerspectiveCamera public data   Enigma/Camera.h:40-76
//   Film / FilmTile / FilmTilePixel       Enigma/Film.h:24-114, Enigma/Film.cpp:39-125,157-209
//   Filter hierarchy                      Enigma/Filter.h:8-117
//   Sampler / SobolSampler public data    Enigma/Sampler.h:13-50,143-158, Enigma/Sampler.cpp:211-222
//   TriangleMesh public data              Enigma/TriangleMesh.h:28-37
// What differs, on purpose:
//  * Transform holds the FINAL numeric matrix (m_mat[r][c] as used by transformPoint,
//    Enigma/Transform.h:36-45); the reference's transpose-on-construct composition is not rebuilt.
//  * Scene carries a flat registry (meshes, materials, lights) filled at creation time: the
//    reference's primitives/materials are private behind shared_ptr graphs (Enigma/BVH.h:52,
//    Enigma/Primitive.h:59-62) and cannot be read back by an integrator.
//  * GpuPathIntegrator::Render does the work on the MI355X through the C ABI (enigma_hip.h) and
//    deposits the film through ONE FilmTile covering the cropped pixel bounds + MergeFilmTile +
//    WriteImage, exactly like the reference's tile tasks do (Enigma/Integrator.cpp:295-321).
#pragma once
#include <cstring>
#include <algorithm>
#include <deque>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/enigma_hip.h"

namespace RayTrace {

struct Vector2i { int x = 0, y = 0; Vector2i() {} Vector2i(int a, int b) : x(a), y(b) {} };
struct Vector2f { float x = 0, y = 0; Vector2f() {} Vector2f(float a, float b) : x(a), y(b) {} float operator[](int i) const { return i ? y : x; } };
struct Vector3f { float x = 0, y = 0, z = 0; Vector3f() {} Vector3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct BBox2i { Vector2i m_min, m_max; BBox2i() {} BBox2i(Vector2i a, Vector2i b) : m_min(a), m_max(b) {}
    int area() const { return (m_max.x - m_min.x) * (m_max.y - m_min.y); } };
struct BBox2f { Vector2f m_min{0, 0}, m_max{1, 1}; };
static constexpr float InfinityF32 = std::numeric_limits<float>::infinity();

// RGBSpectrum (Enigma/Spectrum.h:295-313): three floats
class Spectrum {
public:
    Spectrum(float v = 0.f) { c[0] = c[1] = c[2] = v; }
    Spectrum(float r, float g, float b) { c[0] = r; c[1] = g; c[2] = b; }
    float& operator[](int i) { return c[i]; }
    float operator[](int i) const { return c[i]; }
    void ToXYZ(float xyz[3]) const { // Enigma/Spectrum.h:28-32
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
    }
private:
    float c[3];
};
inline void XYZToRGB(const float xyz[3], float rgb[3]) { // Enigma/Spectrum.h:22-26
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

// ---- filters (Enigma/Filter.h, Enigma/Filter.cpp:8-32) ----
class Filter {
public:
    Filter(const Vector2f& r) : m_radius(r), m_invRadius(1.0f / r.x, 1.0f / r.y) {}
    virtual ~Filter() {}
    virtual float Evaluate(const Vector2f& xy) const = 0;
    const Vector2f m_radius, m_invRadius;
};
class BoxFilter : public Filter { public: BoxFilter(const Vector2f& r) : Filter(r) {} float Evaluate(const Vector2f&) const override { return 1.f; } };
class TriangleFilter : public Filter {
public:
    TriangleFilter(const Vector2f& r) : Filter(r) {}
    float Evaluate(const Vector2f& p) const override { return std::max((float)0, m_radius.x - std::abs(p.x)) * std::max((float)0, m_radius.y - std::abs(p.y)); }
};
class GaussianFilter : public Filter {
public:
    GaussianFilter(const Vector2f& r, float a) : Filter(r), m_alpha(a), m_expX(expf(-a * r.x * r.x)), m_expY(expf(-a * r.y * r.y)) {}
    float Evaluate(const Vector2f& p) const override { return G(p.x, m_expX) * G(p.y, m_expY); }
private:
    float G(float d, float e) const { return std::max(0.f, float(expf(-m_alpha * d * d) - e)); }
    const float m_alpha, m_expX, m_expY;
};

// ---- film (Enigma/Film.h, Enigma/Film.cpp) ----
struct FilmTilePixel { Spectrum contribSum = 0.f; float filterWeightSum = 0.f; };

class FilmTile {
public:
    FilmTile(const BBox2i& pixelBounds) : m_pixelBounds(pixelBounds), m_pixels(std::max(0, pixelBounds.area())) {}
    FilmTilePixel& GetPixel(const Vector2i& p) {
        int width = m_pixelBounds.m_max.x - m_pixelBounds.m_min.x;
        return m_pixels[(p.x - m_pixelBounds.m_min.x) + (p.y - m_pixelBounds.m_min.y) * width];
    }
    BBox2i GetPixelBounds() const { return m_pixelBounds; }
private:
    const BBox2i m_pixelBounds;
    std::vector<FilmTilePixel> m_pixels;
};

class Film {
public:
    Film(const Vector2i& resolution, const BBox2f& cropWindow, std::unique_ptr<Filter> filter, float diagonal, const std::string& filename,
         float scale, float maxSampleLuminance = InfinityF32)
        : m_fullResolution(resolution), m_diagonal(diagonal * .001f), m_filter(std::move(filter)), m_filename(filename), m_scale(scale),
          m_maxSampleLuminance(maxSampleLuminance) {
        m_croppedPixelBounds = BBox2i(Vector2i((int)std::ceil(resolution.x * cropWindow.m_min.x), (int)std::ceil(resolution.y * cropWindow.m_min.y)),
                                      Vector2i((int)std::ceil(resolution.x * cropWindow.m_max.x), (int)std::ceil(resolution.y * cropWindow.m_max.y)));
        m_pixels.assign((size_t)std::max(0, m_croppedPixelBounds.area()), Pixel());
        int offset = 0; // Enigma/Film.cpp:61-70
        for (int y = 0; y < filterTableWidth; ++y)
            for (int x = 0; x < filterTableWidth; ++x, ++offset) {
                Vector2f p((x + 0.5f) * m_filter->m_radius.x / filterTableWidth, (y + 0.5f) * m_filter->m_radius.y / filterTableWidth);
                m_filterTable[offset] = m_filter->Evaluate(p);
            }
    }
    BBox2i GetSampleBounds() const { // Enigma/Film.cpp:73-81
        float mnx = (float)m_croppedPixelBounds.m_min.x + 0.5f - m_filter->m_radius.x, mny = (float)m_croppedPixelBounds.m_min.y + 0.5f - m_filter->m_radius.y;
        float mxx = (float)m_croppedPixelBounds.m_max.x - 0.5f + m_filter->m_radius.x, mxy = (float)m_croppedPixelBounds.m_max.y - 0.5f + m_filter->m_radius.y;
        return BBox2i(Vector2i((int)std::floor(mnx), (int)std::floor(mny)), Vector2i((int)std::ceil(mxx), (int)std::ceil(mxy)));
    }
    std::unique_ptr<FilmTile> GetFilmTile(const BBox2i& sampleBounds) { // Enigma/Film.cpp:92-99
        int p0x = (int)std::ceil((float)sampleBounds.m_min.x - 0.5f - m_filter->m_radius.x), p0y = (int)std::ceil((float)sampleBounds.m_min.y - 0.5f - m_filter->m_radius.y);
        int p1x = (int)std::floor((float)sampleBounds.m_max.x - 0.5f + m_filter->m_radius.x) + 1, p1y = (int)std::floor((float)sampleBounds.m_max.y - 0.5f + m_filter->m_radius.y) + 1;
        BBox2i b(Vector2i(std::max(p0x, m_croppedPixelBounds.m_min.x), std::max(p0y, m_croppedPixelBounds.m_min.y)),
                 Vector2i(std::min(p1x, m_croppedPixelBounds.m_max.x), std::min(p1y, m_croppedPixelBounds.m_max.y)));
        return std::unique_ptr<FilmTile>(new FilmTile(b));
    }
    void MergeFilmTile(std::unique_ptr<FilmTile> tile) { // Enigma/Film.cpp:101-125
        const BBox2i b = tile->GetPixelBounds();
        for (int y = b.m_min.y; y < b.m_max.y; ++y)
            for (int x = b.m_min.x; x < b.m_max.x; ++x) {
                const FilmTilePixel& tp = tile->GetPixel(Vector2i(x, y));
                Pixel& mp = GetPixel(Vector2i(x, y));
                float xyz[3];
                tp.contribSum.ToXYZ(xyz);
                for (int i = 0; i < 3; ++i) mp.xyz[i] += xyz[i];
                mp.filterWeightSum += tp.filterWeightSum;
            }
    }
    // Enigma/Film.cpp:157-209.  Keeps the RGB image (m_rgb) and writes a PFM when a file name is set;
    // the reference's TGA/EXR writers (Enigma/IO.cpp) are outside this path.
    virtual void WriteImage(float splatScale) {
        (void)splatScale; // PathIntegrator adds no splats
        const BBox2i& b = m_croppedPixelBounds;
        m_rgb.assign((size_t)3 * std::max(0, b.area()), 0.f);
        int offset = 0;
        for (int y = b.m_min.y; y < b.m_max.y; ++y)
            for (int x = b.m_min.x; x < b.m_max.x; ++x, ++offset) {
                Pixel& pixel = GetPixel(Vector2i(x, y));
                float* rgb = &m_rgb[3 * (size_t)offset];
                XYZToRGB(pixel.xyz, rgb);
                float w = pixel.filterWeightSum;
                if (w != 0) { float invWt = (float)1 / w; for (int c = 0; c < 3; ++c) rgb[c] = std::max((float)0, rgb[c] * invWt); }
                for (int c = 0; c < 3; ++c) rgb[c] *= m_scale;
            }
        if (!m_filename.empty()) {
            int w = b.m_max.x - b.m_min.x, h = b.m_max.y - b.m_min.y;
            if (FILE* f = fopen(m_filename.c_str(), "wb")) {
                fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
                for (int y = h - 1; y >= 0; --y) fwrite(&m_rgb[(size_t)3 * y * w], sizeof(float), (size_t)3 * w, f);
                fclose(f);
            }
        }
    }
    void Clear() { for (auto& p : m_pixels) p = Pixel(); }
    const std::vector<float>& GetRGB() const { return m_rgb; }
    const float* GetFilterTable() const { return m_filterTable; }
    float GetMaxSampleLuminance() const { return m_maxSampleLuminance; }

    const Vector2i m_fullResolution;
    const float m_diagonal;
    std::unique_ptr<Filter> m_filter;
    const std::string m_filename;
    BBox2i m_croppedPixelBounds;
    struct Pixel { float xyz[3] = {0, 0, 0}; float filterWeightSum = 0; };
    Pixel& GetPixel(const Vector2i& p) {
        int width = m_croppedPixelBounds.m_max.x - m_croppedPixelBounds.m_min.x;
        return m_pixels[(p.x - m_croppedPixelBounds.m_min.x) + (p.y - m_croppedPixelBounds.m_min.y) * width];
    }
private:
    static constexpr int filterTableWidth = 16;
    std::vector<Pixel> m_pixels;
    float m_filterTable[filterTableWidth * filterTableWidth];
    const float m_scale, m_maxSampleLuminance;
    std::vector<float> m_rgb;
};

// ---- transforms / camera (Enigma/Camera.h:40-76) ----
class Transform { public: float m_mat[4][4]; Transform() { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m_mat[r][c] = r == c; } };

class Camera {
public:
    Camera(const Transform& c2w, float sOpen, float sClose, std::unique_ptr<Film> film)
        : m_cameraToWorld(c2w), m_shutterOpen(sOpen), m_shutterClose(sClose), m_film(std::move(film)) {}
    virtual ~Camera() {}
    const Film& GetFilm() const { return *m_film; }
    Transform m_cameraToWorld;
    const float m_shutterOpen, m_shutterClose;
    std::unique_ptr<Film> m_film;
};
class ProjectiveCamera : public Camera {
public:
    ProjectiveCamera(const Transform& c2w, const Transform& rasterToCamera, float sopen, float sclose, float lensr, float focald, std::unique_ptr<Film> film)
        : Camera(c2w, sopen, sclose, std::move(film)), m_rasterToCamera(rasterToCamera), m_lensRadius(lensr), m_focalDistance(focald) {}
    Transform m_rasterToCamera;
    float m_lensRadius, m_focalDistance;
};
class PerspectiveCamera : public ProjectiveCamera { public: using ProjectiveCamera::ProjectiveCamera; };
class EnvironmentCamera : public Camera { public: using Camera::Camera; }; // Enigma/Camera.h:96-104
using CameraPtr = std::shared_ptr<Camera>;

// ---- sampler (Enigma/Sampler.h:13-50,143-158) ----
class Sampler { public: Sampler(int64_t spp) : m_samplesPerPixel(spp) {} virtual ~Sampler() {} const int64_t m_samplesPerPixel; };
class SobolSampler : public Sampler {
public:
    static int64_t RoundUpPow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }
    SobolSampler(int64_t samplesPerPixel, const BBox2i& sampleBounds) : Sampler(RoundUpPow2(samplesPerPixel)), m_sampleBounds(sampleBounds) {
        m_resolution = (int)RoundUpPow2(std::max(sampleBounds.m_max.x - sampleBounds.m_min.x, sampleBounds.m_max.y - sampleBounds.m_min.y));
        m_log2Resolution = 0;
        while ((1 << m_log2Resolution) < m_resolution) ++m_log2Resolution;
    }
    const BBox2i m_sampleBounds;
    int m_resolution, m_log2Resolution;
};
// Enigma/Sampler.h:115-141, Enigma/Sampler.cpp:305-334.  The reference keeps m_sampleAtPixelCenter private and does not
// keep the sample bounds at all: inside Enigma the binding notes the flag where the sampler is created
// (CreateHaltonSampler, Enigma/Sampler.cpp:521-528) and takes the bounds from the film, as that factory does.
class HaltonSampler : public Sampler {
public:
    HaltonSampler(int nsamp, const BBox2i& sampleBounds, bool sampleAtCenter = false) : Sampler(nsamp), m_sampleBounds(sampleBounds), m_sampleAtPixelCenter(sampleAtCenter) {}
    const BBox2i m_sampleBounds;
    const bool m_sampleAtPixelCenter;
};
using SamplerPtr = std::shared_ptr<Sampler>;
// what an integrator desc takes from either GlobalSampler
struct SamplerFields { int64_t m_samplesPerPixel; BBox2i m_sampleBounds; };

// ---- scene registry ----
struct TriangleMesh { // Enigma/TriangleMesh.h:28-37: world-space m_p, m_vertexIndices, optional m_n (world space) and m_uv
    int m_nTriangles = 0, m_nVertices = 0;
    std::vector<int> m_vertexIndices;
    std::vector<Vector3f> m_p;
    std::vector<Vector3f> m_n;      // empty or m_nVertices normals, already through ObjectToWorld.transformNormal
    std::vector<Vector2f> m_uv;     // empty or m_nVertices uvs
    int materialIndex = 0;          // registry index of the bound material; -1 = none (MakeMaterial "" / "none": a medium boundary)
    int mediumInside = -1, mediumOutside = -1; // GeometricPrimitive::m_mi: registry indices of the media, -1 = none
    bool reverseOrientation = false;        // Shape::m_reverseOrientation
    bool transformSwapsHandedness = false;  // Shape::m_transformSwapsHandedness
    bool emissive = false;          // an AreaLightSource was active when the shape was created
    Spectrum Le = 0.f;
    bool twoSided = false;
    int nSamples = 1;               // Light::m_nSamples of the area lights made for this mesh
};

class Scene {
public:
    // Registry calls: the equivalent of hooking MakeShapes / MakeMaterial / MakeLight (Enigma/Api.cpp:843-932).
    int AddMaterial(const EnigmaMaterial& m) { m_materials.push_back(m); return (int)m_materials.size() - 1; }
    // ImageTexture<RGBSpectrum, Spectrum>(UVMapping2D(su, sv, du, dv), TexInfo(...)) (Enigma/Texture.h:313-324,
    // Enigma/Texture.cpp:412-453) for texels already read: rgb is height x width x 3 as ReadImage returns it (top row
    // first); gamma is TexInfo::m_gamma, the exponent the reference applies (its bool "gamma" turned float).  Returns the
    // value for a material's tex_* field.
    uint32_t AddImageTexture(int width, int height, std::vector<float> rgb, bool trilinear, float gamma, float scale, uint32_t wrap,
                             float su = 1.f, float sv = 1.f, float du = 0.f, float dv = 0.f) {
        EnigmaTexture t{};
        t.width = (uint32_t)width; t.height = (uint32_t)height; t.scale = scale; t.gamma = gamma; t.trilinear = trilinear ? 1u : 0u; t.wrap = wrap;
        t.su = su; t.sv = sv; t.du = du; t.dv = dv;
        m_textureTexels.push_back(std::move(rgb));
        m_textures.push_back(t);
        return (uint32_t)m_textures.size();
    }
    // Sphere(o2w, w2o, reverseOrientation, radius, zMin, zMax, phiMax) with its GeometricPrimitive's material
    // and MediumInterface (registry indices, -1 = none) (Enigma/ShapeImp.cpp:13-27, Enigma/Api.cpp:843-874).  emission: the
    // radiance of its DiffuseAreaLight (CreateDiffuseAreaLight, Enigma/Lights.cpp:812-822), which takes its place among
    // the scene's lights in creation order; nullptr = not emissive.
    void AddSphere(const Transform& objectToWorld, const Transform& worldToObject, bool reverseOrientation, bool transformSwapsHandedness,
                   float radius, float zMin, float zMax, float phiMax, int materialIndex, int mediumInside = -1, int mediumOutside = -1,
                   const Spectrum* emission = nullptr, bool twoSided = false, int nSamples = 1) {
        EnigmaSphere s{};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { s.object_to_world[4 * r + c] = objectToWorld.m_mat[r][c]; s.world_to_object[4 * r + c] = worldToObject.m_mat[r][c]; }
        s.radius = radius; s.z_min = zMin; s.z_max = zMax; s.phi_max = phiMax;
        s.material = materialIndex < 0 ? ENIGMA_NO_MATERIAL : (uint32_t)materialIndex; // no material: a medium boundary
        s.medium_inside = mediumInside + 1; s.medium_outside = mediumOutside + 1; // the ABI holds index + 1
        s.flags = ((reverseOrientation ^ transformSwapsHandedness) ? ENIGMA_TRI_FLIP_NORMAL : 0u) | (reverseOrientation ? ENIGMA_TRI_REVERSE_ORIENTATION : 0u);
        if (emission) { // the sphere's `light` field is filled when the lights are laid out in order
            EnigmaLight l{}; l.type = ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE; l.triangle = (uint32_t)m_spheres.size(); l.two_sided = twoSided; l.n_samples = (uint32_t)std::max(1, nSamples);
            l.L[0] = (*emission)[0]; l.L[1] = (*emission)[1]; l.L[2] = (*emission)[2];
            AddLight(l);
        }
        m_spheres.push_back(s);
    }
    // HomogeneousMedium(sigma_a, sigma_s, g) (Enigma/Medium.cpp:69-76)
    int AddMedium(const EnigmaMedium& m) { m_media.push_back(m); return (int)m_media.size() - 1; }
    // One GeometricPrimitive (and one DiffuseAreaLight if emissive) per triangle, in file order (Enigma/Api.cpp:863-874).
    void AddTriangleMesh(const TriangleMesh& mesh) { m_meshes.push_back(mesh); }
    // Lights without a shape (point, spot, distant) keep their place among the meshes' area lights.
    void AddLight(const EnigmaLight& l, int mapIndex = -1) { m_lights.push_back(LightRec{l, NextLightOrder(), mapIndex}); }
    void AddPointLight(const Vector3f& from, const Spectrum& I) { // PointLight(light2world, mi, I): from = light2world(0,0,0)
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_POINT;
        l.pos[0] = from.x; l.pos[1] = from.y; l.pos[2] = from.z; l.L[0] = I[0]; l.L[1] = I[1]; l.L[2] = I[2];
        AddLight(l);
    }
    // ProjectionLight(light2world, mi, I, texname, fov) (Enigma/Lights.cpp:194-235).  The constructor drops the image it reads
    // (:220-221), the aspect ratio is therefore 1, and what the light keeps is its position, its intensity and the cosine of
    // the cone around the projection, which only Power() reads.
    void AddProjectionLight(const Vector3f& from, const Spectrum& I, float fov) {
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_PROJECTION;
        l.pos[0] = from.x; l.pos[1] = from.y; l.pos[2] = from.z; l.L[0] = I[0]; l.L[1] = I[1]; l.L[2] = I[2];
        const float aspect = 1.f;
        float opposite = tanf(3.14159265358979323846f / 180.f * fov / 2.f);
        float tanDiag = opposite * sqrtf(1.f + 1.f / (aspect * aspect));
        l.cos_total_width = cosf(atanf(tanDiag));
        AddLight(l);
    }
    // GonioPhotometricLight(light2world, mi, I, texname) (Enigma/Lights.cpp:301-317): position, intensity and the texels
    // ReadImage returned (none: no map).  m_gonioMap is private in the reference; the binding keeps the texels where
    // CreateGoniometricLight makes the light (Enigma/Lights.cpp:774-780).
    void AddGoniometricLight(const Vector3f& from, const Spectrum& I, std::vector<float> map = std::vector<float>(), int mapWidth = 0, int mapHeight = 0) {
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_GONIOMETRIC;
        l.pos[0] = from.x; l.pos[1] = from.y; l.pos[2] = from.z; l.L[0] = I[0]; l.L[1] = I[1]; l.L[2] = I[2];
        int mapIndex = -1;
        if (!map.empty()) {
            m_lightMaps.push_back(std::move(map));
            mapIndex = (int)m_lightMaps.size() - 1; l.map_width = (uint32_t)mapWidth; l.map_height = (uint32_t)mapHeight;
        }
        AddLight(l, mapIndex);
    }
    // SpotLight(light2world, mi, I, width, fall) (Enigma/Lights.cpp:133-140): position light2world(0,0,0), the cosines
    // of the two cone angles (degrees), and the upper 3x3 of the inverse for Falloff.
    void AddSpotLight(const Transform& lightToWorld, const Transform& worldToLight, const Spectrum& I, float width, float fall) {
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_SPOT;
        for (int c = 0; c < 3; ++c) { l.pos[c] = lightToWorld.m_mat[c][3]; l.L[c] = I[c]; }
        const float toRad = 3.14159265358979323846f / 180.f;
        l.cos_total_width = cosf(toRad * width);
        l.cos_falloff_start = cosf(toRad * fall);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) l.world_to_light[3 * r + c] = worldToLight.m_mat[r][c];
        AddLight(l);
    }
    // DistantLight(light2world, L, dir) (Enigma/Lights.cpp:656-663): m_lightDir = light2world.transformVector(dir), not normalised.
    void AddDistantLight(const Transform& lightToWorld, const Spectrum& L, const Vector3f& dir) {
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_DISTANT;
        for (int r = 0; r < 3; ++r) l.dir[r] = lightToWorld.m_mat[r][0] * dir.x + lightToWorld.m_mat[r][1] * dir.y + lightToWorld.m_mat[r][2] * dir.z;
        for (int c = 0; c < 3; ++c) l.L[c] = L[c];
        AddLight(l);
    }
    // InfiniteAreaLight(light2world, L, nSamples, texmap) (Enigma/Lights.cpp:395-442); without a map a constant environment.
    // map: the texels ReadImage returns for "mapname" (RGB, top row first = theta 0), kept by the scene until the upload
    void AddInfiniteLight(const Transform& lightToWorld, const Transform& worldToLight, const Spectrum& L, int nSamples = 1,
                          std::vector<float> map = std::vector<float>(), int mapWidth = 0, int mapHeight = 0) {
        EnigmaLight l{}; l.type = ENIGMA_LIGHT_INFINITE; l.n_samples = (uint32_t)std::max(1, nSamples);
        int mapIndex = -1;
        if (!map.empty()) { // map_rgb is pointed at the scene's copy when the lights are handed to the library (a Scene may be copied before)
            m_lightMaps.push_back(std::move(map));
            mapIndex = (int)m_lightMaps.size() - 1; l.map_width = (uint32_t)mapWidth; l.map_height = (uint32_t)mapHeight;
        }
        for (int c = 0; c < 3; ++c) l.L[c] = L[c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { l.light_to_world[3 * r + c] = lightToWorld.m_mat[r][c]; l.world_to_light[3 * r + c] = worldToLight.m_mat[r][c]; }
        AddLight(l, mapIndex);
    }
    int getNumLights() const {
        int n = (int)m_lights.size();
        for (auto& m : m_meshes) if (m.emissive) n += m.m_nTriangles;
        return n;
    }
    struct LightRec { EnigmaLight light; int order; int map; }; // map: index into m_lightMaps, -1 = none
    std::deque<std::vector<float>> m_lightMaps; // maps of the infinite and goniometric lights
    std::vector<TriangleMesh> m_meshes;
    std::vector<EnigmaMaterial> m_materials;
    std::vector<EnigmaMedium> m_media;
    std::vector<LightRec> m_lights;
    std::vector<EnigmaSphere> m_spheres;
    std::vector<EnigmaTexture> m_textures;              // rgb pointers are filled in at Render time
    std::vector<std::vector<float>> m_textureTexels;
private:
    int NextLightOrder() const { return (int)m_meshes.size(); } // point lights keep their place among the meshes' lights
};

// ---- integrators (Enigma/Integrator.h:35-70) ----
class Integrator {
public:
    virtual ~Integrator() {}
    virtual void Render(const Scene& scene) = 0;
    virtual const CameraPtr& GetCamera() const = 0;
};

class SamplerIntegrator : public Integrator {
public:
    SamplerIntegrator(const CameraPtr& camera, std::shared_ptr<Sampler> sampler, const BBox2i& pixelBounds)
        : m_camera(camera), m_sampler(sampler), m_pixelBounds(pixelBounds) {}
    virtual void Preprocess(const Scene&, Sampler&) {}
    const BBox2i& GetPixelBounds() const { return m_pixelBounds; }
    const SamplerPtr& GetSampler() const { return m_sampler; }
    const CameraPtr& GetCamera() const override { return m_camera; }
protected:
    CameraPtr m_camera;
private:
    std::shared_ptr<Sampler> m_sampler;
    const BBox2i m_pixelBounds;
};

// SamplerIntegrator whose Render runs on the MI355X: flattens the scene, hands camera, film and sampler state
// to the library, renders, and deposits the device film through the reference's own MergeFilmTile / WriteImage.
// Subclasses configure the integrator proper.
class GpuSamplerIntegrator : public SamplerIntegrator {
public:
    GpuSamplerIntegrator(std::shared_ptr<Camera> camera, std::shared_ptr<Sampler> sampler, const BBox2i& pixelBounds, int device)
        : SamplerIntegrator(camera, sampler, pixelBounds), m_device(device) {}

    // Errors: the reference prints and continues (Enigma/Error.h:5-10); a failing GPU call cannot
    // "continue", so it throws std::runtime_error carrying enigma_last_error().
    void Render(const Scene& scene) override {
        EnigmaCtx* h = nullptr;
        if (enigma_ctx_create(m_device, &h) != ENIGMA_OK) throw std::runtime_error("enigma_ctx_create failed: no HIP device");
        struct Guard { EnigmaCtx* h; ~Guard() { enigma_ctx_destroy(h); } } guard{h};
        auto check = [&](int rc) { if (rc != ENIGMA_OK) throw std::runtime_error(std::string("enigma: ") + enigma_last_error(h)); };
        // flatten the registry
        std::vector<EnigmaSphere> spheres(scene.m_spheres);
        std::vector<float> verts, normals, uvs; std::vector<uint32_t> idx, triMat, triFlags; std::vector<int32_t> triLight, triIn, triOut; std::vector<EnigmaLight> lights;
        size_t pl = 0;
        auto flushPointLights = [&](int upTo) {
            for (; pl < scene.m_lights.size() && scene.m_lights[pl].order <= upTo; ++pl) {
                lights.push_back(scene.m_lights[pl].light);
                if (scene.m_lights[pl].map >= 0) lights.back().map_rgb = scene.m_lightMaps[(size_t)scene.m_lights[pl].map].data();
                if (lights.back().type == ENIGMA_LIGHT_DIFFUSE_AREA_SPHERE) spheres[lights.back().triangle].light = (uint32_t)lights.size();
            }
        };
        for (size_t mi = 0; mi < scene.m_meshes.size(); ++mi) {
            flushPointLights((int)mi);
            const TriangleMesh& m = scene.m_meshes[mi];
            uint32_t base = (uint32_t)(verts.size() / 3);
            for (const Vector3f& p : m.m_p) { verts.push_back(p.x); verts.push_back(p.y); verts.push_back(p.z); }
            const bool hasN = (int)m.m_n.size() == m.m_nVertices && m.m_nVertices > 0, hasUV = (int)m.m_uv.size() == m.m_nVertices && m.m_nVertices > 0;
            if (hasN) { normals.resize(3 * (size_t)base, 0.f); for (const Vector3f& n : m.m_n) { normals.push_back(n.x); normals.push_back(n.y); normals.push_back(n.z); } }
            if (hasUV) { uvs.resize(2 * (size_t)base, 0.f); for (const Vector2f& t : m.m_uv) { uvs.push_back(t.x); uvs.push_back(t.y); } }
            for (int t = 0; t < m.m_nTriangles; ++t) {
                for (int k = 0; k < 3; ++k) idx.push_back(base + (uint32_t)m.m_vertexIndices[3 * t + k]);
                triMat.push_back(m.materialIndex < 0 ? ENIGMA_NO_MATERIAL : (uint32_t)m.materialIndex);
                triIn.push_back(m.mediumInside); triOut.push_back(m.mediumOutside);
                triFlags.push_back(((m.reverseOrientation ^ m.transformSwapsHandedness) ? ENIGMA_TRI_FLIP_NORMAL : 0u) | (hasN ? ENIGMA_TRI_HAS_NORMALS : 0u) |
                                   (hasUV ? ENIGMA_TRI_HAS_UVS : 0u) | (m.reverseOrientation ? ENIGMA_TRI_REVERSE_ORIENTATION : 0u));
                if (m.emissive) {
                    EnigmaLight l{}; l.type = ENIGMA_LIGHT_DIFFUSE_AREA; l.triangle = (uint32_t)triMat.size() - 1; l.two_sided = m.twoSided; l.n_samples = (uint32_t)std::max(1, m.nSamples);
                    l.L[0] = m.Le[0]; l.L[1] = m.Le[1]; l.L[2] = m.Le[2];
                    triLight.push_back((int32_t)lights.size());
                    lights.push_back(l);
                } else triLight.push_back(-1);
            }
        }
        flushPointLights(std::numeric_limits<int>::max());
        EnigmaSceneDesc sd{};
        sd.n_vertices = (uint32_t)(verts.size() / 3); sd.n_triangles = (uint32_t)triMat.size();
        sd.vertices = verts.data(); sd.indices = idx.data(); sd.tri_material = triMat.data(); sd.tri_light = triLight.data(); sd.tri_flags = triFlags.data();
        if (!normals.empty()) { normals.resize(verts.size(), 0.f); sd.normals = normals.data(); }
        if (!uvs.empty()) { uvs.resize(verts.size() / 3 * 2, 0.f); sd.uvs = uvs.data(); }
        sd.n_materials = (uint32_t)scene.m_materials.size(); sd.materials = scene.m_materials.data();
        sd.n_lights = (uint32_t)lights.size(); sd.lights = lights.data();
        if (!scene.m_media.empty()) { sd.n_media = (uint32_t)scene.m_media.size(); sd.media = scene.m_media.data(); sd.tri_medium_inside = triIn.data(); sd.tri_medium_outside = triOut.data(); }
        std::vector<EnigmaTexture> textures(scene.m_textures);
        for (size_t i = 0; i < textures.size(); ++i) textures[i].rgb = scene.m_textureTexels[i].data();
        sd.n_textures = (uint32_t)textures.size(); sd.textures = textures.empty() ? nullptr : textures.data();
        sd.n_spheres = (uint32_t)spheres.size(); sd.spheres = spheres.empty() ? nullptr : spheres.data();
        check(enigma_scene_upload(h, &sd));

        const ProjectiveCamera* cam = dynamic_cast<const ProjectiveCamera*>(m_camera.get());
        const bool environment = dynamic_cast<const EnvironmentCamera*>(m_camera.get()) != nullptr;
        if (!cam && !environment) throw std::runtime_error("the GPU integrators need a PerspectiveCamera or an EnvironmentCamera");
        EnigmaCameraDesc cd{};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
            cd.raster_to_camera[4 * r + c] = cam ? cam->m_rasterToCamera.m_mat[r][c] : (r == c ? 1.f : 0.f);
            cd.camera_to_world[4 * r + c] = m_camera->m_cameraToWorld.m_mat[r][c];
        }
        cd.lens_radius = cam ? cam->m_lensRadius : 0.f; cd.focal_distance = cam ? cam->m_focalDistance : 1e30f;
        cd.type = environment ? ENIGMA_CAMERA_ENVIRONMENT : ENIGMA_CAMERA_PERSPECTIVE;
        check(enigma_camera_set(h, &cd));

        Film& film = *m_camera->m_film;
        EnigmaFilmDesc fd{};
        fd.full_resolution[0] = film.m_fullResolution.x; fd.full_resolution[1] = film.m_fullResolution.y;
        const BBox2i& cb = film.m_croppedPixelBounds;
        fd.cropped_pixel_bounds[0] = cb.m_min.x; fd.cropped_pixel_bounds[1] = cb.m_min.y; fd.cropped_pixel_bounds[2] = cb.m_max.x; fd.cropped_pixel_bounds[3] = cb.m_max.y;
        fd.filter_radius[0] = film.m_filter->m_radius.x; fd.filter_radius[1] = film.m_filter->m_radius.y;
        std::copy(film.GetFilterTable(), film.GetFilterTable() + 256, fd.filter_table);
        fd.max_sample_luminance = film.GetMaxSampleLuminance();
        check(enigma_film_config(h, &fd));

        const SamplerFields samplerFields = SelectSampler(h);
        const SamplerFields* sob = &samplerFields;
        check(ConfigureIntegrator(h, lights.size(), *sob));

        // one process per GPU: this rank's share of the sample numbers, then the films are summed onto rank 0
        const int64_t spp = sob->m_samplesPerPixel;
        if (m_world > 1) check(enigma_comm_init(h, m_rank, m_world, m_commId));
        check(enigma_render(h, spp * m_rank / m_world, spp * (m_rank + 1) / m_world));
        check(enigma_film_reduce(h, 0));
        check(enigma_get_stats(h, &m_stats));
        if (m_rank != 0) return; // the film lives on rank 0

        // deposit: one FilmTile over the cropped bounds, then the reference's own merge + write
        std::vector<float> rgbw((size_t)4 * std::max(0, cb.area()));
        check(enigma_film_read(h, rgbw.data(), rgbw.size()));
        std::unique_ptr<FilmTile> tile(new FilmTile(cb));
        size_t o = 0;
        for (int y = cb.m_min.y; y < cb.m_max.y; ++y)
            for (int x = cb.m_min.x; x < cb.m_max.x; ++x, o += 4) {
                FilmTilePixel& px = tile->GetPixel(Vector2i(x, y));
                px.contribSum = Spectrum(rgbw[o], rgbw[o + 1], rgbw[o + 2]);
                px.filterWeightSum = rgbw[o + 3];
            }
        film.MergeFilmTile(std::move(tile));
        film.WriteImage(1.0f / sob->m_samplesPerPixel);
    }
    const EnigmaStats& GetStats() const { return m_stats; }
    // Several GPUs, one process each: every process builds the same scene and integrator, names its rank, and passes
    // the 128 bytes rank 0 got from enigma_comm_unique_id (shipped by whatever channel the application has).  Render
    // then takes sample numbers [spp * rank / world, spp * (rank + 1) / world) and rank 0 ends up with the whole film.
    void SetRanks(int rank, int world, const void* commId) {
        m_rank = rank; m_world = world;
        std::memcpy(m_commId, commId, ENIGMA_COMM_ID_BYTES);
    }
protected:
    // tells the library which GlobalSampler m_sampler is (Enigma/Api.cpp:425-434 made it by name)
    SamplerFields SelectSampler(EnigmaCtx* h) const {
        auto check = [&](int rc) { if (rc != ENIGMA_OK) throw std::runtime_error(std::string("enigma: ") + enigma_last_error(h)); };
        if (const SobolSampler* sob = dynamic_cast<const SobolSampler*>(GetSampler().get())) {
            check(enigma_sampler_select(h, ENIGMA_SAMPLER_SOBOL, 0));
            return SamplerFields{sob->m_samplesPerPixel, sob->m_sampleBounds};
        }
        if (const HaltonSampler* hal = dynamic_cast<const HaltonSampler*>(GetSampler().get())) {
            check(enigma_sampler_select(h, ENIGMA_SAMPLER_HALTON, hal->m_sampleAtPixelCenter ? 1u : 0u));
            return SamplerFields{hal->m_samplesPerPixel, hal->m_sampleBounds};
        }
        throw std::runtime_error("the GPU integrators need a SobolSampler or a HaltonSampler");
    }
    virtual int ConfigureIntegrator(EnigmaCtx* h, size_t nLights, const SamplerFields& sob) = 0;
    template <class DESC> void FillSamplerFields(DESC& id, const SamplerFields& sob) const {
        const BBox2i& pb = GetPixelBounds();
        id.pixel_bounds[0] = pb.m_min.x; id.pixel_bounds[1] = pb.m_min.y; id.pixel_bounds[2] = pb.m_max.x; id.pixel_bounds[3] = pb.m_max.y;
        id.samples_per_pixel = sob.m_samplesPerPixel;
        id.sample_bounds[0] = sob.m_sampleBounds.m_min.x; id.sample_bounds[1] = sob.m_sampleBounds.m_min.y;
        id.sample_bounds[2] = sob.m_sampleBounds.m_max.x; id.sample_bounds[3] = sob.m_sampleBounds.m_max.y;
    }
private:
    const int m_device;
    int m_rank = 0, m_world = 1;
    unsigned char m_commId[ENIGMA_COMM_ID_BYTES] = {};
    EnigmaStats m_stats{};
};

// PathIntegrator on the GPU (constructor arguments: Enigma/PathIntegrator.cpp:30-39).
class GpuPathIntegrator : public GpuSamplerIntegrator {
public:
    GpuPathIntegrator(int maxDepth, std::shared_ptr<Camera> camera, std::shared_ptr<Sampler> sampler, const BBox2i& pixelBounds,
                      float rrThreshold = 1, const std::string& lightSampleStrategy = "spatial", int device = 0)
        : GpuSamplerIntegrator(camera, sampler, pixelBounds, device), m_maxDepth(maxDepth), m_rrThreshold(rrThreshold),
          m_lightSampleStrategy(lightSampleStrategy) {}
protected:
    int ConfigureIntegrator(EnigmaCtx* h, size_t nLights, const SamplerFields& sob) override {
        EnigmaIntegratorDesc id{};
        id.max_depth = m_maxDepth; id.rr_threshold = m_rrThreshold;
        if (m_lightSampleStrategy == "uniform") id.light_strategy = ENIGMA_LIGHTS_UNIFORM;
        else if (m_lightSampleStrategy == "power") id.light_strategy = ENIGMA_LIGHTS_POWER;
        else id.light_strategy = ENIGMA_LIGHTS_SPATIAL; // "spatial", and unknown names fall back to it (Enigma/LightDist.cpp:28-37); one light is uniform inside the library
        FillSamplerFields(id, sob);
        return enigma_integrator_config(h, &id);
    }
private:
    const int m_maxDepth;
    const float m_rrThreshold;
    const std::string m_lightSampleStrategy;
};

// VolPathIntegrator on the GPU (constructor arguments: Enigma/VolPathIntegrator.cpp:27-35).
class GpuVolPathIntegrator : public GpuSamplerIntegrator {
public:
    GpuVolPathIntegrator(int maxDepth, std::shared_ptr<Camera> camera, std::shared_ptr<Sampler> sampler, const BBox2i& pixelBounds,
                         float rrThreshold = 1, const std::string& lightSampleStrategy = "spatial", int device = 0)
        : GpuSamplerIntegrator(camera, sampler, pixelBounds, device), m_maxDepth(maxDepth), m_rrThreshold(rrThreshold),
          m_lightSampleStrategy(lightSampleStrategy) {}
protected:
    int ConfigureIntegrator(EnigmaCtx* h, size_t nLights, const SamplerFields& sob) override {
        EnigmaIntegratorDesc id{};
        id.max_depth = m_maxDepth; id.rr_threshold = m_rrThreshold;
        if (m_lightSampleStrategy == "uniform") id.light_strategy = ENIGMA_LIGHTS_UNIFORM;
        else if (m_lightSampleStrategy == "power") id.light_strategy = ENIGMA_LIGHTS_POWER;
        else id.light_strategy = ENIGMA_LIGHTS_SPATIAL; // "spatial", and unknown names fall back to it (Enigma/LightDist.cpp:28-37); one light is uniform inside the library
        FillSamplerFields(id, sob);
        return enigma_volpath_config(h, &id);
    }
private:
    const int m_maxDepth;
    const float m_rrThreshold;
    const std::string m_lightSampleStrategy;
};

// DirectLightingIntegrator on the GPU (constructor arguments: Enigma/DirectLightingIntegrator.cpp:30-35).
class GpuDirectLightingIntegrator : public GpuSamplerIntegrator {
public:
    enum class eLightStrategy { UniformSampleAll, UniformSampleOne }; // Enigma/DirectLightingIntegrator.h
    GpuDirectLightingIntegrator(eLightStrategy strategy, int maxDepth, std::shared_ptr<Camera> camera, std::shared_ptr<Sampler> sampler,
                                const BBox2i& pixelBounds, int device = 0)
        : GpuSamplerIntegrator(camera, sampler, pixelBounds, device), m_strategy(strategy), m_maxDepth(maxDepth) {}
protected:
    int ConfigureIntegrator(EnigmaCtx* h, size_t, const SamplerFields& sob) override {
        EnigmaDirectLightingDesc id{};
        id.max_depth = m_maxDepth;
        id.strategy = m_strategy == eLightStrategy::UniformSampleAll ? ENIGMA_DL_SAMPLE_ALL : ENIGMA_DL_SAMPLE_ONE;
        FillSamplerFields(id, sob);
        return enigma_direct_lighting_config(h, &id);
    }
private:
    const eLightStrategy m_strategy;
    const int m_maxDepth;
};

} // namespace RayTrace
