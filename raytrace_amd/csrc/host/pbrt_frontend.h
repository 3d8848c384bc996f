// Scene-description front end: the reference's pbrt-format parser and pbrt* API for the part of the
// format the GPU integrators can render, written against the mirrored classes of enigma_host.h.
//
// Replaces, for that subset: the tokenizer / directive loop of Enigma/Parser.cpp:1-1140, ParamSet
// (Enigma/ParameterSet.h:14-175), the API state machine (Enigma/Api.cpp:483-1254: transforms, option block,
// attribute stack, named materials, shapes, lights, WorldEnd -> MakeIntegrator / MakeScene / Render) and the
// Create* factories it calls (Film.cpp:330-372, Filter.cpp:38-75, Camera.cpp:385-432, Sampler.cpp:554-559,
// PathIntegrator.cpp:182-201, DirectLightingIntegrator.cpp:89-114, Material.cpp:594-646, Lights.cpp:742-750,
// 812-822).  Parameter names and defaults are the reference's.
//
// Deliberate divergence: transforms compose as the pbrt format defines them (CTM = CTM * T, points map
// through the CTM, the camera transform is Inverse(CTM)), evaluated in double and rounded once.  The
// reference's Transform composition transposes on construction (Enigma/Transform.cpp:295-314) and gives
// wrong matrices for chained operations; that defect is not reproduced.
//
// Inside the subset: trianglemesh (with N, uv / st), plymesh and sphere shapes; matte, metal (default spectra: the reference's
// copper), glass, plastic, mirror, translucent, substrate, uber, mix and "none" materials; constant and imagemap (PFM)
// textures, constant float textures, scale and mix of constant operands; point, spot, distant, projection, goniometric,
// infinite and diffuse area lights; named media (homogeneous with the measured presets; heterogeneous as the reference has
// them: black) and medium interfaces; the path, directlighting and volpath integrators; the sobol and halton samplers
// (halton when the file names none, as in the reference); the perspective and environment cameras; the image film with
// its five filters.  Outside it (other shapes and texture classes, spectra other than rgb, object instancing, the pixel
// samplers, other cameras / integrators) the front end throws std::runtime_error naming the directive; it never renders
// something else silently.
#pragma once
#include "enigma_host.h"
#include "medium_presets.inc"
#include "ply_reader.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace RayTrace {

// ---------------------------------------------------------------- ParamSet (Enigma/ParameterSet.h:14-175)
struct ParamSetItem {
    std::string type, name;
    std::vector<double> nums;
    std::vector<std::string> strs;
    mutable bool lookedUp = false;
};

class ParamSet {
public:
    void Add(const ParamSetItem& it) { m_items.push_back(it); }
    int FindOneInt(const std::string& n, int d) const { auto* i = find(n, "integer", 1); return i ? (int)i->nums[0] : d; }
    float FindOneFloat(const std::string& n, float d) const { auto* i = find(n, "float", 1); return i ? (float)i->nums[0] : d; }
    bool FindOneBool(const std::string& n, bool d) const { auto* i = find(n, "bool", 1); return i ? i->strs[0] == "true" : d; }
    std::string FindOneString(const std::string& n, const std::string& d) const { auto* i = find(n, "string", 1); return i ? i->strs[0] : d; }
    Vector3f FindOnePoint3f(const std::string& n, const Vector3f& d) const {
        auto* i = find(n, "point", 3);
        return i ? Vector3f((float)i->nums[0], (float)i->nums[1], (float)i->nums[2]) : d;
    }
    // rgb / color only: sampled spectra, blackbody and xyz need the CIE tables of Enigma/Spectrum.cpp
    Spectrum FindOneSpectrum(const std::string& n, const Spectrum& d) const {
        for (const auto& it : m_items)
            if (it.name == n && (it.type == "spectrum" || it.type == "blackbody" || it.type == "xyz"))
                throw std::runtime_error("parameter \"" + it.type + " " + n + "\": only rgb / color values are supported");
        auto* i = find(n, "rgb", 3);
        return i ? Spectrum((float)i->nums[0], (float)i->nums[1], (float)i->nums[2]) : d;
    }
    bool HasSpectrum(const std::string& n) const { return find(n, "rgb", 3, false) != nullptr; }
    bool Has(const std::string& n) const { for (const auto& it : m_items) if (it.name == n) return true; return false; }
    std::vector<int> FindInt(const std::string& n) const {
        std::vector<int> out;
        if (auto* i = find(n, "integer", 0)) for (double v : i->nums) out.push_back((int)v);
        return out;
    }
    std::vector<float> FindFloat(const std::string& n) const {
        std::vector<float> out;
        if (auto* i = find(n, "float", 0)) for (double v : i->nums) out.push_back((float)v);
        return out;
    }
    std::vector<double> FindPoint3(const std::string& n) const { auto* i = find(n, "point", 0); return i ? i->nums : std::vector<double>(); }
    std::vector<double> FindNormal3(const std::string& n) const { auto* i = find(n, "normal", 0); return i ? i->nums : std::vector<double>(); }
    std::vector<double> FindVector3(const std::string& n) const { auto* i = find(n, "vector", 0); return i ? i->nums : std::vector<double>(); }
    std::vector<double> FindPoint2(const std::string& n) const { auto* i = find(n, "point2", 0); return i ? i->nums : std::vector<double>(); }
    std::string FindTexture(const std::string& n) const { auto* i = find(n, "texture", 1, false); return i ? i->strs[0] : ""; }
    std::vector<std::string> Unused() const { // ReportUnused
        std::vector<std::string> out;
        for (const auto& it : m_items) if (!it.lookedUp) out.push_back(it.type + " " + it.name);
        return out;
    }
    const std::vector<ParamSetItem>& Items() const { return m_items; }
private:
    static std::string canon(const std::string& t) {
        if (t == "color") return "rgb";
        if (t == "point3" || t == "point3f") return "point";
        if (t == "vector3" || t == "vector3f") return "vector";
        if (t == "normal3" || t == "normal3f") return "normal";
        return t;
    }
    const ParamSetItem* find(const std::string& n, const std::string& type, size_t minCount, bool mark = true) const {
        for (const auto& it : m_items)
            if (it.name == n && canon(it.type) == type) {
                size_t have = (type == "bool" || type == "string" || type == "texture") ? it.strs.size() : it.nums.size();
                if (minCount && have < minCount) throw std::runtime_error("parameter \"" + n + "\" has too few values");
                if (mark) it.lookedUp = true;
                return &it;
            }
        return nullptr;
    }
    std::vector<ParamSetItem> m_items;
};

// ---------------------------------------------------------------- 4x4 algebra in double (Enigma/Transform.h)
struct Mat4d {
    double m[4][4];
    Mat4d() { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m[r][c] = r == c; }
    Mat4d operator*(const Mat4d& b) const {
        Mat4d o;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += m[r][k] * b.m[k][c];
                o.m[r][c] = s;
            }
        return o;
    }
    Mat4d Inverse() const { // Gauss-Jordan with full pivoting
        double a[4][8];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[r][c]; a[r][4 + c] = r == c; }
        for (int col = 0; col < 4; ++col) {
            int piv = col;
            for (int r = col + 1; r < 4; ++r) if (std::abs(a[r][col]) > std::abs(a[piv][col])) piv = r;
            if (a[piv][col] == 0) throw std::runtime_error("singular matrix in transform inversion");
            if (piv != col) for (int c = 0; c < 8; ++c) std::swap(a[piv][c], a[col][c]);
            double inv = 1.0 / a[col][col];
            for (int c = 0; c < 8; ++c) a[col][c] *= inv;
            for (int r = 0; r < 4; ++r)
                if (r != col && a[r][col] != 0) {
                    double f = a[r][col];
                    for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
                }
        }
        Mat4d o;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o.m[r][c] = a[r][4 + c];
        return o;
    }
    void Point(const double p[3], double out[3]) const {
        double x = m[0][0] * p[0] + m[0][1] * p[1] + m[0][2] * p[2] + m[0][3];
        double y = m[1][0] * p[0] + m[1][1] * p[1] + m[1][2] * p[2] + m[1][3];
        double z = m[2][0] * p[0] + m[2][1] * p[1] + m[2][2] * p[2] + m[2][3];
        double w = m[3][0] * p[0] + m[3][1] * p[1] + m[3][2] * p[2] + m[3][3];
        if (w != 1.0) { x /= w; y /= w; z /= w; }
        out[0] = x; out[1] = y; out[2] = z;
    }
    bool SwapsHandedness() const { // Transform::swapsHandedness: negative determinant of the upper 3x3
        double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                     m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        return det < 0;
    }
    Transform ToFloat() const { Transform t; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t.m_mat[r][c] = (float)m[r][c]; return t; }
};
inline Mat4d TranslateD(double x, double y, double z) { Mat4d t; t.m[0][3] = x; t.m[1][3] = y; t.m[2][3] = z; return t; }
inline Mat4d ScaleD(double x, double y, double z) { Mat4d t; t.m[0][0] = x; t.m[1][1] = y; t.m[2][2] = z; return t; }
inline Mat4d RotateD(double deg, double ax, double ay, double az) { // Rotate(theta, axis)
    double len = std::sqrt(ax * ax + ay * ay + az * az);
    if (len == 0) throw std::runtime_error("Rotate: zero axis");
    ax /= len; ay /= len; az /= len;
    double th = deg * 3.14159265358979323846 / 180.0, s = std::sin(th), c = std::cos(th);
    Mat4d t;
    t.m[0][0] = ax * ax + (1 - ax * ax) * c; t.m[0][1] = ax * ay * (1 - c) - az * s; t.m[0][2] = ax * az * (1 - c) + ay * s;
    t.m[1][0] = ax * ay * (1 - c) + az * s; t.m[1][1] = ay * ay + (1 - ay * ay) * c; t.m[1][2] = ay * az * (1 - c) - ax * s;
    t.m[2][0] = ax * az * (1 - c) - ay * s; t.m[2][1] = ay * az * (1 - c) + ax * s; t.m[2][2] = az * az + (1 - az * az) * c;
    return t;
}
inline Mat4d LookAtD(const double e[3], const double l[3], const double u[3]) { // world-to-camera, left-handed, +z forward
    double d[3] = {l[0] - e[0], l[1] - e[1], l[2] - e[2]};
    double dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (dl == 0) throw std::runtime_error("LookAt: eye and look-at point coincide");
    for (double& v : d) v /= dl;
    double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double un[3] = {u[0] / ul, u[1] / ul, u[2] / ul};
    double r[3] = {un[1] * d[2] - un[2] * d[1], un[2] * d[0] - un[0] * d[2], un[0] * d[1] - un[1] * d[0]};
    double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (rl == 0) throw std::runtime_error("LookAt: up vector and viewing direction are parallel");
    for (double& v : r) v /= rl;
    double nu[3] = {d[1] * r[2] - d[2] * r[1], d[2] * r[0] - d[0] * r[2], d[0] * r[1] - d[1] * r[0]};
    Mat4d c2w;
    for (int i = 0; i < 3; ++i) { c2w.m[i][0] = r[i]; c2w.m[i][1] = nu[i]; c2w.m[i][2] = d[i]; c2w.m[i][3] = e[i]; }
    return c2w.Inverse();
}

class LanczosSincFilter : public Filter { // Enigma/Filter.h:63-88, Enigma/Filter.cpp:18-21: the argument is already divided by the radius
public:
    LanczosSincFilter(const Vector2f& r, float tau) : Filter(r), m_tau(tau) {}
    float Evaluate(const Vector2f& p) const override { return Sinc1D(p.x * m_invRadius.x) * Sinc1D(p.y * m_invRadius.y); }
private:
    float Sinc1D(float x) const {
        x = fabsf(x);
        if (x < 1e-5) return 1.f;
        if (x > 1.) return 0.f;
        x *= 3.1415926535897932384f;
        float sinc = sinf(x) / x;
        float lanczos = sinf(x * m_tau) / (x * m_tau);
        return sinc * lanczos;
    }
    const float m_tau;
};

class MitchellFilter : public Filter { // Enigma/Filter.h:78-103
public:
    MitchellFilter(const Vector2f& r, float B, float C) : Filter(r), m_B(B), m_C(C) {}
    float Evaluate(const Vector2f& p) const override { return M1(p.x * m_invRadius.x) * M1(p.y * m_invRadius.y); }
private:
    float M1(float x) const {
        x = std::abs(2 * x);
        if (x > 1) return ((-m_B - 6 * m_C) * x * x * x + (6 * m_B + 30 * m_C) * x * x + (-12 * m_B - 48 * m_C) * x + (8 * m_B + 24 * m_C)) * (1.f / 6.f);
        return ((12 - 9 * m_B - 6 * m_C) * x * x * x + (-18 + 12 * m_B + 6 * m_C) * x * x + (6 - 2 * m_B)) * (1.f / 6.f);
    }
    const float m_B, m_C;
};

// ---------------------------------------------------------------- API state (Enigma/Api.h:96-200)
struct RenderOptions {
    std::string FilterName = "box", FilmName = "image", SamplerName = "halton", IntegratorName = "path", CameraName = "perspective";
    ParamSet FilterParams, FilmParams, SamplerParams, IntegratorParams, CameraParams;
    Mat4d CameraToWorld;
    bool haveCamera = false;
};

struct GraphicsState {
    int currentMaterial = -1; // index into the scene's material registry; -1 = the default matte, made on first use; -2 = none
    std::string currentInsideMedium, currentOutsideMedium;
    std::string areaLight;
    ParamSet areaLightParams;
    bool reverseOrientation = false;
    std::map<std::string, int> namedMaterials;
};

// What WorldEnd built: kept so that callers (and tests) can look at it after Render.
struct FrontEndResult {
    Scene scene;
    CameraPtr camera;
    SamplerPtr sampler;
    std::unique_ptr<GpuSamplerIntegrator> integrator;
    std::string integratorName, lightStrategy, filterName;
    int maxDepth = 5, directStrategy = 0;
    float rrThreshold = 1.f;
    BBox2i pixelBounds;
    std::vector<std::string> warnings;
};

class PbrtApi {
public:
    explicit PbrtApi(bool renderOnWorldEnd = true, int device = 0) : m_render(renderOnWorldEnd), m_device(device) {}

    // ---- transforms (Enigma/Api.cpp:507-586)
    void pbrtIdentity() { m_ctm = Mat4d(); }
    void pbrtTranslate(float dx, float dy, float dz) { m_ctm = m_ctm * TranslateD(dx, dy, dz); }
    void pbrtRotate(float angle, float ax, float ay, float az) { m_ctm = m_ctm * RotateD(angle, ax, ay, az); }
    void pbrtScale(float sx, float sy, float sz) { m_ctm = m_ctm * ScaleD(sx, sy, sz); }
    void pbrtLookAt(float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy, float uz) {
        double e[3] = {ex, ey, ez}, l[3] = {lx, ly, lz}, u[3] = {ux, uy, uz};
        m_ctm = m_ctm * LookAtD(e, l, u);
    }
    static Mat4d FromColumnMajor(const float tr[16]) { // the file lists matrices column by column
        Mat4d t;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t.m[r][c] = tr[4 * c + r];
        return t;
    }
    void pbrtConcatTransform(const float tr[16]) { m_ctm = m_ctm * FromColumnMajor(tr); }
    void pbrtTransform(const float tr[16]) { m_ctm = FromColumnMajor(tr); }
    void pbrtCoordinateSystem(const std::string& name) { m_namedCoordinateSystems[name] = m_ctm; }
    void pbrtCoordSysTransform(const std::string& name) {
        auto it = m_namedCoordinateSystems.find(name);
        if (it != m_namedCoordinateSystems.end()) m_ctm = it->second;
        else Warn("Couldn't find named coordinate system \"" + name + "\"");
    }

    // ---- option block (Enigma/Api.cpp:600-641)
    void pbrtPixelFilter(const std::string& name, const ParamSet& p) { VerifyOptions("PixelFilter"); m_opt.FilterName = name; m_opt.FilterParams = p; }
    void pbrtFilm(const std::string& type, const ParamSet& p) { VerifyOptions("Film"); m_opt.FilmName = type; m_opt.FilmParams = p; }
    void pbrtSampler(const std::string& name, const ParamSet& p) { VerifyOptions("Sampler"); m_opt.SamplerName = name; m_opt.SamplerParams = p; }
    void pbrtAccelerator(const std::string&, const ParamSet&) { VerifyOptions("Accelerator"); } // the device builds its own BVH
    void pbrtIntegrator(const std::string& name, const ParamSet& p) { VerifyOptions("Integrator"); m_opt.IntegratorName = name; m_opt.IntegratorParams = p; }
    void pbrtCamera(const std::string& name, const ParamSet& p) {
        VerifyOptions("Camera");
        m_opt.CameraName = name; m_opt.CameraParams = p;
        m_opt.CameraToWorld = m_ctm.Inverse();
        m_opt.haveCamera = true;
        m_namedCoordinateSystems["camera"] = m_opt.CameraToWorld;
    }

    // ---- world block (Enigma/Api.cpp:667-905)
    void pbrtWorldBegin() {
        VerifyOptions("WorldBegin");
        m_inWorld = true;
        m_ctm = Mat4d();
        m_namedCoordinateSystems["world"] = m_ctm;
    }
    void pbrtAttributeBegin() { VerifyWorld("AttributeBegin"); m_pushedGraphicsStates.push_back(m_gs); m_pushedTransforms.push_back(m_ctm); }
    void pbrtAttributeEnd() {
        VerifyWorld("AttributeEnd");
        if (m_pushedGraphicsStates.empty()) { Warn("Unmatched AttributeEnd encountered. Ignoring it."); return; }
        m_gs = m_pushedGraphicsStates.back(); m_pushedGraphicsStates.pop_back();
        m_ctm = m_pushedTransforms.back(); m_pushedTransforms.pop_back();
    }
    void pbrtTransformBegin() { VerifyWorld("TransformBegin"); m_pushedTransforms.push_back(m_ctm); }
    void pbrtTransformEnd() {
        VerifyWorld("TransformEnd");
        if (m_pushedTransforms.empty()) { Warn("Unmatched TransformEnd encountered. Ignoring it."); return; }
        m_ctm = m_pushedTransforms.back(); m_pushedTransforms.pop_back();
    }
    void pbrtReverseOrientation() { VerifyWorld("ReverseOrientation"); m_gs.reverseOrientation = !m_gs.reverseOrientation; }
    void pbrtMaterial(const std::string& name, const ParamSet& p) { VerifyWorld("Material"); m_gs.currentMaterial = MakeMaterial(name, p); }
    // MakeMedium (Enigma/Api.cpp:282-331) into the named-media table of the render options (:644-655)
    void pbrtMakeNamedMedium(const std::string& name, const ParamSet& p) {
        std::string type = p.FindOneString("type", "");
        if (type.empty()) throw std::runtime_error("No parameter string \"type\" found in MakeNamedMedium");
        if (type != "homogeneous" && type != "heterogeneous") throw std::runtime_error("Medium \"" + type + "\" unknown.");
        Spectrum sigA(.0011f, .0024f, .014f), sigS(2.55f, 3.21f, 3.77f);
        // GetMediumScatteringProperties (Enigma/MediumInterface.cpp:184-194): a measured medium by name; an unknown name keeps the defaults
        const std::string preset = p.FindOneString("preset", "");
        bool found = false;
        for (const MediumPreset& mp : kMediumPresets)
            if (preset == mp.name) {
                sigA = Spectrum(mp.sigma_a[0], mp.sigma_a[1], mp.sigma_a[2]);
                sigS = Spectrum(mp.sigma_prime_s[0], mp.sigma_prime_s[1], mp.sigma_prime_s[2]);
                found = true;
                break;
            }
        if (!preset.empty() && !found) Warn("Material preset \"" + preset + "\" not found.  Using defaults.");
        float scale = p.FindOneFloat("scale", 1.f);
        Spectrum sa = p.FindOneSpectrum("sigma_a", sigA), ss = p.FindOneSpectrum("sigma_s", sigS);
        EnigmaMedium m{};
        for (int c = 0; c < 3; ++c) { m.sigma_a[c] = sa[c] * scale; m.sigma_s[c] = ss[c] * scale; }
        m.g = p.FindOneFloat("g", 0.0f);
        if (type == "heterogeneous") {
            // GridDensityMedium (Enigma/Api.cpp:303-326, Enigma/Medium.cpp:16-64): the factory's checks are the reference's; the
            // medium itself is the reference's too -- its Tr and Sample return black whatever the grid holds -- so the grid is
            // read for those checks and goes no further
            const std::vector<float> data = p.FindFloat("density");
            const int nitems = (int)data.size();
            if (data.empty()) throw std::runtime_error("No \"density\" values provided for heterogeneous medium?");
            const int nx = p.FindOneInt("nx", 1), ny = p.FindOneInt("ny", 1), nz = p.FindOneInt("nz", 1);
            (void)p.FindOnePoint3f("p0", Vector3f(0.f, 0.f, 0.f));
            (void)p.FindOnePoint3f("p1", Vector3f(1.f, 1.f, 1.f));
            if (nitems != nx * ny * nz)
                throw std::runtime_error("GridDensityMedium has " + std::to_string(nitems) + " density values; expected nx*ny*nz = " + std::to_string(nx * ny * nz));
            if (!(m.sigma_a[0] + m.sigma_s[0] == m.sigma_a[1] + m.sigma_s[1] && m.sigma_a[0] + m.sigma_s[0] == m.sigma_a[2] + m.sigma_s[2]))
                Warn("GridDensityMedium requires a spectrally uniform attenuation coefficient!");
            m.kind = ENIGMA_MEDIUM_GRID_DENSITY;
        }
        m_namedMedia[name] = m_res->scene.AddMedium(m);
        NoteUnused("MakeNamedMedium", p);
    }
    void pbrtMediumInterface(const std::string& insideName, const std::string& outsideName) { // Enigma/Api.cpp:657-662
        m_gs.currentInsideMedium = insideName;
        m_gs.currentOutsideMedium = outsideName;
    }
    int LookupMedium(const std::string& name) const { // GraphicsState::CreateMediumInterface, Enigma/Api.cpp:1017-1036
        if (name.empty()) return -1;
        auto it = m_namedMedia.find(name);
        if (it == m_namedMedia.end()) throw std::runtime_error("Named medium \"" + name + "\" undefined.");
        return it->second;
    }
    void pbrtMakeNamedMaterial(const std::string& name, const ParamSet& p) {
        VerifyWorld("MakeNamedMaterial");
        std::string type = p.FindOneString("type", "");
        if (type.empty()) throw std::runtime_error("No parameter string \"type\" found in MakeNamedMaterial");
        m_gs.namedMaterials[name] = MakeMaterial(type, p);
    }
    void pbrtNamedMaterial(const std::string& name) {
        VerifyWorld("NamedMaterial");
        auto it = m_gs.namedMaterials.find(name);
        if (it == m_gs.namedMaterials.end()) throw std::runtime_error("NamedMaterial \"" + name + "\" unknown.");
        m_gs.currentMaterial = it->second;
    }
    // pbrtTexture (Enigma/Api.cpp:742-790) for spectrum textures of class "constant" and "imagemap"
    // (CreateImageSpectrumTexture, Enigma/Texture.cpp:412-453; uv mapping; PFM files only)
    void pbrtTexture(const std::string& name, const std::string& type, const std::string& texname, const ParamSet& p, const std::string& dir) {
        VerifyWorld("Texture");
        // TextureParams::GetFloatTexture / GetSpectrumTexture (Enigma/ParameterSet.cpp:686-743): "texture <n>" names a texture made
        // before, else the parameter's own value, else the default.  An operand here must be constant: the texture classes that
        // combine others (scale, mix) are folded into constants when they are made, with the arithmetic their Evaluate has.
        auto markTexture = [&](const char* n) { for (const auto& it : p.Items()) if (it.name == n && it.type == "texture") it.lookedUp = true; };
        auto floatOperand = [&](const char* n, float def) -> float {
            const std::string t = p.FindTexture(n);
            if (t.empty()) return p.FindOneFloat(n, def);
            markTexture(n);
            auto f = m_floatTextures.find(t);
            if (f == m_floatTextures.end()) throw std::runtime_error(std::string("Couldn't find float texture named \"") + t + "\" for parameter \"" + n + "\"");
            return f->second;
        };
        auto spectrumOperand = [&](const char* n, float def) -> Spectrum {
            const std::string t = p.FindTexture(n);
            if (t.empty()) return p.FindOneSpectrum(n, Spectrum(def));
            markTexture(n);
            auto f = m_spectrumTextures.find(t);
            if (f == m_spectrumTextures.end()) throw std::runtime_error(std::string("Couldn't find spectrum texture named \"") + t + "\" for parameter \"" + n + "\"");
            if (f->second.index) throw std::runtime_error("Texture \"" + name + "\": \"" + texname + "\" of an image texture is evaluated per hit, which is not supported; its operands must be constant");
            return f->second.value;
        };
        auto lerp = [](float a, float b, float t) { if (t < 0.f) return a; if (t > 1.f) return b; const float diff = (b - a) * t; return a + diff; }; // Lerp, Enigma/MathCommon.h:31-39
        if (type == "float") {
            if (m_floatTextures.count(name)) Warn("Texture \"" + name + "\" being redefined");
            float v;
            if (texname == "constant") v = p.FindOneFloat("value", 1.f);                                            // Enigma/Texture.cpp:158-161
            else if (texname == "scale") v = floatOperand("tex1", 1.f) * floatOperand("tex2", 1.f);                  // ScaleTexture::Evaluate, Enigma/Texture.h:147-149
            else if (texname == "mix") { const float a = floatOperand("tex1", 0.f), b = floatOperand("tex2", 1.f); v = lerp(a, b, floatOperand("amount", 0.5f)); } // MixTexture, :171-176
            else throw std::runtime_error("Texture \"" + name + "\": float class \"" + texname + "\" is not supported (constant, scale and mix of constants)");
            m_floatTextures[name] = v;
            NoteUnused("Texture", p);
            return;
        }
        if (type != "spectrum" && type != "color") throw std::runtime_error("Texture type \"" + type + "\" unknown.");
        if (m_spectrumTextures.count(name)) Warn("Texture \"" + name + "\" being redefined");
        NamedTexture nt;
        if (texname == "constant") {
            nt.value = p.FindOneSpectrum("value", Spectrum(1.f));
        } else if (texname == "scale") { // CreateScaleSpectrumTexture, Enigma/Texture.cpp:369-373
            const Spectrum a = spectrumOperand("tex1", 1.f), b = spectrumOperand("tex2", 1.f);
            nt.value = Spectrum(a[0] * b[0], a[1] * b[1], a[2] * b[2]);
        } else if (texname == "mix") { // CreateMixSpectrumTexture, Enigma/Texture.cpp:375-380
            const Spectrum a = spectrumOperand("tex1", 0.f), b = spectrumOperand("tex2", 1.f);
            const float t = floatOperand("amount", 0.5f);
            nt.value = Spectrum(lerp(a[0], b[0], t), lerp(a[1], b[1], t), lerp(a[2], b[2], t));
        } else if (texname == "imagemap") {
            if (p.FindOneString("mapping", "uv") != "uv") throw std::runtime_error("Texture \"" + name + "\": only the uv mapping is supported");
            const float su = p.FindOneFloat("uscale", 1.f), sv = p.FindOneFloat("vscale", 1.f), du = p.FindOneFloat("udelta", 0.f), dv = p.FindOneFloat("vdelta", 0.f);
            p.FindOneFloat("maxanisotropy", 8.f); // no effect: the EWA path never looks below the coarsest level
            const bool trilerp = p.FindOneBool("trilinear", false);
            const std::string wrap = p.FindOneString("wrap", "repeat");
            const uint32_t wrapMode = wrap == "black" ? ENIGMA_WRAP_BLACK : (wrap == "clamp" ? ENIGMA_WRAP_CLAMP : ENIGMA_WRAP_REPEAT);
            const float scale = p.FindOneFloat("scale", 1.f);
            std::string filename = p.FindOneString("filename", "");
            auto hasExt = [&](const char* e) { size_t n = strlen(e); return filename.size() >= n && filename.compare(filename.size() - n, n, e) == 0; };
            const bool gamma = p.FindOneBool("gamma", hasExt(".tga") || hasExt(".png")); // ends up as the exponent 1 or 0
            if (!hasExt(".pfm")) throw std::runtime_error("Texture \"" + name + "\": only PFM image files are read (\"" + filename + "\")");
            if (!filename.empty() && filename[0] != '/') filename = dir + filename;
            int w = 0, hgt = 0;
            std::vector<float> rgb = ReadPfm(filename, &w, &hgt);
            nt.index = m_res->scene.AddImageTexture(w, hgt, std::move(rgb), trilerp, gamma ? 1.f : 0.f, scale, wrapMode, su, sv, du, dv);
        } else throw std::runtime_error("Texture \"" + name + "\": class \"" + texname + "\" is not supported (constant, imagemap, scale and mix of constants)");
        m_spectrumTextures[name] = nt;
        NoteUnused("Texture", p);
    }
    // PFM: "PF", width height, scale (negative = little endian), rows bottom to top; returned top row first like ReadImage
    static std::vector<float> ReadPfm(const std::string& path, int* w, int* h) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("unable to read image \"" + path + "\"");
        std::string magic; float scale = 0.f;
        f >> magic >> *w >> *h >> scale;
        if (magic != "PF" || *w <= 0 || *h <= 0 || scale == 0.f) throw std::runtime_error("\"" + path + "\" is not a colour PFM file");
        f.get();
        std::vector<float> rows((size_t)*w * *h * 3);
        f.read(reinterpret_cast<char*>(rows.data()), (std::streamsize)(rows.size() * sizeof(float)));
        if (!f) throw std::runtime_error("\"" + path + "\" is truncated");
        if (scale > 0) for (float& v : rows) { unsigned char* b = reinterpret_cast<unsigned char*>(&v); std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
        std::vector<float> out(rows.size());
        const size_t rowLen = (size_t)*w * 3;
        for (int y = 0; y < *h; ++y) std::copy(rows.begin() + (size_t)(*h - 1 - y) * rowLen, rows.begin() + (size_t)(*h - y) * rowLen, out.begin() + (size_t)y * rowLen);
        return out;
    }
    void pbrtLightSource(const std::string& name, const ParamSet& p, const std::string& dir = "") {
        VerifyWorld("LightSource");
        if (name == "spot") { MakeSpotLight(p); NoteUnused("LightSource", p); return; }
        if (name == "distant") { MakeDistantLight(p); NoteUnused("LightSource", p); return; }
        if (name == "infinite" || name == "exinfinite") { // CreateInfiniteLight, Enigma/Lights.cpp:801-810
            std::string mapname = p.FindOneString("mapname", "");
            std::vector<float> map;
            int mw = 0, mh = 0;
            if (!mapname.empty()) { // InfiniteAreaLight's constructor reads it (Enigma/Lights.cpp:395-420); PFM is the format this front end reads
                if (mapname.size() < 4 || mapname.compare(mapname.size() - 4, 4, ".pfm") != 0)
                    throw std::runtime_error("LightSource \"infinite\": only PFM image files are read (\"" + mapname + "\")");
                if (mapname[0] != '/') mapname = dir + mapname;
                map = ReadPfm(mapname, &mw, &mh);
            }
            Spectrum L = p.FindOneSpectrum("L", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
            int nSamples = p.FindOneInt("samples", p.FindOneInt("nsamples", 1));
            m_res->scene.AddInfiniteLight(m_ctm.ToFloat(), m_ctm.Inverse().ToFloat(), Spectrum(L[0] * sc[0], L[1] * sc[1], L[2] * sc[2]), nSamples, std::move(map), mw, mh);
            NoteUnused("LightSource", p);
            return;
        }
        if (name == "goniometric") { // CreateGoniometricLight, Enigma/Lights.cpp:774-780: at lightToWorld(0,0,0), I x scale, the image named by "mapname"
            Spectrum I = p.FindOneSpectrum("I", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
            std::string mapname = p.FindOneString("mapname", "");
            std::vector<float> map;
            int mw = 0, mh = 0;
            if (!mapname.empty()) {
                if (mapname.size() < 4 || mapname.compare(mapname.size() - 4, 4, ".pfm") != 0)
                    throw std::runtime_error("LightSource \"goniometric\": only PFM image files are read (\"" + mapname + "\")");
                if (mapname[0] != '/') mapname = dir + mapname;
                map = ReadPfm(mapname, &mw, &mh);
            }
            double zero[3] = {0, 0, 0}, pos[3];
            m_ctm.Point(zero, pos);
            m_res->scene.AddGoniometricLight(Vector3f((float)pos[0], (float)pos[1], (float)pos[2]), Spectrum(I[0] * sc[0], I[1] * sc[1], I[2] * sc[2]), std::move(map), mw, mh);
            NoteUnused("LightSource", p);
            return;
        }
        if (name == "projection") { // CreateProjectionLight, Enigma/Lights.cpp:782-789: the light sits at lightToWorld(0,0,0); there is no "from"
            Spectrum I = p.FindOneSpectrum("I", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
            float fov = p.FindOneFloat("fov", 45.f);
            if (!p.FindOneString("mapname", "").empty())
                Warn("LightSource \"projection\": the reference reads \"mapname\" and drops the image before it looks at it (Enigma/Lights.cpp:218-222); the light shines like a point light");
            double zero[3] = {0, 0, 0}, pos[3];
            m_ctm.Point(zero, pos);
            m_res->scene.AddProjectionLight(Vector3f((float)pos[0], (float)pos[1], (float)pos[2]), Spectrum(I[0] * sc[0], I[1] * sc[1], I[2] * sc[2]), fov);
            NoteUnused("LightSource", p);
            return;
        }
        if (name != "point") throw std::runtime_error("LightSource \"" + name + "\" is not supported (point, spot, distant, projection, goniometric, infinite, and diffuse area lights)");
        Spectrum I = p.FindOneSpectrum("I", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
        Vector3f P = p.FindOnePoint3f("from", Vector3f(0, 0, 0));
        // l2w = Translate(P) * lightToWorld, position = l2w(0,0,0) (Enigma/Lights.cpp:742-750,89-95)
        Mat4d l2w = TranslateD(P.x, P.y, P.z) * m_ctm;
        double zero[3] = {0, 0, 0}, pos[3];
        l2w.Point(zero, pos);
        Spectrum Isc(I[0] * sc[0], I[1] * sc[1], I[2] * sc[2]);
        m_res->scene.AddPointLight(Vector3f((float)pos[0], (float)pos[1], (float)pos[2]), Isc);
        NoteUnused("LightSource", p);
    }
    // CreateSpotLight (Enigma/Lights.cpp:752-771): light2world = CTM * Translate(from) * inverse(dirToZ), where the rows
    // of dirToZ are (du, dv, dir) from CoordinateSystem(Normalize(to - from)) in float, as the reference computes them.
    void MakeSpotLight(const ParamSet& p) {
        Spectrum I = p.FindOneSpectrum("I", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
        float coneangle = p.FindOneFloat("coneangle", 30.f), conedelta = p.FindOneFloat("conedeltaangle", 5.f);
        Vector3f from = p.FindOnePoint3f("from", Vector3f(0, 0, 0)), to = p.FindOnePoint3f("to", Vector3f(0, 0, 1));
        float dx = to.x - from.x, dy = to.y - from.y, dz = to.z - from.z;
        float len = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (len == 0) throw std::runtime_error("LightSource \"spot\": from and to coincide");
        float dir[3] = {dx / len, dy / len, dz / len}, du[3], dv[3];
        if (std::fabs(dir[0]) > std::fabs(dir[1])) { // CoordinateSystem, Enigma/MathCommon.h:371-382
            float inv = 1.f / std::sqrt(dir[0] * dir[0] + dir[2] * dir[2]);
            du[0] = -dir[2] * inv; du[1] = 0.f; du[2] = dir[0] * inv;
        } else {
            float inv = 1.f / std::sqrt(dir[1] * dir[1] + dir[2] * dir[2]);
            du[0] = 0.f; du[1] = dir[2] * inv; du[2] = -dir[1] * inv;
        }
        dv[0] = dir[1] * du[2] - dir[2] * du[1]; dv[1] = dir[2] * du[0] - dir[0] * du[2]; dv[2] = dir[0] * du[1] - dir[1] * du[0];
        Mat4d dirToZ;
        for (int c = 0; c < 3; ++c) { dirToZ.m[0][c] = du[c]; dirToZ.m[1][c] = dv[c]; dirToZ.m[2][c] = dir[c]; }
        Mat4d l2w = m_ctm * TranslateD(from.x, from.y, from.z) * dirToZ.Inverse();
        Spectrum Isc(I[0] * sc[0], I[1] * sc[1], I[2] * sc[2]);
        m_res->scene.AddSpotLight(l2w.ToFloat(), l2w.Inverse().ToFloat(), Isc, coneangle, coneangle - conedelta);
    }
    // CreateDistantLight (Enigma/Lights.cpp:791-799): dir = from - to, carried through the CTM without normalisation.
    void MakeDistantLight(const ParamSet& p) {
        Spectrum L = p.FindOneSpectrum("L", Spectrum(1.f)), sc = p.FindOneSpectrum("scale", Spectrum(1.f));
        Vector3f from = p.FindOnePoint3f("from", Vector3f(0, 0, 0)), to = p.FindOnePoint3f("to", Vector3f(0, 0, 1));
        Spectrum Lsc(L[0] * sc[0], L[1] * sc[1], L[2] * sc[2]);
        m_res->scene.AddDistantLight(m_ctm.ToFloat(), Lsc, Vector3f(from.x - to.x, from.y - to.y, from.z - to.z));
    }
    void pbrtAreaLightSource(const std::string& name, const ParamSet& p) { VerifyWorld("AreaLightSource"); m_gs.areaLight = name; m_gs.areaLightParams = p; }
    void pbrtShape(const std::string& name, const ParamSet& p, const std::string& dir = "") {
        VerifyWorld("Shape");
        if (name == "sphere") { MakeSphere(p); return; }
        if (name == "plymesh") { MakePlyMesh(p, dir); return; }
        // MakeShapes (Enigma/Api.cpp:80-128): the factories of these three return no shape at all (Enigma/Curve.cpp:10-14,
        // Enigma/SubDiv.cpp:6-9, Enigma/Nurbs.cpp:10-14), and a name the reference does not know is a warning there
        if (name == "curve" || name == "loopsubdiv" || name == "nurbs") { Warn("Shape \"" + name + "\": the reference's factory returns no shape; none made"); return; }
        if (name == "cylinder" || name == "disk") throw std::runtime_error("Shape \"" + name + "\" is not supported (trianglemesh, plymesh, sphere)");
        if (name != "trianglemesh") { Warn("Shape \"" + name + "\" unknown."); return; }
        std::vector<int> vi = p.FindInt("indices");
        std::vector<double> P = p.FindPoint3("P");
        if (P.empty() || P.size() % 3) throw std::runtime_error("trianglemesh: \"point P\" missing or not a multiple of 3 values");
        int nVerts = (int)P.size() / 3;
        if (vi.empty()) {
            if (nVerts == 3) vi = {0, 1, 2};
            else throw std::runtime_error("Vertex indices \"indices\" not provided with triangle mesh shape");
        }
        if (vi.size() % 3) throw std::runtime_error("trianglemesh: number of indices is not a multiple of 3");
        for (int v : vi) if (v < 0 || v >= nVerts) throw std::runtime_error("trianglemesh has out-of-bounds vertex index");
        for (const auto& it : p.Items()) {
            if (it.name == "S") throw std::runtime_error("trianglemesh: shading tangents (\"S\") are not supported");
            if (it.name == "alpha" || it.name == "shadowalpha") throw std::runtime_error("trianglemesh: alpha textures are not supported");
        }
        // uv / st as point2 or float pairs (Enigma/TriangleMesh.cpp:693-722); normal N (:737-741)
        std::vector<double> uv = p.FindPoint2("uv");
        if (uv.empty()) uv = p.FindPoint2("st");
        if (uv.empty()) { std::vector<float> f = p.FindFloat("uv"); if (f.empty()) f = p.FindFloat("st"); uv.assign(f.begin(), f.end()); }
        if (!uv.empty() && (int)uv.size() / 2 < nVerts) { Warn("Not enough of \"uv\"s for triangle mesh.  Discarding."); uv.clear(); }
        std::vector<double> N = p.FindNormal3("N");
        if (!N.empty() && (int)N.size() / 3 != nVerts) { Warn("Number of \"N\"s for triangle mesh must match \"P\"s"); N.clear(); }
        AddMesh(vi, P, uv, N);
        NoteUnused("Shape", p);
    }
    // CreatePLYMesh (Enigma/TriangleMesh.cpp:791-931): the file's vertices, normals, uvs and faces into CreateTriangleMesh
    void MakePlyMesh(const ParamSet& p, const std::string& dir) {
        std::string filename = p.FindOneString("filename", "");
        if (!filename.empty() && filename[0] != '/') filename = dir + filename; // ParamSet::FindOneFilename: relative to the scene file
        for (const auto& it : p.Items())
            if (it.name == "alpha" || it.name == "shadowalpha") throw std::runtime_error("plymesh: alpha textures are not supported");
        PlyMesh ply = PlyReader::Read(filename);
        for (const std::string& w : ply.warnings) Warn(w);
        if (ply.indices.empty()) { Warn("plymesh: \"" + filename + "\" holds no triangle or quad under the face property \"vertex_indices\"; no shape made"); NoteUnused("Shape", p); return; }
        AddMesh(ply.indices, std::vector<double>(ply.p.begin(), ply.p.end()), std::vector<double>(ply.uv.begin(), ply.uv.end()),
                std::vector<double>(ply.n.begin(), ply.n.end()));
        NoteUnused("Shape", p);
    }
    // CreateTriangleMesh + the primitives MakeShapes wraps its triangles in (Enigma/TriangleMesh.cpp:662-682, Enigma/Api.cpp:843-874)
    void AddMesh(const std::vector<int>& vi, const std::vector<double>& P, const std::vector<double>& uv, const std::vector<double>& N) {
        const int nVerts = (int)P.size() / 3;
        TriangleMesh m;
        m.m_nTriangles = (int)vi.size() / 3; m.m_nVertices = nVerts;
        m.m_vertexIndices = vi;
        m.m_p.reserve(nVerts);
        for (int i = 0; i < nVerts; ++i) { // TriangleMesh ctor: vertices to world space (Enigma/TriangleMesh.cpp:23-58)
            double w[3];
            m_ctm.Point(&P[3 * (size_t)i], w);
            m.m_p.push_back(Vector3f((float)w[0], (float)w[1], (float)w[2]));
        }
        if (m_gs.currentMaterial == -1 && m_defaultMaterial < 0) m_defaultMaterial = MakeMaterial("matte", ParamSet()); // GraphicsState ctor, Enigma/Api.cpp:992-999
        m.materialIndex = m_gs.currentMaterial == -1 ? m_defaultMaterial : (m_gs.currentMaterial == -2 ? -1 : m_gs.currentMaterial);
        m.mediumInside = LookupMedium(m_gs.currentInsideMedium); m.mediumOutside = LookupMedium(m_gs.currentOutsideMedium);
        m.reverseOrientation = m_gs.reverseOrientation; // Enigma/TriangleMesh.cpp:287-289,369 use both flags
        m.transformSwapsHandedness = m_ctm.SwapsHandedness();
        if (!N.empty()) { // ObjectToWorld.transformNormal: the inverse transpose (Enigma/Transform.h:54-59, Enigma/TriangleMesh.cpp:44-48)
            Mat4d inv = m_ctm.Inverse();
            for (int i = 0; i < nVerts; ++i) {
                const double* n = &N[3 * (size_t)i];
                m.m_n.push_back(Vector3f((float)(inv.m[0][0] * n[0] + inv.m[1][0] * n[1] + inv.m[2][0] * n[2]), (float)(inv.m[0][1] * n[0] + inv.m[1][1] * n[1] + inv.m[2][1] * n[2]),
                                         (float)(inv.m[0][2] * n[0] + inv.m[1][2] * n[1] + inv.m[2][2] * n[2])));
            }
        }
        for (int i = 0; i < nVerts && !uv.empty(); ++i) m.m_uv.push_back(Vector2f((float)uv[2 * (size_t)i], (float)uv[2 * (size_t)i + 1]));
        if (!m_gs.areaLight.empty()) {
            if (m_gs.areaLight != "diffuse" && m_gs.areaLight != "area") throw std::runtime_error("AreaLightSource \"" + m_gs.areaLight + "\" is not supported (diffuse)");
            const ParamSet& ap = m_gs.areaLightParams; // CreateDiffuseAreaLight, Enigma/Lights.cpp:812-822
            Spectrum L = ap.FindOneSpectrum("L", Spectrum(1.f)), sc = ap.FindOneSpectrum("scale", Spectrum(1.f));
            m.emissive = true;
            m.Le = Spectrum(L[0] * sc[0], L[1] * sc[1], L[2] * sc[2]);
            m.nSamples = ap.FindOneInt("samples", ap.FindOneInt("nsamples", 1));
            m.twoSided = ap.FindOneBool("twosided", false);
        }
        m_res->scene.AddTriangleMesh(m);
    }
    // CreateSphereShape (Enigma/ShapeImp.cpp:723-732) + the primitive MakeShapes wraps it in (Enigma/Api.cpp:843-874)
    void MakeSphere(const ParamSet& p) {
        const float radius = p.FindOneFloat("radius", 1.f);
        const float zmin = p.FindOneFloat("zmin", -radius), zmax = p.FindOneFloat("zmax", radius), phimax = p.FindOneFloat("phimax", 360.f);
        if (m_gs.currentMaterial == -1 && m_defaultMaterial < 0) m_defaultMaterial = MakeMaterial("matte", ParamSet());
        const int material = m_gs.currentMaterial == -1 ? m_defaultMaterial : (m_gs.currentMaterial == -2 ? -1 : m_gs.currentMaterial); // -1: Material "none"
        Transform o2w = m_ctm.ToFloat(), w2o = m_ctm.Inverse().ToFloat();
        Spectrum Le(0.f); bool emissive = false, twoSided = false; int nSamples = 1;
        if (!m_gs.areaLight.empty()) {
            if (m_gs.areaLight != "diffuse" && m_gs.areaLight != "area") throw std::runtime_error("AreaLightSource \"" + m_gs.areaLight + "\" is not supported (diffuse)");
            const ParamSet& ap = m_gs.areaLightParams; // CreateDiffuseAreaLight, Enigma/Lights.cpp:812-822
            Spectrum L = ap.FindOneSpectrum("L", Spectrum(1.f)), sc = ap.FindOneSpectrum("scale", Spectrum(1.f));
            emissive = true;
            Le = Spectrum(L[0] * sc[0], L[1] * sc[1], L[2] * sc[2]);
            nSamples = ap.FindOneInt("samples", ap.FindOneInt("nsamples", 1));
            twoSided = ap.FindOneBool("twosided", false);
        }
        m_res->scene.AddSphere(o2w, w2o, m_gs.reverseOrientation, m_ctm.SwapsHandedness(), radius, zmin, zmax, phimax, material,
                               LookupMedium(m_gs.currentInsideMedium), LookupMedium(m_gs.currentOutsideMedium), emissive ? &Le : nullptr, twoSided, nSamples);
        NoteUnused("Shape", p);
    }
    void pbrtObjectBegin(const std::string&) { throw std::runtime_error("ObjectBegin: object instancing is not supported"); }

    // WorldEnd: MakeIntegrator (camera, film, filter, sampler) + MakeScene + Render (Enigma/Api.cpp:1170-1254, 1103-1135)
    void pbrtWorldEnd() {
        VerifyWorld("WorldEnd");
        while (!m_pushedGraphicsStates.empty()) { Warn("Missing end to AttributeBegin"); m_pushedGraphicsStates.pop_back(); m_pushedTransforms.pop_back(); }
        FrontEndResult& R = *m_res;
        // filter (Enigma/Filter.cpp:38-75)
        const ParamSet& fp = m_opt.FilterParams;
        std::unique_ptr<Filter> filter;
        if (m_opt.FilterName == "box") filter.reset(new BoxFilter(Vector2f(fp.FindOneFloat("xwidth", 0.5f), fp.FindOneFloat("ywidth", 0.5f))));
        else if (m_opt.FilterName == "gaussian") {
            float xw = fp.FindOneFloat("xwidth", 2.f), yw = fp.FindOneFloat("ywidth", 2.f);
            filter.reset(new GaussianFilter(Vector2f(xw, yw), fp.FindOneFloat("alpha", 2.f)));
        } else if (m_opt.FilterName == "mitchell") {
            float xw = fp.FindOneFloat("xwidth", 2.f), yw = fp.FindOneFloat("ywidth", 2.f);
            float B = fp.FindOneFloat("B", 1.f / 3.f), C = fp.FindOneFloat("C", 1.f / 3.f);
            filter.reset(new MitchellFilter(Vector2f(xw, yw), B, C));
        } else if (m_opt.FilterName == "sinc") { // CreateSincFilter, Enigma/Filter.cpp:62-68
            float xw = fp.FindOneFloat("xwidth", 4.f), yw = fp.FindOneFloat("ywidth", 4.f);
            filter.reset(new LanczosSincFilter(Vector2f(xw, yw), fp.FindOneFloat("tau", 3.f)));
        } else if (m_opt.FilterName == "triangle") filter.reset(new TriangleFilter(Vector2f(fp.FindOneFloat("xwidth", 2.f), fp.FindOneFloat("ywidth", 2.f))));
        else throw std::runtime_error("Filter \"" + m_opt.FilterName + "\" is not supported (box, gaussian, mitchell, sinc, triangle)");
        R.filterName = m_opt.FilterName;
        // film (Enigma/Film.cpp:330-372)
        if (m_opt.FilmName != "image") throw std::runtime_error("Film \"" + m_opt.FilmName + "\" unknown.");
        const ParamSet& flp = m_opt.FilmParams;
        std::string filename = m_imageFile.empty() ? flp.FindOneString("filename", "") : m_imageFile;
        int xres = flp.FindOneInt("xresolution", 1280), yres = flp.FindOneInt("yresolution", 720);
        BBox2f crop;
        std::vector<float> cr = flp.FindFloat("cropwindow");
        if (cr.size() == 4) {
            auto c01 = [](float v) { return std::min(1.f, std::max(0.f, v)); };
            crop.m_min.x = c01(std::min(cr[0], cr[1])); crop.m_max.x = c01(std::max(cr[0], cr[1]));
            crop.m_min.y = c01(std::min(cr[2], cr[3])); crop.m_max.y = c01(std::max(cr[2], cr[3]));
        } else if (!cr.empty()) throw std::runtime_error("\"cropwindow\" expects 4 values");
        float scale = flp.FindOneFloat("scale", 1.f), diagonal = flp.FindOneFloat("diagonal", 35.f);
        float maxLum = flp.FindOneFloat("maxsampleluminance", InfinityF32);
        std::unique_ptr<Film> film(new Film(Vector2i(xres, yres), crop, std::move(filter), diagonal, filename, scale, maxLum));
        Film* filmPtr = film.get();
        // camera (Enigma/Camera.cpp:385-432, 80-97)
        if (m_opt.CameraName != "perspective" && m_opt.CameraName != "environment")
            throw std::runtime_error("Camera \"" + m_opt.CameraName + "\" is not supported (perspective, environment)");
        const ParamSet& cp = m_opt.CameraParams;
        float lensradius = cp.FindOneFloat("lensradius", 0.f), focaldistance = cp.FindOneFloat("focaldistance", 1e30f);
        float frame = cp.FindOneFloat("frameaspectratio", float(xres) / float(yres));
        float screen[4];
        if (frame > 1.f) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.f; screen[3] = 1.f; }
        else { screen[0] = -1.f; screen[1] = 1.f; screen[2] = -1.f / frame; screen[3] = 1.f / frame; }
        std::vector<float> sw = cp.FindFloat("screenwindow");
        if (sw.size() == 4) for (int i = 0; i < 4; ++i) screen[i] = sw[i];
        float fov = cp.FindOneFloat("fov", 90.f), halffov = cp.FindOneFloat("halffov", -1.f);
        if (halffov > 0.f) fov = 2.f * halffov;
        const double n = 1e-2, f = 1000.0;
        Mat4d persp;
        persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n); persp.m[3][2] = 1; persp.m[3][3] = 0;
        double invTan = 1.0 / std::tan((double)fov * 3.14159265358979323846 / 180.0 / 2);
        Mat4d cameraToScreen = ScaleD(invTan, invTan, 1) * persp;
        Mat4d screenToRaster = ScaleD(xres, yres, 1) * ScaleD(1.0 / ((double)screen[1] - screen[0]), 1.0 / ((double)screen[2] - screen[3]), 1) *
                               TranslateD(-(double)screen[0], -(double)screen[3], 0);
        Mat4d rasterToCamera = cameraToScreen.Inverse() * screenToRaster.Inverse();
        if (m_opt.CameraName == "environment") // CreateEnvironmentCamera (Enigma/Camera.cpp:476-510): lens and screen window are read and not used
            R.camera.reset(new EnvironmentCamera(m_opt.CameraToWorld.ToFloat(), cp.FindOneFloat("shutteropen", 0.f), cp.FindOneFloat("shutterclose", 1.f), std::move(film)));
        else
            R.camera.reset(new PerspectiveCamera(m_opt.CameraToWorld.ToFloat(), rasterToCamera.ToFloat(), cp.FindOneFloat("shutteropen", 0.f),
                                                 cp.FindOneFloat("shutterclose", 1.f), lensradius, focaldistance, std::move(film)));
        // sampler (Enigma/Sampler.cpp:554-559)
        if (m_opt.SamplerName == "sobol") // CreateSobolSampler, Enigma/Sampler.cpp:554-559
            R.sampler.reset(new SobolSampler(m_opt.SamplerParams.FindOneInt("pixelsamples", 16), filmPtr->GetSampleBounds()));
        else if (m_opt.SamplerName == "halton") // CreateHaltonSampler, Enigma/Sampler.cpp:521-528; the name RenderOptions starts with (Enigma/Api.h:82)
            R.sampler.reset(new HaltonSampler(m_opt.SamplerParams.FindOneInt("pixelsamples", 16), filmPtr->GetSampleBounds(),
                                              m_opt.SamplerParams.FindOneBool("samplepixelcenter", false)));
        else
            throw std::runtime_error("Sampler \"" + m_opt.SamplerName + "\" is not supported: the GPU integrators reproduce the sobol and halton samplers "
                                     "(the pixel samplers draw from a generator that runs on through the pixels of a tile)");
        // integrator (Enigma/PathIntegrator.cpp:182-201, Enigma/DirectLightingIntegrator.cpp:89-114)
        const ParamSet& ip = m_opt.IntegratorParams;
        R.integratorName = m_opt.IntegratorName;
        R.maxDepth = ip.FindOneInt("maxdepth", 5);
        BBox2i pixelBounds = filmPtr->GetSampleBounds();
        std::vector<int> pb = ip.FindInt("pixelbounds");
        if (pb.size() == 4) { // Intersection(sampleBounds, {{pb[0], pb[2]}, {pb[1], pb[3]}})
            pixelBounds = BBox2i(Vector2i(std::max(pixelBounds.m_min.x, pb[0]), std::max(pixelBounds.m_min.y, pb[2])),
                                 Vector2i(std::min(pixelBounds.m_max.x, pb[1]), std::min(pixelBounds.m_max.y, pb[3])));
        }
        R.pixelBounds = pixelBounds;
        if (m_opt.IntegratorName == "path") {
            R.rrThreshold = ip.FindOneFloat("rrthreshold", 1.f);
            R.lightStrategy = ip.FindOneString("lightsamplestrategy", "spatial");
            R.integrator.reset(new GpuPathIntegrator(R.maxDepth, R.camera, R.sampler, pixelBounds, R.rrThreshold, R.lightStrategy, m_device));
        } else if (m_opt.IntegratorName == "directlighting") {
            std::string st = ip.FindOneString("strategy", "all");
            R.directStrategy = st == "one" ? 1 : 0;
            R.integrator.reset(new GpuDirectLightingIntegrator(st == "one" ? GpuDirectLightingIntegrator::eLightStrategy::UniformSampleOne
                                                                           : GpuDirectLightingIntegrator::eLightStrategy::UniformSampleAll,
                                                               R.maxDepth, R.camera, R.sampler, pixelBounds, m_device));
        } else if (m_opt.IntegratorName == "volpath") { // CreateVolPathIntegrator, Enigma/VolPathIntegrator.cpp:179-200
            R.rrThreshold = ip.FindOneFloat("rrthreshold", 1.f);
            R.lightStrategy = ip.FindOneString("lightsamplestrategy", "spatial");
            R.integrator.reset(new GpuVolPathIntegrator(R.maxDepth, R.camera, R.sampler, pixelBounds, R.rrThreshold, R.lightStrategy, m_device));
        } else throw std::runtime_error("Integrator \"" + m_opt.IntegratorName + "\" is not supported (path, directlighting, volpath)");
        for (const char* what : {"Film", "Camera", "Sampler", "Integrator", "PixelFilter"}) {
            const ParamSet& ps = std::string(what) == "Film" ? flp : std::string(what) == "Camera" ? cp : std::string(what) == "Sampler" ? m_opt.SamplerParams
                                 : std::string(what) == "Integrator" ? ip : fp;
            NoteUnused(what, ps);
        }
        if (R.scene.getNumLights() == 0) Warn("No light sources defined in scene; rendering a black image.");
        m_inWorld = false;
        if (m_world > 1) R.integrator->SetRanks(m_rank, m_world, m_commId);
        if (m_render) R.integrator->Render(R.scene);
    }

    // ---- parsing (Enigma/Parser.cpp)
    void pbrtParseFile(const std::string& filename) {
        std::ifstream f(filename);
        if (!f) throw std::runtime_error("Couldn't open scene file \"" + filename + "\"");
        std::stringstream ss;
        ss << f.rdbuf();
        std::string dir;
        size_t slash = filename.find_last_of('/');
        if (slash != std::string::npos) dir = filename.substr(0, slash + 1);
        Parse(ss.str(), dir);
    }
    void pbrtParseString(const std::string& s) { Parse(s, ""); }

    void SetImageFile(const std::string& f) { m_imageFile = f; } // Options::imageFile overrides the Film's filename
    // one process per GPU: see GpuSamplerIntegrator::SetRanks (the device index is the constructor's)
    void SetRanks(int rank, int world, const void* commId) { m_rank = rank; m_world = world; std::memcpy(m_commId, commId, ENIGMA_COMM_ID_BYTES); }
    FrontEndResult& Result() { return *m_res; }
    std::unique_ptr<FrontEndResult> TakeResult() { return std::move(m_res); }

private:
    // ---- Create*Material (Enigma/Material.cpp:594-646)
    static void texGuard(const ParamSet& p, const char* n) {
        if (!p.FindTexture(n).empty()) throw std::runtime_error(std::string("material parameter \"") + n + "\" refers to a texture; only constant values are supported");
    }
    // Float parameters bound to a float texture ("texture roughness" "name") take that texture's value: every float texture
    // this front end makes is a constant (GetFloatTexture, Enigma/ParameterSet.cpp:716-743).
    ParamSet ResolveFloatTextures(const ParamSet& in) const {
        ParamSet out;
        for (const ParamSetItem& it : in.Items()) {
            static const char* floats[] = {"sigma", "roughness", "uroughness", "vroughness", "index", "eta"};
            bool isFloatParam = false;
            for (const char* n : floats) isFloatParam = isFloatParam || it.name == n;
            auto f = (it.type == "texture" && isFloatParam && !it.strs.empty()) ? m_floatTextures.find(it.strs[0]) : m_floatTextures.end();
            if (f == m_floatTextures.end()) { out.Add(it); continue; }
            ParamSetItem v;
            v.type = "float"; v.name = it.name; v.nums.push_back((double)f->second);
            out.Add(v);
        }
        return out;
    }
    int MakeMaterial(const std::string& name, const ParamSet& given) {
        const ParamSet p = ResolveFloatTextures(given);
        EnigmaMaterial m{};
        for (const char* n : {"sigma", "roughness", "uroughness", "vroughness", "index", "bumpmap"}) texGuard(p, n);
        if (p.Has("bumpmap")) throw std::runtime_error("bump mapping is not supported");
        auto put = [](float* dst, const Spectrum& s) { dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2]; };
        // TextureParams::GetSpectrumTexture (Enigma/ParameterSet.cpp:686-743): a named texture if "texture <n>" is given, else a constant
        auto bind = [&](const char* n, float* dst, uint32_t* tex) {
            const std::string t = p.FindTexture(n);
            if (t.empty()) return;
            for (const auto& it : p.Items()) if (it.name == n && it.type == "texture") it.lookedUp = true;
            auto f = m_spectrumTextures.find(t);
            if (f == m_spectrumTextures.end()) throw std::runtime_error(std::string("Couldn't find spectrum texture named \"") + t + "\" for parameter \"" + n + "\"");
            if (f->second.index) *tex = f->second.index;
            else put(dst, f->second.value);
        };
        // parameters that have no texture slot in the ABI (reflect, transmit, opacity, amount, the uber's eta)
        auto constantOnly = [&](const char* n) {
            if (!p.FindTexture(n).empty()) throw std::runtime_error("Material \"" + name + "\": parameter \"" + std::string(n) + "\" refers to a texture; this material takes constant values only");
        };
        m.remap_roughness = p.FindOneBool("remaproughness", true) ? 1u : 0u;
        if (name == "" || name == "none") return -2; // MakeMaterial returns no material (Enigma/Api.cpp:135): a medium boundary
        if (name == "matte") {
            m.type = ENIGMA_MAT_MATTE;
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(0.5f))); bind("Kd", m.kd, &m.tex_kd);
            m.sigma = p.FindOneFloat("sigma", 0.f);
        } else if (name == "metal") {
            m.type = ENIGMA_MAT_METAL;
            // the reference's defaults are the sampled copper spectra (Material.cpp:604-608) as RGB
            float cuEta[3], cuK[3];
            if (enigma_default_metal_spectra(cuEta, cuK) != ENIGMA_OK) throw std::runtime_error("metal: default spectra unavailable");
            put(m.eta, p.FindOneSpectrum("eta", Spectrum(cuEta[0], cuEta[1], cuEta[2]))); bind("eta", m.eta, &m.tex_eta);
            put(m.k, p.FindOneSpectrum("k", Spectrum(cuK[0], cuK[1], cuK[2]))); bind("k", m.k, &m.tex_k);
            float rough = p.FindOneFloat("roughness", .5f);
            m.uroughness = p.FindOneFloat("uroughness", rough); // MetalMaterial: u/v fall back to roughness (Material.cpp:121-126)
            m.vroughness = p.FindOneFloat("vroughness", rough);
        } else if (name == "glass") {
            m.type = ENIGMA_MAT_GLASS;
            put(m.kr, p.FindOneSpectrum("Kr", Spectrum(1.f))); bind("Kr", m.kr, &m.tex_kr);
            put(m.kt, p.FindOneSpectrum("Kt", Spectrum(1.f))); bind("Kt", m.kt, &m.tex_kt);
            m.index = p.Has("eta") ? p.FindOneFloat("eta", 1.5f) : p.FindOneFloat("index", 1.5f);
            m.uroughness = p.FindOneFloat("uroughness", 0.f);
            m.vroughness = p.FindOneFloat("vroughness", 0.f);
        } else if (name == "plastic") {
            m.type = ENIGMA_MAT_PLASTIC;
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(0.25f))); bind("Kd", m.kd, &m.tex_kd);
            put(m.ks, p.FindOneSpectrum("Ks", Spectrum(0.25f))); bind("Ks", m.ks, &m.tex_ks);
            m.uroughness = m.vroughness = p.FindOneFloat("roughness", .1f);
        } else if (name == "mirror") { // CreateMirrorMaterial, Enigma/Material.cpp:648-653
            m.type = ENIGMA_MAT_MIRROR;
            put(m.kr, p.FindOneSpectrum("Kr", Spectrum(0.9f))); bind("Kr", m.kr, &m.tex_kr);
        } else if (name == "translucent") { // CreateTranslucentMaterial, Enigma/Material.cpp:738-748
            m.type = ENIGMA_MAT_TRANSLUCENT;
            for (const char* n : {"reflect", "transmit"}) constantOnly(n);
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(0.25f))); bind("Kd", m.kd, &m.tex_kd);
            put(m.ks, p.FindOneSpectrum("Ks", Spectrum(0.25f))); bind("Ks", m.ks, &m.tex_ks);
            put(m.reflect, p.FindOneSpectrum("reflect", Spectrum(0.5f)));
            put(m.transmit, p.FindOneSpectrum("transmit", Spectrum(0.5f)));
            m.uroughness = m.vroughness = p.FindOneFloat("roughness", .1f);
        } else if (name == "substrate") { // CreateSubstrateMaterial, Enigma/Material.cpp:673-682
            m.type = ENIGMA_MAT_SUBSTRATE;
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(.5f))); bind("Kd", m.kd, &m.tex_kd);
            put(m.ks, p.FindOneSpectrum("Ks", Spectrum(.5f))); bind("Ks", m.ks, &m.tex_ks);
            m.uroughness = p.FindOneFloat("uroughness", .1f);
            m.vroughness = p.FindOneFloat("vroughness", .1f);
        } else if (name == "uber") { // CreateUberMaterial, Enigma/Material.cpp:720-736
            m.type = ENIGMA_MAT_UBER;
            for (const char* n : {"opacity", "eta"}) constantOnly(n);
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(0.25f))); bind("Kd", m.kd, &m.tex_kd);
            put(m.ks, p.FindOneSpectrum("Ks", Spectrum(0.25f))); bind("Ks", m.ks, &m.tex_ks);
            put(m.kr, p.FindOneSpectrum("Kr", Spectrum(0.f))); bind("Kr", m.kr, &m.tex_kr);
            put(m.kt, p.FindOneSpectrum("Kt", Spectrum(0.f))); bind("Kt", m.kt, &m.tex_kt);
            put(m.opacity, p.FindOneSpectrum("opacity", Spectrum(1.f)));
            const float rough = p.FindOneFloat("roughness", .1f);
            m.uroughness = p.Has("uroughness") ? p.FindOneFloat("uroughness", rough) : rough; // UberMaterial::computeScatteringFunctions, :563-570
            m.vroughness = p.Has("vroughness") ? p.FindOneFloat("vroughness", m.uroughness) : m.uroughness;
            m.index = p.Has("eta") ? p.FindOneFloat("eta", 1.5f) : p.FindOneFloat("index", 1.5f);
        } else if (name == "mix") { // MakeMaterial's "mix" branch and CreateMixMaterial, Enigma/Api.cpp:153-176, Enigma/Material.cpp:617-621
            m.type = ENIGMA_MAT_MIX;
            constantOnly("amount");
            const std::string names[2] = {p.FindOneString("namedmaterial1", ""), p.FindOneString("namedmaterial2", "")};
            uint32_t* slots[2] = {&m.mix_a, &m.mix_b};
            for (int k = 0; k < 2; ++k) {
                auto it = m_gs.namedMaterials.find(names[k]);
                int index;
                if (it == m_gs.namedMaterials.end()) {
                    Warn("Named material \"" + names[k] + "\" undefined.  Using \"matte\"");
                    ParamSet forMatte = p; // the reference builds the stand-in from the mix's own parameters
                    index = MakeMaterial("matte", forMatte);
                } else index = it->second;
                if (index < 0) throw std::runtime_error("Material \"mix\": named material \"" + names[k] + "\" is \"none\"; a mix needs two materials");
                *slots[k] = (uint32_t)index;
            }
            put(m.amount, p.FindOneSpectrum("amount", Spectrum(0.5f)));
        } else if (name == "fourier" || name == "subsurface" || name == "kdsubsurface" || name == "hair" || name == "disney") {
            throw std::runtime_error("Material \"" + name + "\" is not supported (matte, metal, glass, plastic, mirror, translucent, substrate, uber, mix)");
        } else { // Enigma/Api.cpp:189-192
            Warn("Material \"" + name + "\" unknown. Using \"matte\".");
            m.type = ENIGMA_MAT_MATTE;
            put(m.kd, p.FindOneSpectrum("Kd", Spectrum(0.5f))); bind("Kd", m.kd, &m.tex_kd);
            m.sigma = p.FindOneFloat("sigma", 0.f);
        }
        p.FindOneString("type", ""); // consumed by MakeNamedMaterial
        NoteUnused("Material", p);
        return m_res->scene.AddMaterial(m);
    }

    void VerifyOptions(const char* fn) const { if (m_inWorld) throw std::runtime_error(std::string("Options cannot be set inside world block; \"") + fn + "\" not allowed."); }
    void VerifyWorld(const char* fn) const { if (!m_inWorld) throw std::runtime_error(std::string("Scene description must be inside world block; \"") + fn + "\" not allowed."); }
    void Warn(const std::string& w) { m_res->warnings.push_back(w); }
    void NoteUnused(const char* what, const ParamSet& p) { for (const auto& u : p.Unused()) Warn(std::string(what) + ": parameter \"" + u + "\" not used"); }

    // ---- tokenizer + directive loop
    struct Tok { std::string s; bool quoted = false; };
    static std::vector<Tok> Tokenize(const std::string& src) {
        std::vector<Tok> out;
        size_t i = 0, n = src.size();
        while (i < n) {
            char c = src[i];
            if (isspace((unsigned char)c)) { ++i; continue; }
            if (c == '#') { while (i < n && src[i] != '\n') ++i; continue; }
            if (c == '[' || c == ']') { out.push_back({std::string(1, c), false}); ++i; continue; }
            if (c == '"') {
                size_t j = i + 1;
                std::string s;
                while (j < n && src[j] != '"') {
                    if (src[j] == '\n') throw std::runtime_error("unterminated string in scene description");
                    if (src[j] == '\\' && j + 1 < n) { ++j; s += src[j] == 'n' ? '\n' : src[j] == 't' ? '\t' : src[j]; }
                    else s += src[j];
                    ++j;
                }
                if (j >= n) throw std::runtime_error("premature EOF inside a string");
                out.push_back({s, true});
                i = j + 1;
                continue;
            }
            size_t j = i;
            while (j < n && !isspace((unsigned char)src[j]) && src[j] != '"' && src[j] != '[' && src[j] != ']' && src[j] != '#') ++j;
            out.push_back({src.substr(i, j - i), false});
            i = j;
        }
        return out;
    }
    static bool IsNumber(const Tok& t) {
        if (t.quoted || t.s.empty()) return false;
        char* end = nullptr;
        strtod(t.s.c_str(), &end);
        return end && *end == 0;
    }
    struct Cursor {
        const std::vector<Tok>& t; size_t i = 0;
        bool done() const { return i >= t.size(); }
        const Tok& peek() const { return t[i]; }
        Tok next(const char* what) { if (done()) throw std::runtime_error(std::string("premature EOF: expected ") + what); return t[i++]; }
        float num(const char* d) { Tok k = next("a number"); if (!IsNumber(k)) throw std::runtime_error(std::string(d) + ": expected a number, got \"" + k.s + "\""); return (float)strtod(k.s.c_str(), nullptr); }
        std::string str(const char* d) { Tok k = next("a quoted string"); if (!k.quoted) throw std::runtime_error(std::string(d) + ": expected a quoted string, got \"" + k.s + "\""); return k.s; }
    };
    static ParamSet ParseParams(Cursor& c) {
        ParamSet ps;
        while (!c.done() && c.peek().quoted) {
            std::string decl = c.next("a parameter declaration").s;
            std::istringstream ds(decl);
            ParamSetItem it;
            if (!(ds >> it.type >> it.name)) throw std::runtime_error("parameter \"" + decl + "\" doesn't have a type declaration");
            static const char* known[] = {"integer", "float", "bool", "string", "point", "point3", "point3f", "point2", "vector", "vector3", "vector3f", "vector2", "normal",
                                          "normal3", "normal3f", "rgb", "color", "spectrum", "blackbody", "xyz", "texture"};
            bool ok = false;
            for (const char* k : known) ok = ok || it.type == k;
            if (!ok) throw std::runtime_error("Unable to decode type of parameter \"" + decl + "\"");
            bool bracket = !c.done() && !c.peek().quoted && c.peek().s == "[";
            if (bracket) c.next("[");
            bool any = false;
            while (!c.done()) {
                const Tok& k = c.peek();
                if (!k.quoted && k.s == "]") break;
                if (!bracket && any) break;
                if (k.quoted) it.strs.push_back(k.s);
                else if (IsNumber(k)) it.nums.push_back(strtod(k.s.c_str(), nullptr));
                else if (k.s == "true" || k.s == "false") it.strs.push_back(k.s);
                else throw std::runtime_error("parameter \"" + decl + "\": unexpected token \"" + k.s + "\"");
                c.next("a value");
                any = true;
            }
            if (bracket) { Tok e = c.next("]"); if (e.quoted || e.s != "]") throw std::runtime_error("parameter \"" + decl + "\": expected ]"); }
            if (!any) throw std::runtime_error("parameter \"" + decl + "\" has no values");
            ps.Add(it);
        }
        return ps;
    }
    void Parse(const std::string& src, const std::string& dir) {
        std::vector<Tok> toks = Tokenize(src);
        Cursor c{toks};
        auto nums = [&](int n, const char* d, float* out) {
            bool bracket = !c.done() && !c.peek().quoted && c.peek().s == "[";
            if (bracket) c.next("[");
            for (int i = 0; i < n; ++i) out[i] = c.num(d);
            if (bracket) { Tok e = c.next("]"); if (e.s != "]") throw std::runtime_error(std::string(d) + ": expected ]"); }
        };
        while (!c.done()) {
            Tok t = c.next("a directive");
            if (t.quoted) throw std::runtime_error("unexpected string \"" + t.s + "\" where a directive was expected");
            const std::string& d = t.s;
            float v[16];
            if (d == "Identity") pbrtIdentity();
            else if (d == "Translate") { nums(3, "Translate", v); pbrtTranslate(v[0], v[1], v[2]); }
            else if (d == "Scale") { nums(3, "Scale", v); pbrtScale(v[0], v[1], v[2]); }
            else if (d == "Rotate") { nums(4, "Rotate", v); pbrtRotate(v[0], v[1], v[2], v[3]); }
            else if (d == "LookAt") { nums(9, "LookAt", v); pbrtLookAt(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
            else if (d == "ConcatTransform") { nums(16, "ConcatTransform", v); pbrtConcatTransform(v); }
            else if (d == "Transform") { nums(16, "Transform", v); pbrtTransform(v); }
            else if (d == "CoordinateSystem") pbrtCoordinateSystem(c.str("CoordinateSystem"));
            else if (d == "CoordSysTransform") pbrtCoordSysTransform(c.str("CoordSysTransform"));
            else if (d == "Camera") { std::string n = c.str("Camera"); pbrtCamera(n, ParseParams(c)); }
            else if (d == "Film") { std::string n = c.str("Film"); pbrtFilm(n, ParseParams(c)); }
            else if (d == "Sampler") { std::string n = c.str("Sampler"); pbrtSampler(n, ParseParams(c)); }
            else if (d == "PixelFilter") { std::string n = c.str("PixelFilter"); pbrtPixelFilter(n, ParseParams(c)); }
            else if (d == "Integrator" || d == "SurfaceIntegrator") { std::string n = c.str("Integrator"); pbrtIntegrator(n, ParseParams(c)); }
            else if (d == "Accelerator") { std::string n = c.str("Accelerator"); pbrtAccelerator(n, ParseParams(c)); }
            else if (d == "WorldBegin") pbrtWorldBegin();
            else if (d == "WorldEnd") pbrtWorldEnd();
            else if (d == "AttributeBegin") pbrtAttributeBegin();
            else if (d == "AttributeEnd") pbrtAttributeEnd();
            else if (d == "TransformBegin") pbrtTransformBegin();
            else if (d == "TransformEnd") pbrtTransformEnd();
            else if (d == "ReverseOrientation") pbrtReverseOrientation();
            else if (d == "Material") { std::string n = c.str("Material"); pbrtMaterial(n, ParseParams(c)); }
            else if (d == "MakeNamedMaterial") { std::string n = c.str("MakeNamedMaterial"); pbrtMakeNamedMaterial(n, ParseParams(c)); }
            else if (d == "NamedMaterial") pbrtNamedMaterial(c.str("NamedMaterial"));
            else if (d == "LightSource") { std::string n = c.str("LightSource"); pbrtLightSource(n, ParseParams(c), dir); }
            else if (d == "AreaLightSource") { std::string n = c.str("AreaLightSource"); pbrtAreaLightSource(n, ParseParams(c)); }
            else if (d == "Shape") { std::string n = c.str("Shape"); pbrtShape(n, ParseParams(c), dir); }
            else if (d == "Texture") { std::string n = c.str("Texture"), ty = c.str("Texture"), cl = c.str("Texture"); pbrtTexture(n, ty, cl, ParseParams(c), dir); }
            else if (d == "Include") {
                std::string inc = c.str("Include");
                pbrtParseFile((!inc.empty() && inc[0] == '/') ? inc : dir + inc);
            }
            else if (d == "ObjectBegin") pbrtObjectBegin(c.str("ObjectBegin"));
            else if (d == "ObjectEnd" || d == "ObjectInstance") throw std::runtime_error(d + ": object instancing is not supported");
            else if (d == "MakeNamedMedium") { std::string n = c.str("MakeNamedMedium"); pbrtMakeNamedMedium(n, ParseParams(c)); }
            else if (d == "MediumInterface") { // one name = both sides (Enigma/Parser.cpp:980-999)
                std::string in = c.str("MediumInterface"), out = in;
                if (!c.done() && c.peek().quoted) out = c.str("MediumInterface");
                pbrtMediumInterface(in, out);
            }
            else if (d == "ActiveTransform" || d == "TransformTimes") throw std::runtime_error(d + ": animated transforms are not supported");
            else throw std::runtime_error("unknown directive \"" + d + "\"");
        }
    }

    bool m_render;
    int m_rank = 0, m_world = 1;
    unsigned char m_commId[ENIGMA_COMM_ID_BYTES] = {};
    int m_device;
    bool m_inWorld = false;
    Mat4d m_ctm;
    std::map<std::string, Mat4d> m_namedCoordinateSystems;
    RenderOptions m_opt;
    GraphicsState m_gs;
    std::vector<GraphicsState> m_pushedGraphicsStates;
    std::vector<Mat4d> m_pushedTransforms;
    std::string m_imageFile;
    int m_defaultMaterial = -1;
    struct NamedTexture { Spectrum value = Spectrum(1.f); uint32_t index = 0; }; // index: tex_* value of an image texture, 0 = the constant
    std::map<std::string, NamedTexture> m_spectrumTextures;
    std::map<std::string, float> m_floatTextures; // every float texture made here is a constant
    std::map<std::string, int> m_namedMedia; // RenderOptions::namedMedia: name -> index in the scene's medium registry
    std::unique_ptr<FrontEndResult> m_res{new FrontEndResult()};
};

} // namespace RayTrace
