// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference renderer's arithmetic.
// Nothing under oracle/ is linked into or called by the product library; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Vector / spectrum helpers with the exact operation order of the reference's
// glm-typed wrappers (Enigma/MathCommon.h:190-382, Enigma/Defines.h:71-101).
// glm itself is not vendored in the reference (unpinned, >= 0.9.9): the formulas
// below are glm 0.9.9's published scalar definitions -- parity is unpinned at that
// boundary.  Build with -O2 -ffp-contract=off -fno-fast-math (strict IEEE binary32).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

static constexpr float PI = 3.1415926535897932384f;         // MathCommon.h:12
static constexpr float INV_PI = 1.0f / PI;                   // MathCommon.h:18
static constexpr float PI_OVER_TWO = PI / 2.0f;              // MathCommon.h:22
static constexpr float PI_OVER_FOUR = PI / 4.0f;             // MathCommon.h:23
static constexpr float OneMinusEpsilon = 0.9999999403953552f; // MathCommon.h:25
static constexpr float InfinityF32 = std::numeric_limits<float>::infinity();
static constexpr float MachineEpsilon = (std::numeric_limits<float>::epsilon() * 0.5); // MathCommon.h:29
static constexpr float ShadowEpsilon = 0.0001f;             // Interaction.h:9

// MathCommon.h:133-135
inline float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }

struct Vec2 {
    float x = 0, y = 0;
    Vec2() {}
    Vec2(float a, float b) : x(a), y(b) {}
    float operator[](int i) const { return i == 0 ? x : y; }
    float& operator[](int i) { return i == 0 ? x : y; }
};
inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator*(Vec2 a, float s) { return {a.x * s, a.y * s}; }
inline Vec2 operator*(float s, Vec2 a) { return a * s; } // MathCommon.h:173-174

struct Vec2i {
    int x = 0, y = 0;
    Vec2i() {}
    Vec2i(int a, int b) : x(a), y(b) {}
    int operator[](int i) const { return i == 0 ? x : y; }
};

struct Vec3 {
    float x = 0, y = 0, z = 0; // GLM_FORCE_CTOR_INIT: zero-initialised (Defines.h:16-17)
    Vec3() {}
    explicit Vec3(float s) : x(s), y(s), z(s) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return a * s; } // MathCommon.h:170-171
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }
inline Vec3& operator*=(Vec3& a, float s) { a = a * s; return a; }

// glm::dot for vec3: tmp = a*b; tmp.x + tmp.y + tmp.z
inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float Dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
inline float LengthSqr(Vec3 v) { return Dot(v, v); }          // MathCommon.h:204-207
inline float LengthSqr(Vec2 v) { return Dot(v, v); }
inline float Length(Vec3 v) { return std::sqrt(Dot(v, v)); }  // glm::length
inline float DistanceSqr(Vec3 a, Vec3 b) { return LengthSqr(a - b); }
inline float Distance(Vec3 a, Vec3 b) { return Length(a - b); } // glm::distance = length(p0 - p1)
inline float AbsDot(Vec3 a, Vec3 b) { return std::abs(Dot(a, b)); }
// glm::cross
inline Vec3 Cross(Vec3 x, Vec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline Vec3 Abs(Vec3 v) { return {std::abs(v.x), std::abs(v.y), std::abs(v.z)}; }
// glm::normalize: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline Vec3 Normalize(Vec3 v) { return v * (1.0f / std::sqrt(Dot(v, v))); }
inline Vec3 FaceForward(Vec3 n1, Vec3 n2) { return Dot(n1, n2) < 0.0f ? -n1 : n1; } // MathCommon.h:266-272
inline float MaxComponent(Vec3 v) { return std::max(v.x, std::max(v.y, v.z)); } // MathCommon.h:274-277
inline int MaxDimension(Vec3 v) { // MathCommon.h:342-353
    if (v.x > v.y) {
        if (v.x > v.z) return 0;
        return 2;
    }
    if (v.y > v.z) return 1;
    return 2;
}
inline Vec3 Permute(Vec3 v, int x, int y, int z) { return {v[x], v[y], v[z]}; }
// glm::min(a,b) = (b < a) ? b : a ; glm::max(a,b) = (a < b) ? b : a
inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }
inline Vec3 Min(Vec3 a, Vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline Vec3 Max(Vec3 a, Vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }

// MathCommon.h:371-382
inline void CoordinateSystem(Vec3 v1, Vec3* v2, Vec3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float invLen = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = Vec3(-v1.z * invLen, 0.f, v1.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = Vec3(0.f, v1.z * invLen, -v1.y * invLen);
    }
    *v3 = Cross(v1, *v2);
}

inline float clampf(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; } // std::clamp

// Misc.cpp:12-52
inline uint32_t FloatToBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float BitsToFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float NextFloatUp(float v) {
    if (std::isinf(v) && v > 0.) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = FloatToBits(v);
    if (v >= 0) ++ui; else --ui;
    return BitsToFloat(ui);
}
inline float NextFloatDown(float v) {
    if (std::isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = FloatToBits(v);
    if (v > 0) --ui; else ++ui;
    return BitsToFloat(ui);
}
// Misc.cpp:54-68
inline Vec3 OffsetRayOrigin(Vec3 p, Vec3 pError, Vec3 n, Vec3 w) {
    float d = Dot(Abs(n), pError);
    Vec3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    Vec3 po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0) po[i] = NextFloatUp(po[i]);
        else if (offset[i] < 0) po[i] = NextFloatDown(po[i]);
    }
    return po;
}

// RGBSpectrum = CoefficientSpectrum<3> (Enigma/Spectrum.h:65-313, Defines.h:157)
struct Spectrum {
    float c[3];
    Spectrum(float v = 0.f) { c[0] = c[1] = c[2] = v; }
    Spectrum(float r, float g, float b) { c[0] = r; c[1] = g; c[2] = b; }
    Spectrum& operator+=(const Spectrum& s) { for (int i = 0; i < 3; ++i) c[i] += s.c[i]; return *this; }
    Spectrum operator+(const Spectrum& s) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] += s.c[i]; return r; }
    Spectrum operator-(const Spectrum& s) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] -= s.c[i]; return r; }
    Spectrum operator/(const Spectrum& s) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] /= s.c[i]; return r; }
    Spectrum operator*(const Spectrum& s) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] *= s.c[i]; return r; }
    Spectrum& operator*=(const Spectrum& s) { for (int i = 0; i < 3; ++i) c[i] *= s.c[i]; return *this; }
    Spectrum operator*(float a) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] *= a; return r; }
    Spectrum& operator*=(float a) { for (int i = 0; i < 3; ++i) c[i] *= a; return *this; }
    Spectrum operator/(float a) const { Spectrum r = *this; for (int i = 0; i < 3; ++i) r.c[i] /= a; return r; }
    Spectrum& operator/=(float a) { for (int i = 0; i < 3; ++i) c[i] /= a; return *this; }
    bool IsBlack() const { for (int i = 0; i < 3; ++i) if (c[i] != 0.) return false; return true; }
    Spectrum Clamp(float low = 0, float high = InfinityF32) const {
        Spectrum r; for (int i = 0; i < 3; ++i) r.c[i] = clampf(c[i], low, high); return r;
    }
    float MaxComponentValue() const { float m = c[0]; for (int i = 1; i < 3; ++i) m = std::max(m, c[i]); return m; }
    bool HasNaNs() const { for (int i = 0; i < 3; ++i) if (std::isnan(c[i])) return true; return false; }
    // Spectrum.cpp:1483-1487
    float y() const { const float YWeight[3] = {0.212671f, 0.715160f, 0.072169f};
        return YWeight[0] * c[0] + YWeight[1] * c[1] + YWeight[2] * c[2]; }
    // Spectrum.h:28-32
    void ToXYZ(float xyz[3]) const {
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
    }
};
inline Spectrum operator*(float a, const Spectrum& s) { return s * a; }
inline Spectrum Sqrt(const Spectrum& s) { Spectrum r; for (int i = 0; i < 3; ++i) r.c[i] = std::sqrt(s.c[i]); return r; }
// Spectrum.h:22-26
inline void XYZToRGB(const float xyz[3], float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

// MathCommon.h:92-124
inline uint32_t RoundUpPow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline int Floor2Int(float v) { return (int)floorf(v); }
inline int Ceil2Int(float v) { return (int)ceilf(v); }
inline float Log2(float x) { static const float invLog2 = 1.f / logf(2.f); return logf(x) * invLog2; }
inline int Log2Int(float v) { return Floor2Int(Log2(v)); }

} // namespace orc
