// Device-side scalar/vector helpers for the gfx950 kernels.
//
// Operation order follows the reference's glm-typed wrappers (Enigma/MathCommon.h:190-382):
// dot = x*x' + y*y' + z*z' left to right, normalize = v * (1/sqrt(dot(v,v))), division per
// component.  The translation unit is built with -ffp-contract=off so no product-sum here is
// fused; HIP's default correctly rounded f32 divide/sqrt gives IEEE results for '/' and
// sqrtf, which the bit-exact tiers (camera rays, slab test, triangle test) rely on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__

namespace edev {

constexpr float kPi = 3.1415926535897932384f;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kPiOver2 = kPi / 2.0f;
constexpr float kPiOver4 = kPi / 4.0f;
constexpr float kOneMinusEpsilon = 0.9999999403953552f;
constexpr float kMachineEpsilon = 5.9604644775390625e-08f; // numeric_limits<float>::epsilon() * 0.5
constexpr float kShadowEpsilon = 0.0001f;
#define EDEV_INF __builtin_huge_valf()
#define EDEV_MAXF 3.402823466e+38f

// (n * eps) / (1 - n * eps), Enigma/MathCommon.h:133-135 -- folded at compile time in float
DEV constexpr float gammaf(int n) { return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon); }

struct F3 {
    float x, y, z;
    DEV F3() {}
    DEV F3(float a, float b, float c) : x(a), y(b), z(c) {}
    DEV explicit F3(float s) : x(s), y(s), z(s) {}
};
DEV F3 operator+(F3 a, F3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV F3 operator-(F3 a, F3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV F3 operator-(F3 a) { return F3(-a.x, -a.y, -a.z); }
DEV F3 operator*(F3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
DEV F3 operator*(float s, F3 a) { return F3(a.x * s, a.y * s, a.z * s); }
DEV F3 operator*(F3 a, F3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV F3 operator/(F3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
DEV F3 operator/(F3 a, F3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }
DEV F3& operator+=(F3& a, F3 b) { a = a + b; return a; }
DEV F3& operator*=(F3& a, F3 b) { a = a * b; return a; }
DEV F3& operator*=(F3& a, float s) { a = a * s; return a; }
DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV float absdot(F3 a, F3 b) { return fabsf(dot(a, b)); }
DEV float lengthSqr(F3 a) { return dot(a, a); }
DEV float length(F3 a) { return sqrtf(dot(a, a)); } // glm::length
DEV F3 cross(F3 x, F3 y) { return F3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
DEV F3 normalize(F3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
DEV F3 fabs3(F3 v) { return F3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }
DEV F3 faceForward(F3 n1, F3 n2) { return dot(n1, n2) < 0.0f ? -n1 : n1; }
// std::max(a,b) = (a < b) ? b : a ; std::min(a,b) = (b < a) ? b : a  (NaN behaviour of the reference)
DEV float smax(float a, float b) { return (a < b) ? b : a; }
DEV float smin(float a, float b) { return (b < a) ? b : a; }
DEV float clampf(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
DEV float maxComponent(F3 v) { return smax(v.x, smax(v.y, v.z)); }
DEV bool isBlack(F3 s) { return s.x == 0.f && s.y == 0.f && s.z == 0.f; }
DEV float maxComponentValue(F3 s) { return smax(smax(s.x, s.y), s.z); } // CoefficientSpectrum::MaxComponentValue
DEV float luminance(F3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; } // Spectrum.cpp:1483-1487
DEV F3 sqrt3(F3 s) { return F3(sqrtf(s.x), sqrtf(s.y), sqrtf(s.z)); }

DEV float4 ld4(const float4* p) { return *p; }
DEV F3 xyz(float4 v) { return F3(v.x, v.y, v.z); }
DEV float4 mk4(F3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// Enigma/MathCommon.h:371-382
DEV void coordinateSystem(F3 v1, F3* v2, F3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float invLen = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = F3(-v1.z * invLen, 0.f, v1.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = F3(0.f, v1.z * invLen, -v1.y * invLen);
    }
    *v3 = cross(v1, *v2);
}

// Enigma/Misc.cpp:26-52
DEV float nextFloatUp(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
DEV float nextFloatDown(float v) {
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}
// Enigma/Misc.cpp:54-68
DEV F3 offsetRayOrigin(F3 p, F3 pError, F3 n, F3 w) {
    float d = dot(fabs3(n), pError);
    F3 offset = d * n;
    if (dot(w, n) < 0) offset = -offset;
    F3 po = p + offset;
    if (offset.x > 0) po.x = nextFloatUp(po.x); else if (offset.x < 0) po.x = nextFloatDown(po.x);
    if (offset.y > 0) po.y = nextFloatUp(po.y); else if (offset.y < 0) po.y = nextFloatDown(po.y);
    if (offset.z > 0) po.z = nextFloatUp(po.z); else if (offset.z < 0) po.z = nextFloatDown(po.z);
    return po;
}

} // namespace edev
