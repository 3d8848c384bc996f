// Sphere shapes (Enigma/ShapeImp.cpp:13-203): intersection with the reference's running-error arithmetic (Enigma/eFloat.h)
// and error-carrying transforms (Enigma/Transform.h:212-330), and the surface interaction at a hit
// (Enigma/Interaction.cpp:101-130, Enigma/Transform.cpp:322-356).  Spheres are leaf entries of the BVH beside the triangles
// (dev_trace8.h).
#pragma once
#include "dev_common.h"
#include "dev_types.h"

namespace edev {

struct EFloat { // release build of the reference's class: value and conservative bounds
    float v, low, high;
    DEV EFloat() : v(0.f), low(0.f), high(0.f) {}
    DEV EFloat(float v_, float err = 0.f) : v(v_) {
        if (err == 0.f) low = high = v_;
        else { low = nextFloatDown(v_ - err); high = nextFloatUp(v_ + err); }
    }
    DEV EFloat operator+(EFloat e) const { EFloat r; r.v = v + e.v; r.low = nextFloatDown(low + e.low); r.high = nextFloatUp(high + e.high); return r; }
    DEV EFloat operator-(EFloat e) const { EFloat r; r.v = v - e.v; r.low = nextFloatDown(low - e.high); r.high = nextFloatUp(high - e.low); return r; }
    DEV EFloat operator*(EFloat e) const {
        EFloat r; r.v = v * e.v;
        float p0 = low * e.low, p1 = high * e.low, p2 = low * e.high, p3 = high * e.high;
        r.low = nextFloatDown(smin(smin(p0, p1), smin(p2, p3)));
        r.high = nextFloatUp(smax(smax(p0, p1), smax(p2, p3)));
        return r;
    }
    DEV EFloat operator/(EFloat e) const {
        EFloat r; r.v = v / e.v;
        if (e.low < 0 && e.high > 0) { r.low = -EDEV_INF; r.high = EDEV_INF; }
        else {
            float d0 = low / e.low, d1 = high / e.low, d2 = low / e.high, d3 = high / e.high;
            r.low = nextFloatDown(smin(smin(d0, d1), smin(d2, d3)));
            r.high = nextFloatUp(smax(smax(d0, d1), smax(d2, d3)));
        }
        return r;
    }
};
// std::min / std::max as the reference's EFloat calls them: (b < a) ? b : a and (a < b) ? b : a -- smin / smax above

// Quadratic, Enigma/eFloat.h:225-243
DEV bool quadraticE(EFloat A, EFloat B, EFloat C, EFloat* t0, EFloat* t1) {
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = sqrt(discrim);
    EFloat floatRootDiscrim((float)rootDiscrim, (float)((double)kMachineEpsilon * rootDiscrim));
    EFloat q;
    if (B.v < 0) q = EFloat(-.5f) * (B - floatRootDiscrim);
    else q = EFloat(-.5f) * (B + floatRootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) { EFloat s = *t0; *t0 = *t1; *t1 = s; }
    return true;
}

// rows 0..2 of an affine matrix, m[4 * r + c]
DEV F3 xfPointErr(const float* m, F3 p, F3* err) { // Transform::transformPoint(p, &pError), Enigma/Transform.h:212-235
    float x = p.x, y = p.y, z = p.z;
    F3 q((m[0] * x + m[1] * y) + (m[2] * z + m[3]), (m[4] * x + m[5] * y) + (m[6] * z + m[7]), (m[8] * x + m[9] * y) + (m[10] * z + m[11]));
    *err = gammaf(3) * F3((fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3])), (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7])),
                          (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11])));
    return q;
}
DEV F3 xfPointErr2(const float* m, F3 p, F3 e, F3* err) { // transformPoint(pt, ptError, &absError), Enigma/Transform.h:237-269
    float x = p.x, y = p.y, z = p.z;
    F3 q((m[0] * x + m[1] * y) + (m[2] * z + m[3]), (m[4] * x + m[5] * y) + (m[6] * z + m[7]), (m[8] * x + m[9] * y) + (m[10] * z + m[11]));
    err->x = (gammaf(3) + 1.f) * (fabsf(m[0]) * e.x + fabsf(m[1]) * e.y + fabsf(m[2]) * e.z) + gammaf(3) * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    err->y = (gammaf(3) + 1.f) * (fabsf(m[4]) * e.x + fabsf(m[5]) * e.y + fabsf(m[6]) * e.z) + gammaf(3) * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    err->z = (gammaf(3) + 1.f) * (fabsf(m[8]) * e.x + fabsf(m[9]) * e.y + fabsf(m[10]) * e.z) + gammaf(3) * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    return q;
}
DEV F3 xfVec3(const float* m, F3 v) { return F3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }
DEV F3 xfVecErr(const float* m, F3 v, F3* err) { // transformVector(v, &absError), Enigma/Transform.h:271-287
    *err = gammaf(3) * F3(fabsf(m[0] * v.x) + fabsf(m[1] * v.y) + fabsf(m[2] * v.z), fabsf(m[4] * v.x) + fabsf(m[5] * v.y) + fabsf(m[6] * v.z),
                          fabsf(m[8] * v.x) + fabsf(m[9] * v.y) + fabsf(m[10] * v.z));
    return xfVec3(m, v);
}
DEV F3 xfNormalByInverse(const float* inv, F3 n) { // Transform::transformNormal: the inverse's transpose, Enigma/Transform.h:54-59
    return F3(inv[0] * n.x + inv[4] * n.y + inv[8] * n.z, inv[1] * n.x + inv[5] * n.y + inv[9] * n.z, inv[2] * n.x + inv[6] * n.y + inv[10] * n.z);
}

// Sphere::intersect / intersectP up to the accepted hit point (Enigma/ShapeImp.cpp:38-92,147-201): object-space hit
// point, object-space ray direction and t
DEV bool sphereHit(const DevSphere& s, F3 wo_, F3 wd_, float tMaxRay, F3* pHitOut, F3* dObjOut, float* tOut) {
    F3 oErr, dErr;
    F3 o = xfPointErr(s.w2o, wo_, &oErr); // Transform::transformRay(r, &oErr, &dErr), Enigma/Transform.h:317-330
    F3 d = xfVecErr(s.w2o, wd_, &dErr);
    float lengthSquared = lengthSqr(d);
    if (lengthSquared > 0) { float dt = dot(fabs3(d), oErr) / lengthSquared; o += d * dt; }
    EFloat ox(o.x, oErr.x), oy(o.y, oErr.y), oz(o.z, oErr.z);
    EFloat dx(d.x, dErr.x), dy(d.y, dErr.y), dz(d.z, dErr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = EFloat(2.f) * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(s.radius) * EFloat(s.radius);
    EFloat t0, t1;
    if (!quadraticE(a, b, c, &t0, &t1)) return false;
    if (t0.high > tMaxRay || t1.low <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > tMaxRay) return false;
    }
    const float PI2 = 2 * 3.14159265358979323846f;
    F3 pHit = o + d * tShapeHit.v;
    pHit *= s.radius / length(pHit - F3(0.f, 0.f, 0.f));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
    float phi = atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += PI2;
    if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || phi > s.phiMax) {
        if (tShapeHit.v == t1.v) return false;
        if (t1.high > tMaxRay) return false;
        tShapeHit = t1;
        pHit = o + d * tShapeHit.v;
        pHit *= s.radius / length(pHit - F3(0.f, 0.f, 0.f));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * s.radius;
        phi = atan2f(pHit.y, pHit.x);
        if (phi < 0) phi += PI2;
        if ((s.zMin > -s.radius && pHit.z < s.zMin) || (s.zMax < s.radius && pHit.z > s.zMax) || phi > s.phiMax) return false;
    }
    *pHitOut = pHit; *dObjOut = d; *tOut = tShapeHit.v;
    return true;
}

// The SurfaceInteraction of a sphere hit in world space: p, pError, geometric normal, dpdu, and wo -- which the
// reference transforms back to world space WITHOUT renormalising (Enigma/Transform.cpp:330)
// dpdv, uv: optional (the texture kernel asks for them)
DEV void sphereSurface(const DevSphere& s, F3 pHit, F3 rayDirWorld, F3* p, F3* pError, F3* n, F3* dpdu, F3* wo, F3* dpdv = nullptr, float* u = nullptr,
                       float* v = nullptr) {
    float theta = acosf(smin(smax(pHit.z / s.radius, -1.0f), 1.0f));
    float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    F3 dpduO(-s.phiMax * pHit.y, s.phiMax * pHit.x, 0.f);
    F3 dpdvO = (s.thetaMax - s.thetaMin) * F3(pHit.z * cosPhi, pHit.z * sinPhi, -s.radius * sinf(theta));
    F3 pErrO = gammaf(5) * fabs3(pHit);
    F3 nObj = normalize(cross(dpduO, dpdvO));
    if (s.mw >> 31) nObj = -nObj;
    F3 dErr;
    F3 dObj = xfVecErr(s.w2o, rayDirWorld, &dErr);
    *p = xfPointErr2(s.o2w, pHit, pErrO, pError);
    *n = normalize(xfNormalByInverse(s.w2o, nObj));
    *wo = xfVec3(s.o2w, normalize(-dObj));
    *dpdu = xfVec3(s.o2w, dpduO);
    if (dpdv) *dpdv = xfVec3(s.o2w, dpdvO);
    if (u) { // Enigma/ShapeImp.cpp:72-73,95-97
        float phi = atan2f(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * 3.14159265358979323846f;
        *u = phi / s.phiMax;
        *v = (theta - s.thetaMin) / (s.thetaMax - s.thetaMin);
    }
}

// Sphere::sample(ref, u) (Enigma/ShapeImp.cpp:210-293) as Shape-side half of DiffuseAreaLight::Sample_Li: the sampled
// point as an Interaction and the pdf over solid angle at the reference point.  The reference compares its object-space
// radius with world-space distances and samples the whole sphere whatever its z range and sweep; so does this.
DEV void sphereSampleFrom(const DevSphere& s, F3 refP, F3 refErr, F3 refN, float u0, float u1, F3* lp, F3* lpErr, F3* ln, float* pdfOut) {
    const float kPiS = 3.14159265358979323846f;
    const bool reverse = (s.mw >> 29) & 1u; // m_reverseOrientation alone
    const F3 pCenter(s.o2w[3], s.o2w[7], s.o2w[11]);
    F3 pOrigin = offsetRayOrigin(refP, refErr, refN, pCenter - refP);
    if (lengthSqr(pOrigin - pCenter) <= s.radius * s.radius) { // inside: the whole sphere by area, Sphere::sample(u)
        float z = 1.f - 2.f * u0; // UniformSampleSphere, Enigma/MonteCarlo.cpp:97-105
        float r = sqrtf(smax(0.f, 1.f - z * z));
        float phi = 2.f * kPiS * u1;
        F3 pObj = F3(0.f, 0.f, 0.f) + s.radius * F3(r * cosf(phi), r * sinf(phi), z);
        F3 unused;
        F3 n = normalize(xfPointErr(s.o2w, pObj, &unused)); // sic: the point transform, translation included (:214)
        if (reverse) n = -n;
        pObj *= s.radius / length(pObj - F3(0.f, 0.f, 0.f));
        F3 pObjError = gammaf(5) * fabs3(pObj);
        F3 p = xfPointErr2(s.o2w, pObj, pObjError, lpErr);
        float pdf = 1 / (s.phiMax * s.radius * (s.zMax - s.zMin));
        F3 wi = p - refP;
        if (lengthSqr(wi) == 0) pdf = 0;
        else {
            wi = normalize(wi);
            pdf *= lengthSqr(refP - p) / absdot(n, -wi);
        }
        if (isinf(pdf)) pdf = 0.f;
        *lp = p; *ln = n; *pdfOut = pdf;
        return;
    }
    float dc = length(refP - pCenter);
    float invDc = 1 / dc;
    F3 wc = (pCenter - refP) * invDc;
    F3 wcX, wcY;
    coordinateSystem(wc, &wcX, &wcY);
    float sinThetaMax = s.radius * invDc;
    float sinThetaMax2 = sinThetaMax * sinThetaMax;
    float invSinThetaMax = 1 / sinThetaMax;
    float cosThetaMax = sqrtf(smax(0.f, 1 - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u0 + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u0;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(smax(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    float sinAlpha = sqrtf(smax(0.f, 1.f - cosAlpha * cosAlpha));
    float phi = u1 * 2 * kPiS;
    F3 nWorld = (sinAlpha * cosf(phi) * (-wcX)) + (sinAlpha * sinf(phi) * (-wcY)) + (cosAlpha * (-wc)); // SphericalDirection, Enigma/MathCommon.cpp:18-23
    F3 pWorld = pCenter + s.radius * nWorld;
    *lp = pWorld;
    *lpErr = gammaf(5) * fabs3(pWorld);
    *ln = reverse ? -nWorld : nWorld;
    *pdfOut = 1 / (2 * kPiS * (1 - cosThetaMax));
}

// Sphere::pdf(ref, wi) (Enigma/ShapeImp.cpp:297-310): the cone's pdf from outside (whether or not wi meets the sphere),
// Shape::pdf (Enigma/Shape.cpp:80-98) over the ray (rayO, wi) = ref.SpawnRay(wi) from inside
DEV float spherePdfFrom(const DevSphere& s, F3 refP, F3 refErr, F3 refN, F3 rayO, F3 wi) {
    const float kPiS = 3.14159265358979323846f;
    const F3 pCenter(s.o2w[3], s.o2w[7], s.o2w[11]);
    F3 pOrigin = offsetRayOrigin(refP, refErr, refN, pCenter - refP);
    if (lengthSqr(pOrigin - pCenter) <= s.radius * s.radius) {
        F3 pHit, dObj; float t;
        if (!sphereHit(s, rayO, wi, EDEV_INF, &pHit, &dObj, &t)) return 0.f;
        F3 p, pErr, n, dpdu, wo;
        sphereSurface(s, pHit, wi, &p, &pErr, &n, &dpdu, &wo, nullptr, nullptr, nullptr);
        float pdf = lengthSqr(refP - p) / (absdot(n, -wi) * (s.phiMax * s.radius * (s.zMax - s.zMin)));
        if (isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    float sinThetaMax2 = s.radius * s.radius / lengthSqr(refP - pCenter);
    float cosThetaMax = sqrtf(smax(0.f, 1 - sinThetaMax2));
    return 1.f / (2.f * kPiS * (1.f - cosThetaMax)); // UniformConePdf, Enigma/MonteCarlo.cpp:112-115
}

} // namespace edev
