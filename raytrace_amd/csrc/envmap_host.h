// Host-side preparation of an InfiniteAreaLight (Enigma/Lights.cpp:395-442): the radiance MipMap
// (Enigma/MipMap.h:111-152: box-filtered pyramid, repeat wrap), the 2w x 2h luminance image its constructor
// tabulates through trilinear lookups (Task, Enigma/Lights.cpp:368-392), the Distribution2D over that image
// (Enigma/MonteCarlo.cpp:236-249) and the EWA lookup Power() makes (Enigma/Lights.cpp:541-546, Enigma/MipMap.h:256-399).
// The device reads level 0 and the flattened distribution only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace enigma {

struct Rgb { float r, g, b; };
inline Rgb operator+(Rgb a, Rgb b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Rgb operator-(Rgb a, Rgb b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Rgb operator*(Rgb a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline Rgb operator*(float s, Rgb a) { return {a.r * s, a.g * s, a.b * s}; }
inline Rgb operator/(Rgb a, float s) { return {a.r / s, a.g / s, a.b / s}; }
inline float luminanceY(Rgb c) { return 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b; } // Enigma/Spectrum.cpp:1483-1487

struct EnvPyramid {
    std::vector<std::vector<Rgb>> level;
    std::vector<int> lw, lh;
    float weightLut[128];
    int levels() const { return (int)level.size(); }
    static float log2f_(float x) { static const float invLog2 = 1.f / logf(2.f); return logf(x) * invLog2; } // Enigma/MathCommon.h:117-120
    // MipMap::resizeImage / resampleWeights (Enigma/MipMap.h:155-237): separable Lanczos zoom to power-of-two sides.  The signed
    // "%" of the repeat mode leaves negative texel indices negative, and those taps are then dropped like the black mode's.
    static float lanczos(float x, float tau = 2.f) { // Enigma/MipMap.h:15-23
        x = fabsf(x);
        if (x < 1e-5) return 1;
        if (x > 1.) return 0;
        x *= 3.1415926535897932384f;
        float s = sinf(x * tau) / (x * tau);
        float l = sinf(x) / x;
        return s * l;
    }
    static uint32_t roundUpPow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
    // One axis of the zoom.  A new texel is the normalised sum of four old ones under the windowed sinc above, two texel
    // widths wide, centred on the new texel's centre in old coordinates; a tap outside the image is wrapped, clamped or
    // dropped by the wrap mode (repeat wraps with the signed remainder of the reference, which leaves taps left of the image
    // outside it: they are dropped after the weights were normalised).  The image is given as `lines` lines of oldN texels,
    // texel k of line l at src[l * lineStep + k * texelStep], so rows and columns go through the same code; the result has
    // the same shape with newN texels a line.  floorNegative: Spectrum::Clamp() of the finished texel.
    void zoomAxis(const Rgb* src, size_t srcLineStep, size_t srcTexelStep, Rgb* dst, size_t dstLineStep, size_t dstTexelStep, int lines, int oldN, int newN,
                  bool floorNegative) const {
        const float halfWidth = 2.f;
        for (int i = 0; i < newN; ++i) {
            const float center = (i + .5f) * oldN / newN;
            const int first = (int)floorf((center - halfWidth) + 0.5f);
            float w[4];
            for (int j = 0; j < 4; ++j) w[j] = lanczos(((first + j + .5f) - center) / halfWidth);
            const float invSum = 1.f / (w[0] + w[1] + w[2] + w[3]);
            int from[4];
            for (int j = 0; j < 4; ++j) {
                w[j] *= invSum;
                int k = first + j;
                if (wrap == 0) k = k % oldN;
                else if (wrap == 2) k = std::min(std::max(k, 0), oldN - 1);
                from[j] = (k >= 0 && k < oldN) ? k : -1;
            }
            for (int l = 0; l < lines; ++l) {
                Rgb acc = Rgb{0.f, 0.f, 0.f};
                for (int j = 0; j < 4; ++j)
                    if (from[j] >= 0) acc = acc + src[(size_t)l * srcLineStep + (size_t)from[j] * srcTexelStep] * w[j];
                if (floorNegative) acc = Rgb{std::max(acc.r, 0.f), std::max(acc.g, 0.f), std::max(acc.b, 0.f)};
                dst[(size_t)l * dstLineStep + (size_t)i * dstTexelStep] = acc;
            }
        }
    }
    // MipMap::resizeImage (Enigma/MipMap.h:155-237): the rows first, then the columns of what the rows became
    std::vector<Rgb> resizeImage(const std::vector<Rgb>& img, int oldWidth, int oldHeight, int* newWidth, int* newHeight) const {
        const int wide = (int)roundUpPow2((uint32_t)oldWidth), tall = (int)roundUpPow2((uint32_t)oldHeight);
        std::vector<Rgb> rows((size_t)wide * oldHeight), out((size_t)wide * tall);
        zoomAxis(img.data(), (size_t)oldWidth, 1, rows.data(), (size_t)wide, 1, oldHeight, oldWidth, wide, false);
        zoomAxis(rows.data(), 1, (size_t)wide, out.data(), 1, (size_t)wide, wide, oldHeight, tall, true);
        *newWidth = wide; *newHeight = tall;
        return out;
    }
    void build(int w, int h, std::vector<Rgb> img) {
        if ((w & (w - 1)) || (h & (h - 1))) { int nw, nh; img = resizeImage(img, w, h, &nw, &nh); w = nw; h = nh; }
        for (int i = 0; i < 128; ++i) { float alpha = 2; float r2 = float(i) / float(128 - 1); weightLut[i] = expf(-alpha * r2) - expf(-alpha); }
        int n = 1 + (int)floorf(log2f_(float(std::max(w, h))));
        level.assign(n, {}); lw.assign(n, 0); lh.assign(n, 0);
        level[0] = std::move(img); lw[0] = w; lh[0] = h;
        for (int i = 1; i < n; ++i) {
            int sRes = std::max(1, lw[i - 1] / 2), tRes = std::max(1, lh[i - 1] / 2);
            lw[i] = sRes; lh[i] = tRes; level[i].resize((size_t)sRes * tRes);
            for (int t = 0; t < tRes; ++t)
                for (int s = 0; s < sRes; ++s)
                    level[i][(size_t)t * sRes + s] = .25f * (texel(i - 1, 2 * s, 2 * t) + texel(i - 1, 2 * s + 1, 2 * t) + texel(i - 1, 2 * s, 2 * t + 1) + texel(i - 1, 2 * s + 1, 2 * t + 1));
        }
    }
    int wrap = 0; // eImageWrapMode: 0 repeat, 1 black, 2 clamp (Enigma/MipMap.h:327-352)
    const Rgb& texel(int l, int s, int t) const { // repeat wrap: int % unsigned, exact for the pyramid's power-of-two sizes
        static const Rgb black{0.f, 0.f, 0.f};
        if (wrap == 2) { s = std::min(std::max(s, 0), lw[l] - 1); t = std::min(std::max(t, 0), lh[l] - 1); }
        else if (wrap == 1) { if (s < 0 || s >= lw[l] || t < 0 || t >= lh[l]) return black; }
        uint32_t us = (uint32_t)s % (uint32_t)lw[l], ut = (uint32_t)t % (uint32_t)lh[l];
        return level[l][(size_t)ut * lw[l] + us];
    }
    Rgb triangle(int l, float s0f, float t0f) const {
        l = std::min(std::max(l, 0), levels() - 1);
        float s = s0f * lw[l] - 0.5f, t = t0f * lh[l] - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * texel(l, s0, t0) + (1 - ds) * dt * texel(l, s0, t0 + 1) + ds * (1 - dt) * texel(l, s0 + 1, t0) + ds * dt * texel(l, s0 + 1, t0 + 1);
    }
    Rgb lookup(float s, float t, float width) const { // Enigma/MipMap.h:355-367
        float lv = levels() - 1.0f + log2f_(std::max(width, 1e-8f));
        if (lv < 0) return triangle(0, s, t);
        else if (lv >= levels() - 1) return texel(levels() - 1, 0, 0);
        int iLevel = (int)floorf(lv);
        float delta = lv - iLevel;
        return (1.f - delta) * triangle(iLevel, s, t) + delta * triangle(iLevel + 1, s, t);
    }
    Rgb ewa(int l, float s, float t, float d0x, float d0y, float d1x, float d1y) const { // Enigma/MipMap.h:258-306
        if (l >= levels()) return texel(levels() - 1, 0, 0);
        s = s * lw[l] - 0.5f; t = t * lh[l] - 0.5f;
        d0x *= lw[l]; d0y *= lh[l]; d1x *= lw[l]; d1y *= lh[l];
        float A = d0y * d0y + d1y * d1y + 1;
        float B = -2 * (d0x * d0y + d1x * d1y);
        float C = d0x * d0x + d1x * d1x + 1;
        float invF = 1 / (A * C - B * B * 0.25f);
        A *= invF; B *= invF; C *= invF;
        float det = -B * B + 4 * A * C;
        float invDet = 1 / det;
        float uSqrt = std::sqrt(det * C), vSqrt = std::sqrt(A * det);
        int s0 = (int)std::ceil(s - 2 * invDet * uSqrt), s1 = (int)std::floor(s + 2 * invDet * uSqrt);
        int t0 = (int)std::ceil(t - 2 * invDet * vSqrt), t1 = (int)std::floor(t + 2 * invDet * vSqrt);
        Rgb sum{0.f, 0.f, 0.f};
        float sumWts = 0;
        for (int it = t0; it <= t1; ++it) {
            float tt = it - t;
            for (int is = s0; is <= s1; ++is) {
                float ss = is - s;
                float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
                if (r2 < 1) {
                    float weight = weightLut[std::min((int)(r2 * 128), 128 - 1)];
                    sum = sum + texel(l, is, it) * weight;
                    sumWts += weight;
                }
            }
        }
        return sum / sumWts;
    }
    // lookup(st, dst0, dst1) with both differentials (0.5, 0.5), EWA filtering, maximum anisotropy 8: what Power() asks for
    Rgb lookupHalf() const {
        float d0x = .5f, d0y = .5f, d1x = .5f, d1y = .5f;
        // Enigma/MipMap.h:380-381 take dst0.length() / dst1.length(): on the reference's glm vectors (Enigma/Defines.h:72) that
        // is the component count, 2, not the Euclidean length, so the level of detail below comes out as levels()
        float majorLength = 2.f, minorLength = 2.f;
        if (minorLength * 8.0f < majorLength && minorLength > 0) { float sc = majorLength / (minorLength * 8.0f); d1x *= sc; d1y *= sc; minorLength *= sc; }
        if (minorLength == 0) return triangle(0, .5f, .5f);
        float lod = std::max((float)0, levels() - (float)1 + log2f_(minorLength));
        int ilod = (int)std::floor(lod);
        Rgb a = ewa(ilod, .5f, .5f, d0x, d0y, d1x, d1y), b = ewa(ilod + 1, .5f, .5f, d0x, d0y, d1x, d1y);
        float tl = lod - ilod; // Lerp(start, end, t), Enigma/MathCommon.h:31-39
        if (tl < 0.f) return a;
        if (tl > 1.f) return b;
        return a + (b - a) * tl;
    }
};

// Distribution1D (Enigma/MonteCarlo.h:17-38) of one row appended to flat arrays
inline float appendDistribution1D(const float* f, int n, std::vector<float>* func, std::vector<float>* cdf) {
    size_t c0 = cdf->size();
    func->insert(func->end(), f, f + n);
    cdf->resize(c0 + n + 1);
    float* c = cdf->data() + c0;
    c[0] = 0.;
    for (int i = 1; i < n + 1; ++i) c[i] = c[i - 1] + f[i - 1] / n;
    float funcInt = c[n];
    if (funcInt == 0.f) for (int i = 1; i < n + 1; ++i) c[i] = float(i) / float(n);
    else for (int i = 1; i < n + 1; ++i) c[i] /= funcInt;
    return funcInt;
}

struct HostEnvMap {
    int w = 0, h = 0, nu = 0, nv = 0;
    std::vector<float> texels;                       // level 0, float4 per texel
    std::vector<float> condFunc, condCdf, condInt, margFunc, margCdf;
    float margInt = 0.f;
    Rgb powerLookup{0.f, 0.f, 0.f};                  // m_radianceMap->lookup(half, half, half)
    // map: h x w RGB (or nullptr: one texel), L: the light's scale colour
    void build(const float* map, int width, int height, const float L[3]) {
        if (!map) { width = 1; height = 1; }
        std::vector<Rgb> tex((size_t)width * height);
        for (size_t i = 0; i < tex.size(); ++i) tex[i] = map ? Rgb{map[3 * i] * L[0], map[3 * i + 1] * L[1], map[3 * i + 2] * L[2]} : Rgb{L[0], L[1], L[2]};
        EnvPyramid pyr;
        pyr.build(width, height, tex);
        width = pyr.lw[0]; height = pyr.lh[0]; // power-of-two sides after MipMap::resizeImage
        w = width; h = height;
        const std::vector<Rgb>& l0 = pyr.level[0];
        texels.resize(l0.size() * 4);
        for (size_t i = 0; i < l0.size(); ++i) { texels[4 * i] = l0[i].r; texels[4 * i + 1] = l0[i].g; texels[4 * i + 2] = l0[i].b; texels[4 * i + 3] = 0.f; }
        nu = 2 * width; nv = 2 * height;
        std::vector<float> img((size_t)nu * nv);
        for (int v = 0; v < nv; ++v) {
            float filter = 0.5f / std::min(nu, nv);
            float vp = (v + 0.5f) / (float)nv;
            float sinTheta = sinf(3.1415926535897932384f * float(v + .5f) / float(nv));
            for (int u = 0; u < nu; ++u) {
                float up = float(u + .5f) / (float)nu;
                size_t idx = u + (size_t)v * nu;
                img[idx] = luminanceY(pyr.lookup(up, vp, filter));
                img[idx] *= sinTheta;
            }
        }
        condFunc.clear(); condCdf.clear(); condInt.clear(); margFunc.clear(); margCdf.clear();
        for (int v = 0; v < nv; ++v) condInt.push_back(appendDistribution1D(&img[(size_t)v * nu], nu, &condFunc, &condCdf));
        std::vector<float> mf(condInt);
        margInt = appendDistribution1D(mf.data(), nv, &margFunc, &margCdf);
        powerLookup = pyr.lookupHalf();
    }
};

} // namespace enigma
