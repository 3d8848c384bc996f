#include "spectrum_host.h"

#include <algorithm>
#include <utility>
#include <vector>

#include "spectral_data.inc"

namespace enigma {

namespace {

// Lerp(start, end, t), Enigma/MathCommon.h:31-39: clamped ends, start + (end - start) * t
float lerpClamped(float a, float b, float t) {
    if (t < 0.f) return a;
    if (t > 1.f) return b;
    const float diff = (b - a) * t;
    return a + diff;
}

// InterpolateSpectrumSamples, Enigma/Spectrum.cpp:297-309 (FindInterval: Enigma/Misc.h:70-83)
float interpolate(const float* lambda, const float* vals, int n, float l) {
    if (l <= lambda[0]) return vals[0];
    if (l >= lambda[n - 1]) return vals[n - 1];
    int first = 0, len = n;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (lambda[middle] <= l) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    const int offset = std::min(std::max(first - 1, 0), n - 2);
    const float t = (l - lambda[offset]) / (lambda[offset + 1] - lambda[offset]);
    return lerpClamped(vals[offset], vals[offset + 1], t);
}

} // namespace

void rgb_from_sampled(const float* lambda, const float* v, int n, float rgb[3]) {
    bool sorted = true;
    for (int i = 0; i + 1 < n; ++i) sorted = sorted && !(lambda[i] > lambda[i + 1]);
    std::vector<float> sl, sv;
    if (!sorted) { // SortSpectrumSamples: by wavelength, then by value
        std::vector<std::pair<float, float>> p;
        for (int i = 0; i < n; ++i) p.push_back({lambda[i], v[i]});
        std::sort(p.begin(), p.end());
        for (const auto& e : p) { sl.push_back(e.first); sv.push_back(e.second); }
        lambda = sl.data(); v = sv.data();
    }
    float xyz[3] = {0, 0, 0};
    for (int i = 0; i < kCieSamples; ++i) {
        const float val = interpolate(lambda, v, n, kCieLambda[i]);
        xyz[0] += val * kCieX[i];
        xyz[1] += val * kCieY[i];
        xyz[2] += val * kCieZ[i];
    }
    const float kCieYIntegral = 106.856895f; // Enigma/Spectrum.h:46
    const float scale = float(kCieLambda[kCieSamples - 1] - kCieLambda[0]) / float(kCieYIntegral * kCieSamples);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    // XYZToRGB, Enigma/Spectrum.h:22-26
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

void default_metal_spectra(float etaRgb[3], float kRgb[3]) {
    rgb_from_sampled(kCopperLambda, kCopperN, kCopperSamples, etaRgb);
    rgb_from_sampled(kCopperLambda, kCopperK, kCopperSamples, kRgb);
}

} // namespace enigma
