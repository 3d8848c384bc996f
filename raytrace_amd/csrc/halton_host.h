// Host side of the HaltonSampler (Enigma/Sampler.cpp:305-375): what the device needs to turn (pixel, sample number,
// dimension) into a sample value without the sampler object.
//
//   * the digit permutations of ScrambledRadicalInverse, made as the reference makes them: the identity of every prime
//     base shuffled by Shuffle (Enigma/MonteCarlo.h:140-147) with the words of a default-seeded std::mt19937, taken
//     through the rejection loop of RNG::randomUInt(b) (Enigma/RNG.cpp:30-38; Enigma/SamplerInclude.h:2639-2655);
//   * the index offset of a pixel (GetIndexForSample, Enigma/Sampler.cpp:336-356) split into a term per column and a
//     term per row, each reduced modulo the sample stride, so that the device adds two table entries;
//   * per dimension the prime base, where its permutation starts, the two numbers that turn a division by the base into
//     a multiplication, and the two float constants of the base that the radical inverse uses.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

namespace enigma {

constexpr int kHaltonDimensions = 1000;   // PrimeTableSize, Enigma/Misc.h:28
constexpr int kHaltonMaxResolution = 128; // kMaxResolution, Enigma/SamplerInclude.h:25

// the first 1000 primes (Primes, Enigma/SamplerInclude.h:2397) by a sieve; 7919 is the 1000th
inline const std::vector<uint32_t>& halton_primes() {
    static const std::vector<uint32_t> primes = [] {
        const uint32_t limit = 7920;
        std::vector<bool> composite(limit, false);
        std::vector<uint32_t> out;
        for (uint32_t c = 2; c < limit; ++c) {
            if (composite[c]) continue;
            out.push_back(c);
            for (uint32_t m = c * c; m < limit; m += c) composite[m] = true;
        }
        return out;
    }();
    return primes;
}

// permutation of base i at [sum of the primes before it, + prime): PrimeSums, Enigma/SamplerInclude.h:2481
inline const std::vector<uint16_t>& halton_permutations(std::vector<uint32_t>* startsOut = nullptr) {
    static std::vector<uint32_t> starts;
    static const std::vector<uint16_t> table = [] {
        const std::vector<uint32_t>& P = halton_primes();
        std::mt19937 engine; // RNG's generator is never seeded (Enigma/RNG.cpp:18-19): the standard's default seed
        auto below = [&](uint32_t b) { // RNG::randomUInt(b)
            const uint32_t threshold = (0u - b) % b;
            uint32_t r;
            do r = (uint32_t)engine(); while (r < threshold);
            return r % b;
        };
        std::vector<uint16_t> t;
        starts.resize(kHaltonDimensions);
        for (int i = 0; i < kHaltonDimensions; ++i) {
            const uint32_t base = P[i], first = (uint32_t)t.size();
            starts[i] = first;
            for (uint32_t d = 0; d < base; ++d) t.push_back((uint16_t)d);
            for (uint32_t d = 0; d < base; ++d) std::swap(t[first + d], t[first + d + below(base - d)]);
        }
        return t;
    }();
    if (startsOut) *startsOut = starts;
    return table;
}

// Per dimension the device holds: the prime, the word offset of its permutation, magic and shift of the division
// (floor(n / d) = (t + ((n - t) >> 1)) >> shift with t = the high word of magic * n, for every 32-bit n: Granlund and
// Montgomery's division by an invariant integer with l = ceil(log2 d), magic = floor(2^32 (2^l - d) / d) + 1, shift = l - 1),
// 1 / base as the reference rounds it (SamplerInclude.h:51,68), and invBase * perm[0] / (1 - invBase) (:82).
constexpr uint32_t kHaltonRecord = 6;
inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

struct HaltonTables {
    uint32_t scale[2] = {1, 1}, exponent[2] = {0, 0}, stride = 1, inverse[2] = {0, 0};
    std::vector<uint32_t> offX, offY; // per column / row of the sample bounds: its term of the index offset, modulo the stride
    std::vector<uint32_t> words;      // device block: {stride, exponent[0], scale[1], sampleAtCenter}, kHaltonRecord words per dimension, permutations
};

// a^-1 modulo n for coprime a, n (multiplicativeInverse, Enigma/SamplerInclude.h:28-45), by the iterative Euclid
inline uint32_t halton_inverse_mod(uint32_t a, uint32_t n) {
    int64_t r0 = n, r1 = a % n, t0 = 0, t1 = 1;
    while (r1 != 0) {
        int64_t q = r0 / r1;
        int64_t r2 = r0 - q * r1; r0 = r1; r1 = r2;
        int64_t t2 = t0 - q * t1; t0 = t1; t1 = t2;
    }
    return (uint32_t)((t0 % (int64_t)n + (int64_t)n) % (int64_t)n);
}

// the digits of `value` (nDigits of them, least significant first) written in reverse: InverseRadicalInverse, Enigma/Misc.h:55-64
inline uint64_t halton_reverse_digits(uint32_t base, uint64_t value, uint32_t nDigits) {
    uint64_t out = 0;
    for (uint32_t i = 0; i < nDigits; ++i) { out = out * base + value % base; value /= base; }
    return out;
}

inline uint32_t halton_wrap(int v, int m) { int r = v % m; return (uint32_t)(r < 0 ? r + m : r); } // Mod, Enigma/MathCommon.h:126-131

inline void build_halton_tables(const int32_t sb[4], uint32_t nDims, bool sampleAtCenter, HaltonTables* T) {
    const int extent[2] = {sb[2] - sb[0], sb[3] - sb[1]};
    for (int i = 0; i < 2; ++i) { // constructor, Enigma/Sampler.cpp:316-326
        const uint32_t base = i == 0 ? 2u : 3u, want = (uint32_t)std::min(extent[i], kHaltonMaxResolution);
        T->scale[i] = 1; T->exponent[i] = 0;
        while (T->scale[i] < want) { T->scale[i] *= base; ++T->exponent[i]; }
    }
    T->stride = T->scale[0] * T->scale[1];
    T->inverse[0] = T->scale[0] > 1 ? halton_inverse_mod(T->scale[1], T->scale[0]) : 0;
    T->inverse[1] = T->scale[1] > 1 ? halton_inverse_mod(T->scale[0], T->scale[1]) : 0;
    auto term = [&](int axis, int pixel) -> uint32_t { // one summand of Enigma/Sampler.cpp:344-350
        if (T->stride <= 1) return 0u;
        uint64_t digits = halton_reverse_digits(axis == 0 ? 2u : 3u, halton_wrap(pixel, kHaltonMaxResolution), T->exponent[axis]);
        return (uint32_t)((digits * (T->stride / T->scale[axis]) * T->inverse[axis]) % T->stride);
    };
    T->offX.resize((size_t)extent[0]); T->offY.resize((size_t)extent[1]);
    for (int x = 0; x < extent[0]; ++x) T->offX[(size_t)x] = term(0, sb[0] + x);
    for (int y = 0; y < extent[1]; ++y) T->offY[(size_t)y] = term(1, sb[1] + y);
    // a kernel reads the samples of up to 5 dimensions past its cursor before the cursor is checked against anything:
    // the per-dimension records run 16 dimensions further than asked for (within the prime table)
    const uint32_t dimsHeld = std::min<uint32_t>(nDims + 16u, (uint32_t)kHaltonDimensions);
    std::vector<uint32_t> starts;
    const std::vector<uint16_t>& perms = halton_permutations(&starts);
    const std::vector<uint32_t>& P = halton_primes();
    const uint32_t R = kHaltonRecord;
    T->words.assign(4 + R * (size_t)(nDims + 16u), 0u);
    T->words[0] = T->stride; T->words[1] = T->exponent[0]; T->words[2] = T->scale[1]; T->words[3] = sampleAtCenter ? 1u : 0u;
    for (uint32_t d = 0; d < nDims + 16u; ++d) {
        const uint32_t use = std::min(d, dimsHeld - 1); // padding dimensions repeat the last real one; their values are never used
        const uint32_t base = P[use];
        uint32_t l = 0;
        while ((1ull << l) < base) ++l;
        uint32_t* rec = &T->words[4 + R * (size_t)d];
        rec[0] = base;
        rec[2] = (uint32_t)(((1ull << 32) * ((1ull << l) - base)) / base + 1);
        rec[3] = l - 1;
        const float invBase = (float)1 / (float)base;
        rec[4] = float_bits(invBase);
        rec[5] = float_bits(invBase * (float)perms[starts[use]] / (1 - invBase));
    }
    for (uint32_t d = 0; d < dimsHeld; ++d) {
        const uint32_t at = (uint32_t)T->words.size();
        for (uint32_t k = 0; k < P[d]; ++k) T->words.push_back(perms[starts[d] + k]);
        T->words[4 + R * (size_t)d + 1] = at;
    }
    for (uint32_t d = dimsHeld; d < nDims + 16u; ++d) T->words[4 + R * (size_t)d + 1] = T->words[4 + R * (size_t)(dimsHeld - 1) + 1];
}

// GetIndexForSample for a pixel given relative to the sample-bounds origin
inline uint64_t halton_index(const HaltonTables& T, uint32_t relx, uint32_t rely, uint64_t sampleNum) {
    return (uint64_t)((T.offX[relx] + T.offY[rely]) % T.stride) + sampleNum * T.stride;
}

} // namespace enigma
