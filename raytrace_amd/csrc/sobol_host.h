// Host-side Sobol' machinery for the device sampler.
//
// The reference evaluates SobolSampler::SampleDimension (Enigma/Sampler.cpp:229-240) as
// SobolSampleFloat(SobolIntervalToIndex(m, sampleNum, pixel), dim)
// (Enigma/SamplerInclude.h:2600-2637): two chains of GF(2) matrix-vector products over
// tables tabulated in Enigma/SobolMatrices.cpp.  Everything in those chains is linear over
// GF(2), so for a 2^m x 2^m pixel grid the 32-bit sample word of dimension d factors as
//
//     v_d(px, py, sampleNum) = Tx_d[px] ^ Ty_d[py] ^ c_d(sampleNum)
//
// with Tx_d/Ty_d tables of 2^m words and one constant per (sampleNum, d).  The device
// sampler is two table reads and two XORs per dimension and is bit-identical to the
// reference's bit loops.  This file regenerates the generator matrices from the Joe-Kuo
// parameters (no tabulated matrices are shipped) and builds those tables.
#pragma once
#include <cstdint>
#include <vector>

namespace enigma {

constexpr int kNumSobolDimensions = 1024; // Enigma/SobolMatrices.h:6
constexpr int kSobolMatrixSize = 52;      // Enigma/SobolMatrices.h:7

struct SobolMatrices {
    std::vector<uint32_t> m32;   // [1024][52]  == SobolMatrices32
    std::vector<uint64_t> vdc;   // [25][52]    == VdCSobolMatrices (zero padded rows)
    std::vector<uint64_t> vdcInv; // [26][52]   == VdCSobolMatricesInv (zero padded rows)
};

// Built once per process (thread safe).
const SobolMatrices& sobol_matrices();

// Enigma/SamplerInclude.h:2600-2620, restated on the regenerated tables (debug taps, host checks).
uint64_t sobol_interval_to_index(uint32_t m, uint64_t frame, uint32_t px, uint32_t py);
// Enigma/SamplerInclude.h:2627-2637 without the float conversion.
uint32_t sobol_sample_bits(uint64_t index, int dimension);

struct SobolPixelTables {
    uint32_t log2Res = 0;
    uint32_t nDims = 0;
    std::vector<uint32_t> tx; // [nDims][2^m]
    std::vector<uint32_t> ty; // [nDims][2^m]
};
// Tables for pixels relative to the sample-bounds origin on a 2^m grid.
void build_pixel_tables(uint32_t log2Res, uint32_t nDims, SobolPixelTables* out);
// c_d(sampleNum) for d in [0, nDims).
void sample_constants(uint32_t log2Res, uint64_t sampleNum, uint32_t nDims, uint32_t* out);

} // namespace enigma
