#include "sobol_host.h"

#include <mutex>

namespace enigma {

static const uint32_t kJoeKuo[] = {
#include "joe_kuo_params.inc"
};

// 52-bit direction numbers V_k = m_k << (52 - k), k = 1..52, by the Joe-Kuo recurrence
//   m_k = 2 a_1 m_{k-1} ^ 4 a_2 m_{k-2} ^ ... ^ 2^{s-1} a_{s-1} m_{k-s+1} ^ 2^s m_{k-s} ^ m_{k-s}.
static void direction_numbers(std::vector<uint64_t>& v52) {
    v52.assign((size_t)kNumSobolDimensions * kSobolMatrixSize, 0);
    for (int k = 0; k < kSobolMatrixSize; ++k) v52[k] = 1ull << (kSobolMatrixSize - 1 - k); // van der Corput
    size_t pos = 0;
    for (int d = 1; d < kNumSobolDimensions; ++d) {
        uint32_t s = kJoeKuo[pos++], a = kJoeKuo[pos++];
        uint64_t m[kSobolMatrixSize];
        for (uint32_t i = 0; i < s; ++i) m[i] = kJoeKuo[pos++];
        for (uint32_t k = s; k < (uint32_t)kSobolMatrixSize; ++k) {
            uint64_t v = m[k - s] ^ (m[k - s] << s);
            for (uint32_t i = 1; i < s; ++i)
                if ((a >> (s - 1 - i)) & 1) v ^= m[k - i] << i;
            m[k] = v;
        }
        for (int k = 0; k < kSobolMatrixSize; ++k) v52[(size_t)d * kSobolMatrixSize + k] = m[k] << (kSobolMatrixSize - 1 - k);
    }
}

// Inverse of an n x n GF(2) matrix given by its columns (n <= 52); returns the columns of the inverse.
static void gf2_inverse(const uint64_t* cols, int n, uint64_t* invCols) {
    // rows of [A | I] as bit masks; the identity half needs n more bits -> two words
    std::vector<uint64_t> ra(n), ri(n);
    for (int i = 0; i < n; ++i) {
        uint64_t r = 0;
        for (int j = 0; j < n; ++j) if ((cols[j] >> i) & 1) r |= 1ull << j;
        ra[i] = r;
        ri[i] = 1ull << i;
    }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        while (!((ra[piv] >> c) & 1)) ++piv; // A is invertible: dimensions 1,2 form a (0,2)-sequence
        std::swap(ra[c], ra[piv]);
        std::swap(ri[c], ri[piv]);
        for (int i = 0; i < n; ++i)
            if (i != c && ((ra[i] >> c) & 1)) { ra[i] ^= ra[c]; ri[i] ^= ri[c]; }
    }
    for (int j = 0; j < n; ++j) {
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) if ((ri[i] >> j) & 1) v |= 1ull << i;
        invCols[j] = v;
    }
}

const SobolMatrices& sobol_matrices() {
    static SobolMatrices M;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<uint64_t> v52;
        direction_numbers(v52);
        M.m32.resize(v52.size());
        for (size_t i = 0; i < v52.size(); ++i) M.m32[i] = (uint32_t)(v52[i] >> 20);
        M.vdc.assign(25 * kSobolMatrixSize, 0);
        M.vdcInv.assign(26 * kSobolMatrixSize, 0);
        const uint32_t* d0 = &M.m32[0];
        const uint32_t* d1 = &M.m32[kSobolMatrixSize];
        for (int m = 1; m <= 26; ++m) {
            auto pix = [&](int col) -> uint64_t {
                return ((uint64_t)(d0[col] >> (32 - m)) << m) | (uint64_t)(d1[col] >> (32 - m));
            };
            uint64_t a[kSobolMatrixSize];
            for (int j = 0; j < 2 * m; ++j) a[j] = pix(j);
            gf2_inverse(a, 2 * m, &M.vdcInv[(size_t)(m - 1) * kSobolMatrixSize]);
            if (m <= 25)
                for (int c = 0; c < kSobolMatrixSize - 2 * m; ++c) M.vdc[(size_t)(m - 1) * kSobolMatrixSize + c] = pix(2 * m + c);
        }
    });
    return M;
}

uint64_t sobol_interval_to_index(uint32_t m, uint64_t frame, uint32_t px, uint32_t py) {
    if (m == 0) return 0;
    const SobolMatrices& M = sobol_matrices();
    const uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= M.vdc[(size_t)(m - 1) * kSobolMatrixSize + c];
    uint64_t b = (((uint64_t)px << m) | py) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= M.vdcInv[(size_t)(m - 1) * kSobolMatrixSize + c];
    return index;
}

uint32_t sobol_sample_bits(uint64_t a, int dimension) {
    const SobolMatrices& M = sobol_matrices();
    uint32_t v = 0;
    for (size_t i = (size_t)dimension * kSobolMatrixSize; a != 0; a >>= 1, i++)
        if (a & 1) v ^= M.m32[i];
    return v;
}

void build_pixel_tables(uint32_t m, uint32_t nDims, SobolPixelTables* out) {
    out->log2Res = m;
    out->nDims = nDims;
    const size_t res = (size_t)1 << m;
    out->tx.assign(nDims * res, 0);
    out->ty.assign(nDims * res, 0);
    if (m == 0) return; // the reference maps every sample of a 1x1 grid to index 0
    const SobolMatrices& M = sobol_matrices();
    const uint64_t* inv = &M.vdcInv[(size_t)(m - 1) * kSobolMatrixSize];
    for (uint32_t d = 0; d < nDims; ++d) {
        uint32_t colX[32], colY[32];
        for (uint32_t j = 0; j < m; ++j) {
            colY[j] = sobol_sample_bits(inv[j], (int)d);
            colX[j] = sobol_sample_bits(inv[m + j], (int)d);
        }
        uint32_t* tx = &out->tx[d * res];
        uint32_t* ty = &out->ty[d * res];
        for (size_t p = 1; p < res; ++p) {
            int low = __builtin_ctzll(p);
            tx[p] = tx[p & (p - 1)] ^ colX[low];
            ty[p] = ty[p & (p - 1)] ^ colY[low];
        }
    }
}

void sample_constants(uint32_t m, uint64_t sampleNum, uint32_t nDims, uint32_t* out) {
    uint64_t index0 = sobol_interval_to_index(m, sampleNum, 0, 0);
    for (uint32_t d = 0; d < nDims; ++d) out[d] = sobol_sample_bits(index0, (int)d);
}

} // namespace enigma
