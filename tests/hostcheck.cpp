// Test shim (CPU only): exposes the product's pure host code -- the Sobol' table generator and the
// BVH builder -- so they can be checked without a GPU.
#include "../raytrace_amd/csrc/bvh_build.h"
#include "../raytrace_amd/csrc/sobol_host.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
using namespace enigma;

extern "C" {
void hc_sobol_tables(uint32_t* m32, uint64_t* vdc, uint64_t* vdcInv) {
    const SobolMatrices& M = sobol_matrices();
    memcpy(m32, M.m32.data(), M.m32.size() * 4);
    memcpy(vdc, M.vdc.data(), M.vdc.size() * 8);
    memcpy(vdcInv, M.vdcInv.data(), M.vdcInv.size() * 8);
}
uint64_t hc_index(uint32_t m, uint64_t frame, uint32_t px, uint32_t py) { return sobol_interval_to_index(m, frame, px, py); }
uint32_t hc_bits(uint64_t index, int dim) { return sobol_sample_bits(index, dim); }
// v_d(px,py,sampleNum) through the factored tables: Tx_d[px] ^ Ty_d[py] ^ c_d(sampleNum)
uint32_t hc_factored(uint32_t m, uint32_t nDims, uint32_t n, const uint32_t* px, const uint32_t* py, const uint64_t* sn, uint32_t* out) {
    SobolPixelTables T;
    build_pixel_tables(m, nDims, &T);
    size_t res = (size_t)1 << m;
    std::vector<uint32_t> c(nDims);
    for (uint32_t i = 0; i < n; ++i) {
        sample_constants(m, sn[i], nDims, c.data());
        for (uint32_t d = 0; d < nDims; ++d) out[(size_t)i * nDims + d] = T.tx[d * res + px[i]] ^ T.ty[d * res + py[i]] ^ c[d];
    }
    return 0;
}
// The 8-wide tree over nT triangles and nB boxes ([nB*6] lo, hi): returns the node count; fills order ([nT + nB]);
// checks = {maxDepth, primitives found in leaves, stored boxes that do not hold what lies below them, bad encodings
// (slot / meta / mask / index / depth), primitives found more than once}
uint32_t hc_bvh(const float* v, const uint32_t* idx, uint32_t nT, const float* boxes, uint32_t nB, uint32_t* orderOut, uint32_t* checks) {
    Bvh8Result R;
    build_bvh8(v, idx, nT, boxes, nB, &R);
    const uint32_t nP = nT + nB;
    uint32_t violations = 0, bad = 0, covered = 0, dup = 0, maxDepth = 0;
    if (R.triOrder.size() != nP) ++bad;
    else memcpy(orderOut, R.triOrder.data(), (size_t)nP * 4);
    std::vector<int> seen(nP, 0);
    struct B { double lo[3], hi[3]; };
    const B none{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
    std::vector<char> reached(R.nodes.size(), 0);
    // true box of the subtree of node ni (explicit recursion: at most 64 levels)
    struct Walk {
        const Bvh8Result& R; const float* v; const uint32_t* idx; const float* boxes; uint32_t nT, nP;
        std::vector<int>& seen; std::vector<char>& reached; uint32_t &violations, &bad, &covered, &dup, &maxDepth; const B& none;
        B operator()(uint32_t ni, uint32_t depth) {
            if (ni >= R.nodes.size() || depth > 64 || reached[ni]) { ++bad; return none; }
            reached[ni] = 1;
            maxDepth = std::max(maxDepth, depth);
            const Bvh8Node& n = R.nodes[ni];
            B all = none;
            uint32_t inner = 0, triOff = 0;
            for (int s = 0; s < 8; ++s) {
                const uint8_t m = n.meta[s];
                const bool isInner = (n.imask >> s) & 1;
                if (m == 0) { if (isInner) ++bad; continue; }
                B cb = none;
                if (isInner) {
                    if (m != (0x20 | (24 + s))) ++bad;
                    cb = (*this)(n.childBase + inner, depth + 1);
                    ++inner;
                } else {
                    const uint32_t cnt = (m >> 5) == 1 ? 1 : ((m >> 5) == 3 ? 2 : ((m >> 5) == 7 ? 3 : 0)), off = m & 31u;
                    if (!cnt || off != triOff || off + cnt > 24 || (size_t)n.triBase + off + cnt > R.triOrder.size()) { ++bad; continue; }
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const uint32_t t = R.triOrder[n.triBase + off + i];
                        if (t >= nP) { ++bad; continue; }
                        if (seen[t]++ == 0) ++covered; else ++dup;
                        for (int k = 0; k < (t < nT ? 3 : 2); ++k)
                            for (int a = 0; a < 3; ++a) {
                                const double x = t < nT ? v[3 * (size_t)idx[3 * (size_t)t + k] + a] : boxes[6 * (size_t)(t - nT) + 3 * k + a];
                                cb.lo[a] = std::min(cb.lo[a], x); cb.hi[a] = std::max(cb.hi[a], x);
                            }
                    }
                    triOff += cnt;
                }
                for (int a = 0; a < 3; ++a) {
                    const double scale = std::ldexp(1.0, (int)n.e[a] - 127);
                    const double lo = (double)n.lo[a] + n.qlo[a][s] * scale, hi = (double)n.lo[a] + n.qhi[a][s] * scale;
                    if (cb.lo[a] <= cb.hi[a] && (lo > cb.lo[a] || hi < cb.hi[a])) ++violations;
                    all.lo[a] = std::min(all.lo[a], cb.lo[a]); all.hi[a] = std::max(all.hi[a], cb.hi[a]);
                }
            }
            return all;
        }
    } walk{R, v, idx, boxes, nT, nP, seen, reached, violations, bad, covered, dup, maxDepth, none};
    if (!R.nodes.empty()) walk(0, 1);
    for (char c : reached) if (!c) ++bad; // every node hangs in the tree
    if (maxDepth > R.maxDepth) ++bad;
    checks[0] = R.maxDepth; checks[1] = covered; checks[2] = violations; checks[3] = bad; checks[4] = dup;
    return (uint32_t)R.nodes.size();
}
}
