// Test shim (CPU only): exposes the product's pure host code -- the Sobol' table generator and the
// BVH builder -- so they can be checked without a GPU.
#include "../raytrace_amd/csrc/bvh_build.h"
#include "../raytrace_amd/csrc/sobol_host.h"
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
// BVH: returns node count; fills order; checks = {maxDepth, leafTrisCovered, boxViolations, badRefs}
uint32_t hc_bvh(const float* v, const uint32_t* idx, uint32_t nT, uint32_t maxLeaf, uint32_t* orderOut, uint32_t* checks) {
    BvhResult R;
    build_bvh(v, idx, nT, maxLeaf, &R);
    memcpy(orderOut, R.triOrder.data(), nT * 4);
    std::vector<int> seen(nT, 0);
    uint32_t violations = 0, bad = 0, covered = 0;
    for (size_t ni = 0; ni < R.nodes.size(); ++ni) {
        const Bvh4Node& n = R.nodes[ni];
        for (int c = 0; c < 4; ++c) {
            float lo[3] = {n.lox[c], n.loy[c], n.loz[c]}, hi[3] = {n.hix[c], n.hiy[c], n.hiz[c]};
            if (n.child[c] == kEmptyChild) { if (!(lo[0] > hi[0])) ++bad; continue; } // unused slot: inverted box
            if (n.child[c] >= 0) { if ((size_t)n.child[c] >= R.nodes.size() || (size_t)n.child[c] <= ni) ++bad; continue; }
            uint32_t leaf = ~(uint32_t)n.child[c], first = leaf >> 3, cnt = (leaf & 7) + 1;
            if (first + cnt > nT || cnt > maxLeaf) { ++bad; continue; }
            for (uint32_t k = first; k < first + cnt; ++k) {
                uint32_t t = R.triOrder[k];
                if (seen[t]++ == 0) ++covered;
                for (int vtx = 0; vtx < 3; ++vtx)
                    for (int a = 0; a < 3; ++a) {
                        float p = v[3 * (size_t)idx[3 * (size_t)t + vtx] + a];
                        if (p < lo[a] || p > hi[a]) ++violations;
                    }
            }
        }
    }
    uint32_t dup = 0;
    for (uint32_t t = 0; t < nT; ++t) if (seen[t] > 1) ++dup;
    checks[0] = R.maxDepth; checks[1] = covered; checks[2] = violations; checks[3] = bad; checks[4] = dup;
    return (uint32_t)R.nodes.size();
}
}
