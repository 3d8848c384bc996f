// Structural check of the 8-wide compressed tree builder (host code, no GPU): every triangle sits in exactly one
// leaf, every stored child box contains the true box of everything below it, the slot / meta / mask encoding is
// consistent, and the depth reported bounds the real one.
//   bvh8_check <nTris> <seed> <mode>     mode 0: random soup, 1: a grid of coplanar quads (many flat nodes), 2: all coincident
#include "../../raytrace_amd/csrc/bvh_build.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

using namespace enigma;

int main(int argc, char** argv) {
    uint32_t nT = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000;
    const uint32_t seed = argc > 2 ? (uint32_t)atoi(argv[2]) : 1;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    uint32_t state = seed * 2654435761u + 12345u;
    auto rnd = [&] { state = state * 1664525u + 1013904223u; return (state >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> v;
    std::vector<uint32_t> idx;
    for (uint32_t t = 0; t < nT; ++t) {
        float c[3] = {rnd() * 1000.f - 300.f, rnd() * 10.f, rnd() * 2000.f};
        if (mode == 1) { c[0] = (float)(t % 37); c[1] = 5.f; c[2] = (float)(t / 37); }
        if (mode == 2) { c[0] = 1.f; c[1] = 2.f; c[2] = 3.f; }
        for (int k = 0; k < 3; ++k) {
            const float s = mode == 2 ? 0.f : (mode == 1 ? 1.f : 3.f * rnd());
            v.push_back(c[0] + (k == 1 ? s : 0.f) + (mode == 0 ? rnd() : 0.f));
            v.push_back(c[1] + (mode == 0 ? rnd() * s : 0.f));
            v.push_back(c[2] + (k == 2 ? s : 0.f));
            idx.push_back((uint32_t)(v.size() / 3 - 1));
        }
    }
    if (mode == 3 && argc > 4) { // a triangle soup from a file: uint32 nV, nT, float[nV*3], uint32[nT*3]
        FILE* f = fopen(argv[4], "rb");
        uint32_t hd[2];
        if (!f || fread(hd, 4, 2, f) != 2) { printf("FAIL: cannot read %s\n", argv[4]); return 1; }
        v.resize((size_t)hd[0] * 3); idx.resize((size_t)hd[1] * 3);
        if (fread(v.data(), 4, v.size(), f) != v.size() || fread(idx.data(), 4, idx.size(), f) != idx.size()) { printf("FAIL: short file\n"); return 1; }
        fclose(f);
    }
    const uint32_t nTris = (uint32_t)(idx.size() / 3);
    Bvh8Result R;
    build_bvh8(v.data(), idx.data(), nTris, &R);
    nT = nTris;
    if (R.triOrder.size() != nT) { printf("FAIL: %zu triangles in the order array\n", R.triOrder.size()); return 1; }
    std::vector<int> seen(nT, 0);
    for (uint32_t t : R.triOrder) { if (t >= nT || seen[t]++) { printf("FAIL: triangle %u repeated or out of range\n", t); return 1; } }
    uint32_t maxDepth = 0;
    size_t visited = 0;
    struct B { double lo[3], hi[3]; };
    // returns the true box of the subtree
    std::function<B(uint32_t, uint32_t)> walk = [&](uint32_t ni, uint32_t depth) -> B {
        if (ni >= R.nodes.size()) { printf("FAIL: node index %u out of range\n", ni); exit(1); }
        ++visited;
        maxDepth = std::max(maxDepth, depth);
        const Bvh8Node& n = R.nodes[ni];
        B all{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
        uint32_t inner = 0, triOff = 0;
        for (int s = 0; s < 8; ++s) {
            const uint8_t m = n.meta[s];
            const bool isInner = (n.imask >> s) & 1;
            if (m == 0) { if (isInner) { printf("FAIL: empty slot flagged inner\n"); exit(1); } continue; }
            B cb{{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
            if (isInner) {
                if (m != (0x20 | (24 + s))) { printf("FAIL: inner meta %02x in slot %d\n", m, s); exit(1); }
                cb = walk(n.childBase + inner, depth + 1);
                ++inner;
            } else {
                const uint32_t cnt = (m >> 5) == 1 ? 1 : ((m >> 5) == 3 ? 2 : ((m >> 5) == 7 ? 3 : 0)), off = m & 31u;
                if (!cnt || off != triOff || off + cnt > 24) { printf("FAIL: leaf meta %02x (offset %u expected %u)\n", m, off, triOff); exit(1); }
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint32_t t = R.triOrder[n.triBase + off + i];
                    for (int k = 0; k < 3; ++k)
                        for (int a = 0; a < 3; ++a) {
                            const double x = v[3 * (size_t)idx[3 * (size_t)t + k] + a];
                            cb.lo[a] = std::min(cb.lo[a], x); cb.hi[a] = std::max(cb.hi[a], x);
                        }
                }
                triOff += cnt;
            }
            for (int a = 0; a < 3; ++a) {
                const double scale = std::ldexp(1.0, (int)n.e[a] - 127);
                const double lo = (double)n.lo[a] + n.qlo[a][s] * scale, hi = (double)n.lo[a] + n.qhi[a][s] * scale;
                if (lo > cb.lo[a] || hi < cb.hi[a]) { printf("FAIL: node %u slot %d axis %d: stored [%.17g, %.17g] does not hold [%.17g, %.17g]\n", ni, s, a, lo, hi, cb.lo[a], cb.hi[a]); exit(1); }
                all.lo[a] = std::min(all.lo[a], cb.lo[a]); all.hi[a] = std::max(all.hi[a], cb.hi[a]);
            }
        }
        return all;
    };
    if (nT) walk(0, 1);
    if (nT && visited != R.nodes.size()) { printf("FAIL: %zu of %zu nodes reachable\n", visited, R.nodes.size()); return 1; }
    if (maxDepth > R.maxDepth) { printf("FAIL: depth %u > reported %u\n", maxDepth, R.maxDepth); return 1; }
    size_t kids = 0, leaves = 0, inner = 0;
    for (const Bvh8Node& n : R.nodes)
        for (int s = 0; s < 8; ++s) if (n.meta[s]) { ++kids; if ((n.imask >> s) & 1) ++inner; else ++leaves; }
    printf("OK %u triangles %zu nodes depth %u: %.2f children per node, %.2f triangles per leaf\n", nT, R.nodes.size(), R.maxDepth,
           (double)kids / R.nodes.size(), leaves ? (double)nT / leaves : 0.0);
    return 0;
}
