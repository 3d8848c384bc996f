#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace enigma {

namespace {

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* p) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    // in double: the extents of a scene that reaches out to 2^60 square to 2^122, and the cost sums below add many of those
    double area() const {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        if (dx < 0) return 0.0;
        return 2.0 * (dx * dy + dx * dz + dy * dz);
    }
};

// the bin of a centroid at offset * scale along an axis: in range whatever the product is (NaN and negatives go to bin 0)
inline int binOf(float offsetTimesScale, int bins) { return offsetTimesScale >= 0.f ? (offsetTimesScale < (float)bins ? (int)offsetTimesScale : bins - 1) : 0; }

struct TmpNode {
    Box box;
    int left = -1, right = -1; // children (TmpNode indices) or -1
    uint32_t first = 0, count = 0; // leaf range in the order array
};

struct Builder {
    const std::vector<Box>& triBox;
    const std::vector<float>& centroid; // [nT*3]
    std::vector<uint32_t>& order;
    std::vector<TmpNode> nodes;
    uint32_t maxLeaf;
    static constexpr int kMaxBins = 64;
    int kBins = 16;
    // Below this depth the surface-area splits stop and ranges are halved by count, which ends within log2(count) more
    // levels: a tree over up to 2^28 primitives is at most 64 levels deep whatever the geometry, the depth the traversal
    // stacks are sized for.
    static constexpr uint32_t kSahDepth = 36;

    Builder(const std::vector<Box>& tb, const std::vector<float>& c, std::vector<uint32_t>& o, uint32_t ml)
        : triBox(tb), centroid(c), order(o), maxLeaf(ml) {}

    int build(uint32_t first, uint32_t count, uint32_t depth = 0) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        Box box, cbox;
        for (uint32_t i = first; i < first + count; ++i) {
            box.grow(triBox[order[i]]);
            cbox.grow(&centroid[3 * (size_t)order[i]]);
        }
        nodes[id].box = box;
        auto makeLeaf = [&]() { nodes[id].first = first; nodes[id].count = count; return id; };
        if (count == 1) return makeLeaf();
        int axis = 0;
        float ext[3] = {cbox.hi[0] - cbox.lo[0], cbox.hi[1] - cbox.lo[1], cbox.hi[2] - cbox.lo[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        uint32_t mid = first + count / 2;
        if (ext[axis] <= 0.f) {
            if (count <= maxLeaf) return makeLeaf();
            // coincident centroids: split by count
        } else if (depth >= kSahDepth) {
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) {
                return centroid[3 * (size_t)x + axis] < centroid[3 * (size_t)y + axis];
            });
        } else {
            double bestCost = HUGE_VAL;
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                if (ext[a] <= 0.f) continue;
                Box bb[kMaxBins];
                uint32_t bc[kMaxBins] = {0};
                float scale = kBins / ext[a];
                for (uint32_t i = first; i < first + count; ++i) {
                    uint32_t t = order[i];
                    int b = binOf((centroid[3 * (size_t)t + a] - cbox.lo[a]) * scale, kBins);
                    bb[b].grow(triBox[t]);
                    bc[b]++;
                }
                double rightArea[kMaxBins];
                uint32_t rightCount[kMaxBins];
                Box acc;
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    rightArea[b] = acc.area(); rightCount[b] = cnt;
                }
                Box accL;
                uint32_t cntL = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    accL.grow(bb[b]); cntL += bc[b];
                    if (cntL == 0 || rightCount[b + 1] == 0) continue;
                    double cost = accL.area() * cntL + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b; }
                }
            }
            double parentArea = std::max(box.area(), 1e-30);
            double splitCost = 1.0 + bestCost / parentArea;
            if (count <= maxLeaf && (bestAxis < 0 || (double)count <= splitCost)) return makeLeaf();
            if (bestAxis >= 0) {
                float scale = kBins / ext[bestAxis];
                float lo = cbox.lo[bestAxis];
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                    int b = binOf((centroid[3 * (size_t)t + bestAxis] - lo) * scale, kBins);
                    return b <= bestBin;
                });
                mid = (uint32_t)(it - order.begin());
            }
            if (mid == first || mid == first + count) {
                mid = first + count / 2;
                std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) {
                    return centroid[3 * (size_t)x + axis] < centroid[3 * (size_t)y + axis];
                });
            }
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l;
        nodes[id].right = r;
        nodes[id].first = first; // an inner node's subtree covers one range of the order array too
        nodes[id].count = count;
        return id;
    }
};

} // namespace

// ---------------------------------------------------------------- 8-wide compressed tree
namespace {

// smallest biased exponent e with 255 * 2^(e - 127) >= extent (so that every offset fits a byte after rounding up)
uint8_t exponentFor(float extent) {
    if (!(extent > 0.f)) return 1; // a flat node: any scale does; 2^-126 keeps the products finite
    int ex;
    std::frexp(extent / 255.0f, &ex); // extent / 255 = m * 2^ex, m in [0.5, 1)  =>  2^ex >= extent / 255
    int e = ex + 127;
    // rounding in the division above may leave 255 * 2^(e-127) a hair short: check with the exact products used below
    while (e < 254 && std::ldexp(255.0f, e - 127) < extent) ++e;
    return (uint8_t)std::min(254, std::max(1, e));
}

} // namespace

void build_bvh8(const float* vertices, const uint32_t* indices, uint32_t nTriangles, const float* boxes, uint32_t nBoxes, Bvh8Result* out) {
    const uint32_t nTris = nTriangles + nBoxes; // primitives of either kind from here on
    std::vector<Box> triBox(nTris);
    std::vector<float> centroid((size_t)nTris * 3);
    for (uint32_t t = 0; t < nTris; ++t) {
        if (t < nTriangles) for (int k = 0; k < 3; ++k) triBox[t].grow(&vertices[3 * (size_t)indices[3 * (size_t)t + k]]);
        else { triBox[t].grow(&boxes[6 * (size_t)(t - nTriangles)]); triBox[t].grow(&boxes[6 * (size_t)(t - nTriangles) + 3]); }
        for (int a = 0; a < 3; ++a) centroid[3 * (size_t)t + a] = 0.5f * triBox[t].lo[a] + 0.5f * triBox[t].hi[a];
    }
    std::vector<uint32_t> order(nTris);
    for (uint32_t t = 0; t < nTris; ++t) order[t] = t;
    out->nodes.clear();
    out->triOrder.clear();
    out->triOrder.reserve(nTris);
    out->maxDepth = 1;
    for (int a = 0; a < 3; ++a) { out->rootLo[a] = 0; out->rootHi[a] = 0; }
    auto emptyNode = [] { Bvh8Node n; memset(&n, 0, sizeof(n)); n.e[0] = n.e[1] = n.e[2] = 1; return n; };
    if (nTris == 0) { out->nodes.push_back(emptyNode()); return; }
    Builder B(triBox, centroid, order, 1); // down to single triangles: the collapse below decides what a leaf is
    if (const char* e = getenv("ENIGMA_BVH_BINS")) B.kBins = std::max(4, std::min(64, atoi(e)));
    B.nodes.reserve((size_t)nTris * 2);
    const int root = B.build(0, nTris);
    for (int a = 0; a < 3; ++a) { out->rootLo[a] = B.nodes[root].box.lo[a]; out->rootHi[a] = B.nodes[root].box.hi[a]; }

    // Which binary nodes become 8-wide nodes, which subtrees become leaves: the collapse of least surface-area cost
    // (Ylitie, Karras, Laine 2017, section 4.1).  cost[n][i] = cheapest way to stand for the subtree of n with at most
    // i roots (wide nodes or leaves), i = 1..7:
    //   cost[n][1] = min(leaf, wide node),  leaf = area * triangles * cPrim (at most 3 triangles),
    //                                       wide node = area * cNode + distribute(n, 8)
    //   cost[n][i] = min(distribute(n, i), cost[n][i - 1]),  distribute(n, j) = min over k of cost[left][k] + cost[right][j - k]
    // Binary nodes are numbered parent before children, so one backward sweep sees children first.
    const double cNode = 1.0;
    double cPrim = 0.6; // a node visit against one triangle test, by their instruction counts
    if (const char* e = getenv("ENIGMA_BVH_CPRIM")) cPrim = atof(e);
    const size_t nb = B.nodes.size();
    std::vector<double> cost(nb * 8, 0.0);       // [n][i], i = 1..7 (slot 0 unused)
    std::vector<uint8_t> splitK(nb * 9, 0);      // [n][j], j = 2..8: the k of distribute(n, j)
    std::vector<uint8_t> takesDist(nb * 8, 0);   // [n][i], i = 2..7: cost[n][i] is distribute(n, i), not cost[n][i - 1]
    std::vector<uint8_t> asLeaf(nb, 0);
    for (size_t ni = nb; ni-- > 0;) {
        const TmpNode& n = B.nodes[ni];
        const double area = n.box.area();
        const double leafCost = n.count <= 3 ? area * (double)n.count * cPrim : HUGE_VAL;
        if (n.left < 0) { // a single triangle
            for (int i = 1; i <= 7; ++i) cost[ni * 8 + i] = leafCost;
            asLeaf[ni] = 1;
            continue;
        }
        const double* cl = &cost[(size_t)n.left * 8];
        const double* cr = &cost[(size_t)n.right * 8];
        double dist[9];
        for (int j = 2; j <= 8; ++j) {
            double best = HUGE_VAL;
            int bk = 1;
            for (int k = std::max(1, j - 7); k <= std::min(7, j - 1); ++k) {
                const double c = cl[k] + cr[j - k];
                if (c < best) { best = c; bk = k; }
            }
            dist[j] = best;
            splitK[ni * 9 + j] = (uint8_t)bk;
        }
        const double wideCost = area * cNode + dist[8];
        asLeaf[ni] = leafCost <= wideCost;
        cost[ni * 8 + 1] = std::min(leafCost, wideCost);
        for (int i = 2; i <= 7; ++i) {
            takesDist[ni * 8 + i] = dist[i] < cost[ni * 8 + i - 1];
            cost[ni * 8 + i] = std::min(dist[i], cost[ni * 8 + i - 1]);
        }
    }
    // the roots that stand for subtree n when it may use up to i of them
    std::function<void(int, int, int*, int*)> expand = [&](int n, int i, int* kids, int* nk) {
        while (i > 1 && !takesDist[(size_t)n * 8 + i]) --i;
        if (i == 1 || B.nodes[n].left < 0) { kids[(*nk)++] = n; return; }
        const int k = splitK[(size_t)n * 9 + i];
        expand(B.nodes[n].left, k, kids, nk);
        expand(B.nodes[n].right, i - k, kids, nk);
    };
    auto isLeaf = [&](int tmp) { return asLeaf[tmp] != 0; };

    struct Item { int tmp; uint32_t outIndex; uint32_t depth; };
    std::vector<Item> queue; // breadth first: the children of a node are consecutive, siblings' subtrees follow level by level
    out->nodes.reserve(nb / 8 + 16);
    out->nodes.push_back(emptyNode());
    queue.push_back({root, 0u, 1u});
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const Item it = queue[qi];
        out->maxDepth = std::max(out->maxDepth, it.depth);
        int kids[8];
        int nk = 0;
        if (isLeaf(it.tmp)) kids[nk++] = it.tmp; // the whole scene is one leaf
        else {
            const int k = splitK[(size_t)it.tmp * 9 + 8];
            expand(B.nodes[it.tmp].left, k, kids, &nk);
            expand(B.nodes[it.tmp].right, 8 - k, kids, &nk);
        }
        // a node's leaves hold at most 24 triangles (24 mask bits): with eight leaves of three all fit
        const Box& nbox = B.nodes[it.tmp].box;
        // slots: greedy assignment of children to the slot whose corner direction they lie furthest towards
        float ctr[3];
        for (int a = 0; a < 3; ++a) ctr[a] = 0.5f * nbox.lo[a] + 0.5f * nbox.hi[a];
        float score[8][8];
        for (int c = 0; c < nk; ++c) {
            const Box& kb = B.nodes[kids[c]].box;
            float rel[3];
            for (int a = 0; a < 3; ++a) rel[a] = (0.5f * kb.lo[a] + 0.5f * kb.hi[a]) - ctr[a];
            for (int s = 0; s < 8; ++s) score[c][s] = ((s & 1) ? -rel[0] : rel[0]) + ((s & 2) ? -rel[1] : rel[1]) + ((s & 4) ? -rel[2] : rel[2]);
        }
        int slotOf[8], kidAt[8];
        for (int i = 0; i < 8; ++i) { slotOf[i] = -1; kidAt[i] = -1; }
        for (int round = 0; round < nk; ++round) {
            int bc = -1, bs = -1;
            float bestScore = -FLT_MAX;
            for (int c = 0; c < nk; ++c) {
                if (slotOf[c] >= 0) continue;
                for (int s = 0; s < 8; ++s)
                    if (kidAt[s] < 0 && score[c][s] > bestScore) { bestScore = score[c][s]; bc = c; bs = s; }
            }
            slotOf[bc] = bs; kidAt[bs] = bc;
        }
        Bvh8Node n = emptyNode();
        for (int a = 0; a < 3; ++a) { n.lo[a] = nbox.lo[a]; n.e[a] = exponentFor((nbox.hi[a] - nbox.lo[a]) * 1.0001f); }
        n.childBase = (uint32_t)out->nodes.size();
        n.triBase = (uint32_t)out->triOrder.size();
        uint32_t nInner = 0, triOff = 0;
        for (int s = 0; s < 8; ++s) {
            if (kidAt[s] < 0) continue;
            const int tmp = kids[kidAt[s]];
            const TmpNode& k = B.nodes[tmp];
            for (int a = 0; a < 3; ++a) {
                const float scale = std::ldexp(1.0f, (int)n.e[a] - 127);
                // offsets rounded outwards; the comparisons repair the last-ulp cases of the division
                int ql = (int)std::floor((k.box.lo[a] - n.lo[a]) / scale), qh = (int)std::ceil((k.box.hi[a] - n.lo[a]) / scale);
                ql = std::min(255, std::max(0, ql)); qh = std::min(255, std::max(0, qh));
                while (ql > 0 && (double)n.lo[a] + (double)ql * (double)scale > (double)k.box.lo[a]) --ql;
                while (qh < 255 && (double)n.lo[a] + (double)qh * (double)scale < (double)k.box.hi[a]) ++qh;
                n.qlo[a][s] = (uint8_t)ql; n.qhi[a][s] = (uint8_t)qh;
            }
            if (!isLeaf(tmp)) {
                n.imask |= (uint8_t)(1u << s);
                n.meta[s] = (uint8_t)(0x20u | (24u + (uint32_t)s));
                queue.push_back({tmp, n.childBase + nInner, it.depth + 1});
                out->nodes.push_back(emptyNode());
                ++nInner;
            } else {
                const uint32_t unary = k.count == 1 ? 0x20u : (k.count == 2 ? 0x60u : 0xe0u);
                n.meta[s] = (uint8_t)(unary | triOff);
                for (uint32_t i = 0; i < k.count; ++i) out->triOrder.push_back(order[k.first + i]);
                triOff += k.count;
            }
        }
        out->nodes[it.outIndex] = n;
    }
}

} // namespace enigma
