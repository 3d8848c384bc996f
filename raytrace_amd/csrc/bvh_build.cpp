#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

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
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0) return 0.f;
        return 2.f * (dx * dy + dx * dz + dy * dz);
    }
};

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
    static constexpr int kBins = 16;

    Builder(const std::vector<Box>& tb, const std::vector<float>& c, std::vector<uint32_t>& o, uint32_t ml)
        : triBox(tb), centroid(c), order(o), maxLeaf(ml) {}

    int build(uint32_t first, uint32_t count) {
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
        } else {
            float bestCost = FLT_MAX;
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                if (ext[a] <= 0.f) continue;
                Box bb[kBins];
                uint32_t bc[kBins] = {0};
                float scale = kBins / ext[a];
                for (uint32_t i = first; i < first + count; ++i) {
                    uint32_t t = order[i];
                    int b = std::min(kBins - 1, (int)((centroid[3 * (size_t)t + a] - cbox.lo[a]) * scale));
                    bb[b].grow(triBox[t]);
                    bc[b]++;
                }
                float rightArea[kBins];
                uint32_t rightCount[kBins];
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
                    float cost = accL.area() * cntL + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b; }
                }
            }
            float parentArea = std::max(box.area(), 1e-30f);
            float splitCost = 1.0f + bestCost / parentArea;
            if (count <= maxLeaf && (bestAxis < 0 || (float)count <= splitCost)) return makeLeaf();
            if (bestAxis >= 0) {
                float scale = kBins / ext[bestAxis];
                float lo = cbox.lo[bestAxis];
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                    int b = std::min(kBins - 1, (int)((centroid[3 * (size_t)t + bestAxis] - lo) * scale));
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
        int l = build(first, mid - first);
        int r = build(mid, first + count - mid);
        nodes[id].left = l;
        nodes[id].right = r;
        return id;
    }
};

void clearNode(Bvh4Node& n) {
    memset(&n, 0, sizeof(n));
    for (int c = 0; c < 4; ++c) {
        n.lox[c] = n.loy[c] = n.loz[c] = FLT_MAX;
        n.hix[c] = n.hiy[c] = n.hiz[c] = -FLT_MAX;
        n.child[c] = kEmptyChild;
    }
}
void setChildBox(Bvh4Node& n, int c, const Box& b) {
    n.lox[c] = b.lo[0]; n.loy[c] = b.lo[1]; n.loz[c] = b.lo[2];
    n.hix[c] = b.hi[0]; n.hiy[c] = b.hi[1]; n.hiz[c] = b.hi[2];
}

} // namespace

void build_bvh(const float* vertices, const uint32_t* indices, uint32_t nTris, uint32_t maxLeaf, BvhResult* out) {
    maxLeaf = std::max(1u, std::min(8u, maxLeaf));
    std::vector<Box> triBox(nTris);
    std::vector<float> centroid((size_t)nTris * 3);
    for (uint32_t t = 0; t < nTris; ++t) {
        for (int k = 0; k < 3; ++k) triBox[t].grow(&vertices[3 * (size_t)indices[3 * (size_t)t + k]]);
        for (int a = 0; a < 3; ++a) centroid[3 * (size_t)t + a] = 0.5f * triBox[t].lo[a] + 0.5f * triBox[t].hi[a];
    }
    out->triOrder.resize(nTris);
    for (uint32_t t = 0; t < nTris; ++t) out->triOrder[t] = t;
    out->nodes.clear();
    out->maxDepth = 1;
    out->stackNeed = 3;
    if (nTris == 0) {
        Bvh4Node n;
        clearNode(n);
        out->nodes.push_back(n);
        for (int a = 0; a < 3; ++a) { out->rootLo[a] = 0; out->rootHi[a] = 0; }
        return;
    }
    Builder B(triBox, centroid, out->triOrder, maxLeaf);
    B.nodes.reserve((size_t)nTris * 2);
    int root = B.build(0, nTris);
    for (int a = 0; a < 3; ++a) { out->rootLo[a] = B.nodes[root].box.lo[a]; out->rootHi[a] = B.nodes[root].box.hi[a]; }
    auto leafRef = [](const TmpNode& n) -> int32_t { return ~(int32_t)((n.first << 3) | (n.count - 1)); };
    if (B.nodes[root].left < 0) { // whole scene is one leaf
        Bvh4Node n;
        clearNode(n);
        setChildBox(n, 0, B.nodes[root].box);
        n.child[0] = leafRef(B.nodes[root]);
        out->nodes.push_back(n);
        return;
    }
    // Collapse to four children per node: start from the two children of a binary inner node and keep
    // replacing the inner child with the largest box by its own two children.  Nodes are emitted in
    // depth-first preorder (explicit stack; depth tracked for the traversal stack bound).
    struct Item { int tmp; int outIndex; uint32_t depth; };
    std::vector<Item> stack;
    out->nodes.reserve(B.nodes.size() / 3 + 1);
    out->nodes.emplace_back();
    stack.push_back({root, 0, 1});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        out->maxDepth = std::max(out->maxDepth, it.depth);
        int kids[4] = {B.nodes[it.tmp].left, B.nodes[it.tmp].right, -1, -1};
        int nk = 2;
        while (nk < 4) {
            int best = -1;
            float bestArea = -1.f;
            for (int c = 0; c < nk; ++c) {
                const TmpNode& k = B.nodes[kids[c]];
                if (k.left >= 0 && k.box.area() > bestArea) { bestArea = k.box.area(); best = c; }
            }
            if (best < 0) break;
            int opened = kids[best];
            kids[best] = B.nodes[opened].left;
            kids[nk++] = B.nodes[opened].right;
        }
        Bvh4Node n;
        clearNode(n);
        int slots[4] = {-1, -1, -1, -1};
        for (int c = 0; c < nk; ++c) {
            const TmpNode& k = B.nodes[kids[c]];
            setChildBox(n, c, k.box);
            if (k.left < 0) n.child[c] = leafRef(k);
            else { n.child[c] = (int32_t)out->nodes.size(); slots[c] = n.child[c]; out->nodes.emplace_back(); }
        }
        out->nodes[it.outIndex] = n;
        for (int c = nk - 1; c >= 0; --c) // first child's subtree is laid out next
            if (slots[c] >= 0) stack.push_back({kids[c], slots[c], it.depth + 1});
    }
    out->stackNeed = 3 * out->maxDepth;
}

} // namespace enigma
