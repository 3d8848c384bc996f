#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

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

// A few threads that live as long as one build.  run(n, fn) calls fn(i) for i in [0, n), items handed out one at a time,
// on the workers and on the calling thread, and returns when all are done.
class Pool {
public:
    explicit Pool(unsigned threads) {
        for (unsigned t = 1; t < threads; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; ++generation_; }
        wake_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    unsigned threads() const { return (unsigned)workers_.size() + 1; }
    template <class F>
    void run(uint32_t n, F fn) {
        if (workers_.empty() || n <= 1) { for (uint32_t i = 0; i < n; ++i) fn(i); return; }
        std::function<void(uint32_t)> job = fn;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job; n_ = n; next_.store(0); busy_ = (unsigned)workers_.size(); ++generation_;
        }
        wake_.notify_all();
        for (uint32_t i; (i = next_.fetch_add(1)) < n;) job(i);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return busy_ == 0; });
        job_ = nullptr;
    }
private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(uint32_t)>* job;
            uint32_t n;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (quit_) return;
                job = job_; n = n_;
            }
            for (uint32_t i; (i = next_.fetch_add(1)) < n;) (*job)(i);
            { std::lock_guard<std::mutex> g(m_); if (--busy_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(uint32_t)>* job_ = nullptr;
    uint32_t n_ = 0;
    std::atomic<uint32_t> next_{0};
    unsigned busy_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

struct Deferred { int node; uint32_t first, count, depth; }; // a subtree of the top of the tree that a worker builds

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
    // Ranges of more than kWide primitives are worked on in chunks of a size that depends on the range alone: boxes and
    // bins are unions and counts, the same in whatever order they are gathered, and the partition moves every chunk's
    // two sides to the front and back of the range in chunk order.  The tree and the primitive order therefore do not
    // depend on how many threads (`pool`) run the chunks, nor on which ranges the builder of the top of the tree leaves to
    // workers (`deferred`: ranges of at most `grain` primitives below the root).
    Pool* pool = nullptr;
    uint32_t grain = 0;
    std::vector<Deferred>* deferred = nullptr;
    std::vector<uint32_t> scratch; // the partition's second buffer
    static constexpr uint32_t kWide = 1u << 14, kChunk = 1u << 12;

    template <class F>
    void forChunks(uint32_t chunks, F fn) {
        if (pool) pool->run(chunks, fn);
        else for (uint32_t c = 0; c < chunks; ++c) fn(c);
    }
    Builder(const std::vector<Box>& tb, const std::vector<float>& c, std::vector<uint32_t>& o, uint32_t ml)
        : triBox(tb), centroid(c), order(o), maxLeaf(ml) {}

    struct Bins { Box bb[3][kMaxBins]; uint32_t bc[3][kMaxBins]; };
    void binRange(uint32_t first, uint32_t count, const Box& cbox, const float* ext, Bins* B) const {
        for (int a = 0; a < 3; ++a) {
            if (!(ext[a] > 0.f)) continue;
            Box bb[kMaxBins]; // local: the compiler keeps what it can of them out of memory
            uint32_t bc[kMaxBins] = {0};
            const float scale = kBins / ext[a], lo = cbox.lo[a];
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t t = order[i];
                const int b = binOf((centroid[3 * (size_t)t + a] - lo) * scale, kBins);
                bb[b].grow(triBox[t]);
                bc[b]++;
            }
            for (int b = 0; b < kBins; ++b) { B->bb[a][b] = bb[b]; B->bc[a][b] = bc[b]; }
        }
    }

    int build(uint32_t first, uint32_t count, uint32_t depth = 0) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        if (deferred && depth > 0 && count <= grain) { // a worker builds this subtree; its root keeps this slot
            deferred->push_back({id, first, count, depth});
            return id;
        }
        const bool wide = count > kWide;
        const uint32_t parts = wide ? std::min(64u, count / kChunk) : 1;
        auto partRange = [&](uint32_t p, uint32_t* f, uint32_t* c) {
            const uint64_t lo = (uint64_t)count * p / parts, hi = (uint64_t)count * (p + 1) / parts;
            *f = first + (uint32_t)lo; *c = (uint32_t)(hi - lo);
        };
        Box box, cbox;
        if (wide) {
            std::vector<Box> pb(parts), pc(parts);
            forChunks(parts, [&](uint32_t p) {
                uint32_t f, c;
                partRange(p, &f, &c);
                Box b, cb;
                for (uint32_t i = f; i < f + c; ++i) { b.grow(triBox[order[i]]); cb.grow(&centroid[3 * (size_t)order[i]]); }
                pb[p] = b; pc[p] = cb;
            });
            for (uint32_t p = 0; p < parts; ++p) { box.grow(pb[p]); cbox.grow(pc[p]); }
        } else {
            for (uint32_t i = first; i < first + count; ++i) {
                box.grow(triBox[order[i]]);
                cbox.grow(&centroid[3 * (size_t)order[i]]);
            }
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
            Bins bins;
            if (wide) {
                std::vector<Bins> pbins(parts);
                forChunks(parts, [&](uint32_t p) {
                    uint32_t f, c;
                    partRange(p, &f, &c);
                    binRange(f, c, cbox, ext, &pbins[p]);
                });
                for (int a = 0; a < 3; ++a) for (int b = 0; b < kBins; ++b) { bins.bb[a][b] = Box(); bins.bc[a][b] = 0; }
                for (uint32_t p = 0; p < parts; ++p)
                    for (int a = 0; a < 3; ++a) {
                        if (!(ext[a] > 0.f)) continue;
                        for (int b = 0; b < kBins; ++b) { bins.bb[a][b].grow(pbins[p].bb[a][b]); bins.bc[a][b] += pbins[p].bc[a][b]; }
                    }
            } else binRange(first, count, cbox, ext, &bins);
            double bestCost = HUGE_VAL;
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                if (ext[a] <= 0.f) continue;
                const Box* bb = bins.bb[a];
                const uint32_t* bc = bins.bc[a];
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
                auto goesLeft = [&](uint32_t t) { return binOf((centroid[3 * (size_t)t + bestAxis] - lo) * scale, kBins) <= bestBin; };
                if (wide) { // every chunk by itself, then the chunks' left sides to the front and their right sides behind them
                    std::vector<uint32_t> nLeft(parts);
                    forChunks(parts, [&](uint32_t p) {
                        uint32_t f, c;
                        partRange(p, &f, &c);
                        nLeft[p] = (uint32_t)(std::partition(order.begin() + f, order.begin() + f + c, goesLeft) - (order.begin() + f));
                    });
                    uint32_t totalLeft = 0;
                    for (uint32_t p = 0; p < parts; ++p) totalLeft += nLeft[p];
                    if (scratch.size() < count) scratch.resize(count);
                    std::vector<uint32_t> leftAt(parts), rightAt(parts);
                    for (uint32_t p = 0, l = 0, r = totalLeft; p < parts; ++p) {
                        uint32_t f, c;
                        partRange(p, &f, &c);
                        leftAt[p] = l; rightAt[p] = r;
                        l += nLeft[p]; r += c - nLeft[p];
                    }
                    forChunks(parts, [&](uint32_t p) {
                        uint32_t f, c;
                        partRange(p, &f, &c);
                        memcpy(&scratch[leftAt[p]], &order[f], (size_t)nLeft[p] * sizeof(uint32_t));
                        memcpy(&scratch[rightAt[p]], &order[f + nLeft[p]], (size_t)(c - nLeft[p]) * sizeof(uint32_t));
                    });
                    forChunks(parts, [&](uint32_t p) {
                        uint32_t f, c;
                        partRange(p, &f, &c);
                        memcpy(&order[f], &scratch[f - first], (size_t)c * sizeof(uint32_t));
                    });
                    mid = first + totalLeft;
                } else {
                    mid = (uint32_t)(std::partition(order.begin() + first, order.begin() + first + count, goesLeft) - order.begin());
                }
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
    // Threads: every step below that runs on several gives the result one thread gives (see Builder), so their number
    // only changes the time (ENIGMA_BVH_THREADS overrides it; at most 16 by default).  Small scenes are not worth a thread.
    unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("ENIGMA_BVH_THREADS")) threads = (unsigned)std::max(1, std::min(64, atoi(e)));
    if (nTris < (1u << 15)) threads = 1;
    Pool pool(threads);
    auto parallelFor = [&](uint32_t n, unsigned, auto fn) { pool.run(n, fn); };
    std::vector<Box> triBox(nTris);
    std::vector<float> centroid((size_t)nTris * 3);
    std::vector<uint32_t> order(nTris);
    {
        const uint32_t parts = threads > 1 ? threads * 4 : 1;
        parallelFor(parts, threads, [&](uint32_t p) {
            const uint32_t lo = (uint32_t)((uint64_t)nTris * p / parts), hi = (uint32_t)((uint64_t)nTris * (p + 1) / parts);
            for (uint32_t t = lo; t < hi; ++t) {
                if (t < nTriangles) for (int k = 0; k < 3; ++k) triBox[t].grow(&vertices[3 * (size_t)indices[3 * (size_t)t + k]]);
                else { triBox[t].grow(&boxes[6 * (size_t)(t - nTriangles)]); triBox[t].grow(&boxes[6 * (size_t)(t - nTriangles) + 3]); }
                for (int a = 0; a < 3; ++a) centroid[3 * (size_t)t + a] = 0.5f * triBox[t].lo[a] + 0.5f * triBox[t].hi[a];
                order[t] = t;
            }
        });
    }
    out->nodes.clear();
    out->triOrder.clear();
    out->maxDepth = 1;
    for (int a = 0; a < 3; ++a) { out->rootLo[a] = 0; out->rootHi[a] = 0; }
    auto emptyNode = [] { Bvh8Node n; memset(&n, 0, sizeof(n)); n.e[0] = n.e[1] = n.e[2] = 1; return n; };
    if (nTris == 0) { out->nodes.push_back(emptyNode()); return; }
    Builder B(triBox, centroid, order, 1); // down to single triangles: the collapse below decides what a leaf is
    if (const char* e = getenv("ENIGMA_BVH_BINS")) B.kBins = std::max(4, std::min(64, atoi(e)));
    // The top of the tree is built here, with the large ranges binned on all threads; ranges of at most `grain`
    // primitives are built by workers, each into a node array of its own over its own piece of the order array, and
    // appended afterwards (the subtree's root takes the slot the top left for it, the rest follow: parents still come
    // before their children, which the sweep of the collapse below relies on).
    std::vector<Deferred> deferred;
    struct Span { size_t base, size; int root; }; // nodes [base, base + size) + `root` hold one worker's subtree
    std::vector<Span> spans;
    if (threads > 1) {
        B.pool = &pool;
        B.grain = std::max(1u << 12, nTris / (threads * 8));
        B.deferred = &deferred;
    }
    B.nodes.reserve(threads > 1 ? (size_t)1 << 12 : (size_t)nTris * 2);
    const int root = B.build(0, nTris);
    if (!deferred.empty()) {
        std::vector<uint32_t> byCost(deferred.size());
        for (uint32_t i = 0; i < byCost.size(); ++i) byCost[i] = i;
        std::sort(byCost.begin(), byCost.end(), [&](uint32_t x, uint32_t y) { return deferred[x].count > deferred[y].count; }); // large ones first
        std::vector<std::vector<TmpNode>> built(deferred.size());
        parallelFor((uint32_t)deferred.size(), threads, [&](uint32_t k) {
            const Deferred& d = deferred[byCost[k]];
            Builder W(triBox, centroid, order, 1);
            W.kBins = B.kBins;
            W.nodes.reserve((size_t)d.count * 2);
            W.build(d.first, d.count, d.depth);
            built[byCost[k]] = std::move(W.nodes);
        });
        size_t total = B.nodes.size();
        spans.resize(deferred.size());
        for (size_t k = 0; k < deferred.size(); ++k) { spans[k] = {total, built[k].size() - 1, deferred[k].node}; total += built[k].size() - 1; }
        B.nodes.resize(total);
        parallelFor((uint32_t)deferred.size(), threads, [&](uint32_t k) {
            const Span& sp = spans[k];
            auto remap = [&](int local) { return local < 0 ? -1 : (local == 0 ? sp.root : (int)(sp.base + (size_t)local - 1)); };
            for (size_t i = 0; i < built[k].size(); ++i) {
                TmpNode n = built[k][i];
                n.left = remap(n.left); n.right = remap(n.right);
                B.nodes[(size_t)remap((int)i)] = n;
            }
            std::vector<TmpNode>().swap(built[k]);
        });
    }
    for (int a = 0; a < 3; ++a) { out->rootLo[a] = B.nodes[root].box.lo[a]; out->rootHi[a] = B.nodes[root].box.hi[a]; }

    // Which binary nodes become 8-wide nodes, which subtrees become leaves: the collapse of least surface-area cost
    // (Ylitie, Karras, Laine 2017, section 4.1).  cost[n][i] = cheapest way to stand for the subtree of n with at most
    // i roots (wide nodes or leaves), i = 1..7:
    //   cost[n][1] = min(leaf, wide node),  leaf = area * triangles * cPrim (at most 3 triangles),
    //                                       wide node = area * cNode + distribute(n, 8)
    //   cost[n][i] = min(distribute(n, i), cost[n][i - 1]),  distribute(n, j) = min over k of cost[left][k] + cost[right][j - k]
    // Binary nodes are numbered parent before children, so one backward sweep sees children first; the workers' subtrees
    // are swept side by side (each is closed under "child of"), then the top of the tree.
    const double cNode = 1.0;
    double cPrim = 0.6; // a node visit against one triangle test, by their instruction counts
    if (const char* e = getenv("ENIGMA_BVH_CPRIM")) cPrim = atof(e);
    const size_t nb = B.nodes.size();
    std::vector<double> cost(nb * 8, 0.0);       // [n][i], i = 1..7 (slot 0 unused)
    std::vector<uint8_t> splitK(nb * 9, 0);      // [n][j], j = 2..8: the k of distribute(n, j)
    std::vector<uint8_t> takesDist(nb * 8, 0);   // [n][i], i = 2..7: cost[n][i] is distribute(n, i), not cost[n][i - 1]
    std::vector<uint8_t> asLeaf(nb, 0);
    std::vector<uint8_t> swept(spans.empty() ? 0 : nb, 0);
    auto sweepNode = [&](size_t ni) {
        const TmpNode& n = B.nodes[ni];
        const double area = n.box.area();
        const double leafCost = n.count <= 3 ? area * (double)n.count * cPrim : HUGE_VAL;
        if (n.left < 0) { // a single triangle
            for (int i = 1; i <= 7; ++i) cost[ni * 8 + i] = leafCost;
            asLeaf[ni] = 1;
            return;
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
    };
    if (!spans.empty()) {
        parallelFor((uint32_t)spans.size(), threads, [&](uint32_t k) {
            const Span& sp = spans[k];
            for (size_t ni = sp.base + sp.size; ni-- > sp.base;) sweepNode(ni);
            sweepNode((size_t)sp.root);
            swept[(size_t)sp.root] = 1;
        });
        for (size_t ni = spans[0].base; ni-- > 0;) if (!swept[ni]) sweepNode(ni); // the top of the tree
    } else {
        for (size_t ni = nb; ni-- > 0;) sweepNode(ni);
    }
    // the roots that stand for subtree n when it may use up to i of them
    struct Expand {
        const std::vector<TmpNode>& nodes; const std::vector<uint8_t>& takesDist; const std::vector<uint8_t>& splitK;
        void operator()(int n, int i, int* kids, int* nk) const {
            while (i > 1 && !takesDist[(size_t)n * 8 + i]) --i;
            if (i == 1 || nodes[n].left < 0) { kids[(*nk)++] = n; return; }
            const int k = splitK[(size_t)n * 9 + i];
            (*this)(nodes[n].left, k, kids, nk);
            (*this)(nodes[n].right, i - k, kids, nk);
        }
    } expand{B.nodes, takesDist, splitK};
    auto isLeaf = [&](int tmp) { return asLeaf[tmp] != 0; };

    // Breadth first, a level at a time: the children of a node are consecutive, siblings' subtrees follow level by
    // level.  Per level: (1) every node's children and their slots, side by side; (2) where each node's inner children
    // and leaf triangles go -- running sums in the level's order, exactly where a node-by-node walk would put them;
    // (3) the nodes themselves, side by side.
    struct Item { int tmp; uint32_t outIndex; };
    struct Plan { int kidTmp[8]; uint32_t nInner, nTri, childBase, triBase; }; // kidTmp[slot] = the binary node in that slot, or -1
    std::vector<Item> level{{root, 0u}}, nextLevel;
    std::vector<Plan> plan;
    out->nodes.reserve(nb / 8 + 16);
    out->nodes.push_back(emptyNode());
    out->triOrder.resize(nTris);
    size_t triCursor = 0;
    for (uint32_t depth = 1; !level.empty(); ++depth) {
        out->maxDepth = std::max(out->maxDepth, depth);
        plan.resize(level.size());
        const uint32_t chunks = threads > 1 ? (uint32_t)std::min<size_t>(threads * 4, (level.size() + 255) / 256) : 1;
        auto chunkRange = [&](uint32_t c, size_t* lo, size_t* hi) { *lo = level.size() * c / chunks; *hi = level.size() * (c + 1) / chunks; };
        parallelFor(chunks, threads, [&](uint32_t c) {
            size_t lo, hi;
            chunkRange(c, &lo, &hi);
            for (size_t q = lo; q < hi; ++q) {
                const Item it = level[q];
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
                for (int ch = 0; ch < nk; ++ch) {
                    const Box& kb = B.nodes[kids[ch]].box;
                    float rel[3];
                    for (int a = 0; a < 3; ++a) rel[a] = (0.5f * kb.lo[a] + 0.5f * kb.hi[a]) - ctr[a];
                    for (int s = 0; s < 8; ++s) score[ch][s] = ((s & 1) ? -rel[0] : rel[0]) + ((s & 2) ? -rel[1] : rel[1]) + ((s & 4) ? -rel[2] : rel[2]);
                }
                int slotOf[8], kidAt[8];
                for (int i = 0; i < 8; ++i) { slotOf[i] = -1; kidAt[i] = -1; }
                for (int round = 0; round < nk; ++round) {
                    int bc = -1, bs = -1;
                    float bestScore = -FLT_MAX;
                    for (int ch = 0; ch < nk; ++ch) {
                        if (slotOf[ch] >= 0) continue;
                        for (int s = 0; s < 8; ++s)
                            if (kidAt[s] < 0 && score[ch][s] > bestScore) { bestScore = score[ch][s]; bc = ch; bs = s; }
                    }
                    slotOf[bc] = bs; kidAt[bs] = bc;
                }
                Plan& pl = plan[q];
                pl.nInner = 0; pl.nTri = 0;
                for (int s = 0; s < 8; ++s) {
                    pl.kidTmp[s] = kidAt[s] < 0 ? -1 : kids[kidAt[s]];
                    if (pl.kidTmp[s] < 0) continue;
                    if (isLeaf(pl.kidTmp[s])) pl.nTri += B.nodes[pl.kidTmp[s]].count; else ++pl.nInner;
                }
            }
        });
        size_t nodeCursor = out->nodes.size();
        for (Plan& pl : plan) {
            pl.childBase = (uint32_t)nodeCursor; pl.triBase = (uint32_t)triCursor;
            nodeCursor += pl.nInner; triCursor += pl.nTri;
        }
        const size_t firstChild = out->nodes.size();
        out->nodes.resize(nodeCursor, emptyNode());
        nextLevel.resize(nodeCursor - firstChild);
        parallelFor(chunks, threads, [&](uint32_t c) {
            size_t lo, hi;
            chunkRange(c, &lo, &hi);
            for (size_t q = lo; q < hi; ++q) {
                const Item it = level[q];
                const Plan& pl = plan[q];
                const Box& nbox = B.nodes[it.tmp].box;
                Bvh8Node n = emptyNode();
                for (int a = 0; a < 3; ++a) { n.lo[a] = nbox.lo[a]; n.e[a] = exponentFor((nbox.hi[a] - nbox.lo[a]) * 1.0001f); }
                n.childBase = pl.childBase;
                n.triBase = pl.triBase;
                uint32_t nInner = 0, triOff = 0;
                for (int s = 0; s < 8; ++s) {
                    if (pl.kidTmp[s] < 0) continue;
                    const int tmp = pl.kidTmp[s];
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
                        nextLevel[(size_t)n.childBase + nInner - firstChild] = {tmp, n.childBase + nInner};
                        ++nInner;
                    } else {
                        const uint32_t unary = k.count == 1 ? 0x20u : (k.count == 2 ? 0x60u : 0xe0u);
                        n.meta[s] = (uint8_t)(unary | triOff);
                        for (uint32_t i = 0; i < k.count; ++i) out->triOrder[(size_t)n.triBase + triOff + i] = order[k.first + i];
                        triOff += k.count;
                    }
                }
                out->nodes[it.outIndex] = n;
            }
        });
        level.swap(nextLevel);
        nextLevel.clear();
    }
    out->triOrder.resize(triCursor);
}

} // namespace enigma
