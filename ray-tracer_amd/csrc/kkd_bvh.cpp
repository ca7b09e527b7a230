/*
 * kkd_bvh.cpp -- host-side builder of the 4-wide BVH the gfx950 traversal kernels walk.
 *
 * Replaces ShapeKDTree::build (reference src/librender/skdtree.cpp:68-110: SAH kd-tree + TriAccel
 * table).  The reference's kd-tree duplicates straddling triangles into several leaves and is walked
 * with a per-ray stack of far children; on a 64-wide wavefront a BVH with disjoint primitive sets,
 * four children per 128-byte node and <= 4 triangles per leaf gives fewer, wider, line-aligned fetches.
 *
 * Build: binned SAH (16 bins, 3 axes) over triangle centroids into a binary tree, large subtrees in
 * parallel std::threads; then collapse to 4 children per node by repeatedly opening the child with the
 * largest surface area; nodes are emitted in depth-first order, triangles in leaf order.
 */
#include "kkd_bvh.h"
#include "kkd_bvh_encode.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <thread>

namespace kkd {

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; } }
    void grow(const Box &b) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); } }
    void grow(const float *p) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
    float area() const {
        float d[3] = { mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2] };
        if (d[0] < 0) return 0.f;
        return 2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
    }
};

struct Prim { Box box; float c[3]; uint32_t id; };


struct Node2 { Box box; uint32_t left, right; uint32_t start, count; /* count > 0: leaf */ uint32_t span; /* primitives below */ };

struct Builder {
    std::vector<Prim> prims;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> nextNode{0};
    std::atomic<int> freeThreads{0};
    static constexpr int kBins = 16;
    static constexpr float kTravCost = 1.0f, kTriCost = 1.0f;

    uint32_t alloc() { return nextNode.fetch_add(1); }

    uint32_t build(uint32_t lo, uint32_t hi, int depth) {
        const uint32_t n = hi - lo, me = alloc();
        Box box, cbox; box.reset(); cbox.reset();
        for (uint32_t i = lo; i < hi; ++i) { box.grow(prims[i].box); cbox.grow(prims[i].c); }
        Node2 nd; nd.box = box; nd.left = nd.right = 0; nd.start = lo; nd.count = 0; nd.span = n;
        if (n <= 1) { nd.count = n; nodes[me] = nd; return me; }
        /* binned SAH: one pass over the primitives fills the bins of all three axes */
        int bestAxis = -1, bestBin = -1; float bestCost = INFINITY;
        {
            Box bb[3][kBins]; uint32_t cnt[3][kBins]; float scale[3]; bool use[3];
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cbox.mx[ax] - cbox.mn[ax];
                use[ax] = ext > 0.f; scale[ax] = use[ax] ? kBins / ext : 0.f;
                for (int k = 0; k < kBins; ++k) { bb[ax][k].reset(); cnt[ax][k] = 0; }
            }
            for (uint32_t i = lo; i < hi; ++i) {
                const Prim &p = prims[i];
                for (int ax = 0; ax < 3; ++ax) {
                    if (!use[ax]) continue;
                    const int k = std::min(kBins - 1, std::max(0, (int)((p.c[ax] - cbox.mn[ax]) * scale[ax])));
                    bb[ax][k].grow(p.box); cnt[ax][k]++;
                }
            }
            for (int ax = 0; ax < 3; ++ax) {
                if (!use[ax]) continue;
                float rightArea[kBins]; uint32_t rightCnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[ax][k]); c += cnt[ax][k]; rightArea[k] = acc.area(); rightCnt[k] = c; }
                acc.reset(); c = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    acc.grow(bb[ax][k]); c += cnt[ax][k];
                    if (c == 0 || rightCnt[k + 1] == 0) continue;
                    float cost = acc.area() * c + rightArea[k + 1] * rightCnt[k + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = k; }
                }
            }
        }
        const float leafCost = kTriCost * n;
        const float splitCost = (bestAxis >= 0 && box.area() > 0.f) ? kTravCost + kTriCost * bestCost / box.area() : INFINITY;
        if (n <= (uint32_t)kMaxLeafTris && leafCost <= splitCost) { nd.count = n; nodes[me] = nd; return me; }
        uint32_t mid;
        if (bestAxis < 0 || depth > 56) {
            /* all centroids coincide (or the tree got too deep): split the range in the middle of the widest axis */
            int ax = 0;
            for (int a = 1; a < 3; ++a) if (cbox.mx[a] - cbox.mn[a] > cbox.mx[ax] - cbox.mn[ax]) ax = a;
            mid = lo + n / 2;
            std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                             [ax](const Prim &a, const Prim &b) { return a.c[ax] < b.c[ax]; });
        } else {
            const float scale = kBins / (cbox.mx[bestAxis] - cbox.mn[bestAxis]), cmin = cbox.mn[bestAxis];
            const int ax = bestAxis, bin = bestBin;
            auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [=](const Prim &p) {
                int b = std::min(kBins - 1, std::max(0, (int)((p.c[ax] - cmin) * scale)));
                return b <= bin;
            });
            mid = (uint32_t)(it - prims.begin());
            if (mid == lo || mid == hi) mid = lo + n / 2;
        }
        uint32_t l, r;
        if (n > 32768 && freeThreads.fetch_sub(1) > 0) {
            auto fut = std::async(std::launch::async, [&, this] { return build(lo, mid, depth + 1); });
            r = build(mid, hi, depth + 1);
            l = fut.get();
            freeThreads.fetch_add(1);
        } else {
            if (n > 32768) freeThreads.fetch_add(1);
            l = build(lo, mid, depth + 1);
            r = build(mid, hi, depth + 1);
        }
        nd.left = l; nd.right = r;
        nodes[me] = nd;
        return me;
    }
};

/* Subtrees of at most deferBelow primitives are not collapsed by the caller's walk: it records them (in DFS order) and they are
 * collapsed side by side into vectors of their own, which are then appended -- nodes after the top of the tree, triangles after
 * the top walk's own leaves. */
struct Deferred { uint32_t root; uint32_t parent; int slot; int depth, stackAbove; };
struct Collapser {
    const Builder &b; const float *pos; const uint32_t *idx; const uint32_t *shapeOfPrim;
    std::vector<Node4F> &out; std::vector<LeafTri> &tris; BVH4::Info &info;
    uint32_t deferBelow = 0; std::vector<Deferred> *deferred = nullptr;

    uint32_t emitLeaf(const Node2 &n) {
        const uint32_t start = (uint32_t)tris.size();
        for (uint32_t i = 0; i < n.count; ++i) {
            const uint32_t p = b.prims[n.start + i].id;
            const float *a = pos + 3 * (size_t)idx[3 * (size_t)p], *bb = pos + 3 * (size_t)idx[3 * (size_t)p + 1], *c = pos + 3 * (size_t)idx[3 * (size_t)p + 2];
            LeafTri t;
            t.v0x = a[0]; t.v0y = a[1]; t.v0z = a[2]; t.prim = p;
            t.v1x = bb[0]; t.v1y = bb[1]; t.v1z = bb[2]; t.shape = shapeOfPrim[p];
            t.v2x = c[0]; t.v2y = c[1]; t.v2z = c[2]; t.emitter = 0;
            tris.push_back(t);
        }
        return kLeafBit | ((n.count - 1) << 28) | start;
    }

    /* returns the child reference for binary node i; depth/stack bookkeeping for the traversal stack bound */
    uint32_t emit(uint32_t i, int depth, int stackAbove) {
        const Node2 &n = b.nodes[i];
        info.maxDepth = std::max(info.maxDepth, depth);
        if (n.count > 0) return emitLeaf(n);
        uint32_t kids[4]; int nk = 2;
        kids[0] = n.left; kids[1] = n.right;
        while (nk < 4) {
            int best = -1; float bestArea = -1.f;
            for (int k = 0; k < nk; ++k) {
                const Node2 &c = b.nodes[kids[k]];
                if (c.count == 0 && c.box.area() > bestArea) { bestArea = c.box.area(); best = k; }
            }
            if (best < 0) break;
            const Node2 &c = b.nodes[kids[best]];
            kids[best] = c.left; kids[nk++] = c.right;
        }
        const uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        Node4F nd; nd.nk = nk;
        for (int k = 0; k < 4; ++k) {
            nd.minx[k] = nd.miny[k] = nd.minz[k] = INFINITY; nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = INFINITY;
            nd.child[k] = 0;   /* empty slot */
        }
        /* a visit pushes at most nk-1 siblings */
        const int stackHere = stackAbove + nk - 1;
        info.maxStack = std::max(info.maxStack, stackHere);
        for (int k = 0; k < nk; ++k) {
            const Node2 &c = b.nodes[kids[k]];
            nd.minx[k] = c.box.mn[0]; nd.miny[k] = c.box.mn[1]; nd.minz[k] = c.box.mn[2];
            nd.maxx[k] = c.box.mx[0]; nd.maxy[k] = c.box.mx[1]; nd.maxz[k] = c.box.mx[2];
            if (deferred && c.count == 0 && c.span <= deferBelow) { deferred->push_back(Deferred{ kids[k], me, k, depth + 1, stackHere }); nd.child[k] = 0; }
            else nd.child[k] = emit(kids[k], depth + 1, stackHere);
        }
        out[me] = nd;
        return me;
    }
};

} /* namespace */

bool BVH4::build(const float *positions, const uint32_t *indices, uint64_t nTris, const uint32_t *shapeOfPrim, std::string &err) {
    nodes.clear(); this->wide.clear(); tris.clear(); info = Info();
    if (nTris == 0) { err = "scene has no triangles"; return false; }
    if (nTris > (uint64_t)kLeafStartMask) { err = "too many triangles for the 28-bit leaf slot field"; return false; }
    const bool timing = getenv("KKD_BVH_TIMING") != nullptr;
    auto tPrev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!timing) return; auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "bvh: %-10s %.3f s\n", what, std::chrono::duration<double>(t - tPrev).count()); tPrev = t; };
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nThreads = std::min(16u, hw ? hw : 1u);
    /* run f(begin, end) over [0, n) on the builder's threads */
    auto parallelFor = [&](size_t n, const std::function<void(size_t, size_t)> &f) {
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nThreads, n / 65536));
        if (nt <= 1) { f(0, n); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(f, n * t / nt, n * (t + 1) / nt);
        for (std::thread &x : th) x.join();
    };
    Builder b;
    b.prims.resize(nTris);
    std::atomic<bool> allFinite{true};
    parallelFor(nTris, [&](size_t lo, size_t hi) {
        bool finite = true;
        for (size_t p = lo; p < hi; ++p) {
            Prim pr; pr.box.reset(); pr.id = (uint32_t)p;
            for (int k = 0; k < 3; ++k) {
                const float *v = positions + 3 * (size_t)indices[3 * p + k];
                for (int a = 0; a < 3; ++a) if (!std::isfinite(v[a])) finite = false;
                pr.box.grow(v);
            }
            for (int a = 0; a < 3; ++a) pr.c[a] = 0.5f * (pr.box.mn[a] + pr.box.mx[a]);
            b.prims[p] = pr;
        }
        if (!finite) allFinite = false;
    });
    if (!allFinite) { err = "non-finite vertex position"; return false; }
    b.nodes.resize(2 * nTris);
    b.freeThreads = (int)nThreads - 1;
    lap("prims");
    const uint32_t root = b.build(0, (uint32_t)nTris, 0);
    lap("sah build");
    tris.reserve(nTris);
    std::vector<Node4F> wide;
    wide.reserve(nTris / 2 + 16);
    Collapser c{ b, positions, indices, shapeOfPrim, wide, tris, info };
    const Node2 &r = b.nodes[root];
    for (int a = 0; a < 3; ++a) { info.bmin[a] = r.box.mn[a]; info.bmax[a] = r.box.mx[a]; }
    if (r.count > 0) {
        /* single-leaf scene: wrap it in a root node */
        Node4F nd; nd.nk = 1;
        for (int k = 0; k < 4; ++k) {
            nd.minx[k] = nd.miny[k] = nd.minz[k] = INFINITY; nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = INFINITY;
            nd.child[k] = 0;
        }
        nd.minx[0] = r.box.mn[0]; nd.miny[0] = r.box.mn[1]; nd.minz[0] = r.box.mn[2];
        nd.maxx[0] = r.box.mx[0]; nd.maxy[0] = r.box.mx[1]; nd.maxz[0] = r.box.mx[2];
        wide.emplace_back();
        nd.child[0] = c.emitLeaf(r);
        wide[0] = nd;
        info.maxDepth = 1; info.maxStack = 0;
    } else {
        std::vector<Deferred> deferred;
        /* the deferral threshold is a function of the triangle count alone, so that the node and triangle order (hence every
         * slot number) is the same on every host, whatever its core count: the top walk emits its inline leaves first and the
         * deferred subtrees' triangles are appended after them, which is NOT the leaf order of one serial walk */
        if (nTris > 262144) { c.deferBelow = (uint32_t)std::max<uint64_t>(65536, nTris / 128); c.deferred = &deferred; }
        c.emit(root, 0, 0);
        if (!deferred.empty()) {
            /* the deferred subtrees, collapsed side by side into their own vectors (local node indices and leaf slots) */
            struct Part { std::vector<Node4F> nodes; std::vector<LeafTri> tris; BVH4::Info info; uint32_t ref = 0; };
            std::vector<Part> parts(deferred.size());
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < std::min<size_t>(nThreads, deferred.size()); ++t) th.emplace_back([&] {
                for (size_t k; (k = next.fetch_add(1)) < deferred.size();) {
                    Part &p = parts[k];
                    p.tris.reserve(b.nodes[deferred[k].root].span);
                    Collapser sub{ b, positions, indices, shapeOfPrim, p.nodes, p.tris, p.info };
                    p.ref = sub.emit(deferred[k].root, deferred[k].depth, deferred[k].stackAbove);
                }
            });
            for (std::thread &x : th) x.join();
            /* append in DFS order of their roots; local references move by the part's base */
            for (size_t k = 0; k < parts.size(); ++k) {
                Part &p = parts[k];
                const uint32_t nodeBase = (uint32_t)wide.size(), triBase = (uint32_t)tris.size();
                auto moved = [&](uint32_t ref) { return ref == 0u ? 0u : ((ref & kLeafBit) ? ref + triBase : ref + nodeBase); };
                for (Node4F &nd : p.nodes) for (int q = 0; q < 4; ++q) nd.child[q] = moved(nd.child[q]);
                /* the part's root is its local node 0: a valid reference here, not the "empty slot" marker */
                wide[deferred[k].parent].child[deferred[k].slot] = (p.ref & kLeafBit) ? p.ref + triBase : p.ref + nodeBase;
                wide.insert(wide.end(), p.nodes.begin(), p.nodes.end());
                tris.insert(tris.end(), p.tris.begin(), p.tris.end());
                info.maxDepth = std::max(info.maxDepth, p.info.maxDepth); info.maxStack = std::max(info.maxStack, p.info.maxStack);
                std::vector<Node4F>().swap(p.nodes); std::vector<LeafTri>().swap(p.tris);
            }
        }
    }
    lap("collapse");
    nodes.resize(wide.size()); this->wide.resize(wide.size());
    parallelFor(wide.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) { quantizeNode(wide[i], nodes[i]); widenNode(wide[i], this->wide[i]); }
    });
    lap("quantise");
    info.nNodes = nodes.size(); info.nLeafTris = tris.size();
    return true;
}

/* TriAccel::load of the reference (include/mitsuba/render/triaccel.h, used at skdtree.cpp:77-104): Wald's
 * precomputed projection record.  Single precision, no contraction, the operation order of the published code, so that
 * the device test reproduces the reference's hit / miss decisions and (t, u, v) bit for bit. */
void makeProjTris(const std::vector<LeafTri> &tris, std::vector<ProjTri> &out) {
    static const int waldModulo[4] = { 1, 2, 0, 1 };
    out.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        const LeafTri &t = tris[i];
        ProjTri r; std::memset(&r, 0, sizeof(r));
        r.prim = t.prim;
        const volatile float A[3] = { t.v0x, t.v0y, t.v0z };
        const float b[3] = { t.v2x - t.v0x, t.v2y - t.v0y, t.v2z - t.v0z };   /* C - A */
        const float c[3] = { t.v1x - t.v0x, t.v1y - t.v0y, t.v1z - t.v0z };   /* B - A */
        const float N[3] = { c[1] * b[2] - c[2] * b[1], c[2] * b[0] - c[0] * b[2], c[0] * b[1] - c[1] * b[0] };
        uint32_t k = 0;
        for (int j = 0; j < 3; ++j) if (std::fabs(N[j]) > std::fabs(N[k])) k = (uint32_t)j;
        const int u = waldModulo[k], v = waldModulo[k + 1];
        const float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
        if (denom == 0.f) { r.k = 3; out[i] = r; continue; }
        r.k = k;
        r.n_u = N[u] / n_k; r.n_v = N[v] / n_k; r.n_d = (A[0] * N[0] + A[1] * N[1] + A[2] * N[2]) / n_k;
        r.b_nu = b[u] / denom; r.b_nv = -b[v] / denom;
        r.a_u = A[u]; r.a_v = A[v];
        r.c_nu = c[v] / denom; r.c_nv = -c[u] / denom;
        out[i] = r;
    }
}

} /* namespace kkd */
