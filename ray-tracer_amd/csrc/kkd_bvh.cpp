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
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
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

/* builder-internal uncompressed 4-wide node */
struct Node4F { float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4]; uint32_t child[4]; int nk; };

struct Node2 { Box box; uint32_t left, right; uint32_t start, count; /* count > 0: leaf */ };

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
        Node2 nd; nd.box = box; nd.left = nd.right = 0; nd.start = lo; nd.count = 0;
        if (n <= 1) { nd.count = n; nodes[me] = nd; return me; }
        /* binned SAH */
        int bestAxis = -1, bestBin = -1; float bestCost = INFINITY;
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = cbox.mx[ax] - cbox.mn[ax];
            if (!(ext > 0.f)) continue;
            Box bb[kBins]; uint32_t cnt[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); cnt[b] = 0; }
            const float scale = kBins / ext;
            for (uint32_t i = lo; i < hi; ++i) {
                int b = std::min(kBins - 1, std::max(0, (int)((prims[i].c[ax] - cbox.mn[ax]) * scale)));
                bb[b].grow(prims[i].box); cnt[b]++;
            }
            float rightArea[kBins]; uint32_t rightCnt[kBins];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); c += cnt[b]; rightArea[b] = acc.area(); rightCnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); c += cnt[b];
                if (c == 0 || rightCnt[b + 1] == 0) continue;
                float cost = acc.area() * c + rightArea[b + 1] * rightCnt[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = b; }
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

struct Collapser {
    const Builder &b; const float *pos; const uint32_t *idx; const uint32_t *shapeOfPrim;
    std::vector<Node4F> &out; std::vector<LeafTri> &tris; BVH4::Info &info;

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
            nd.child[k] = emit(kids[k], depth + 1, stackHere);
        }
        out[me] = nd;
        return me;
    }
};

} /* namespace */

/* 8-bit grid encoding of a node's child boxes (layout: kkd_types.h).  Per axis: origin = min over the children,
 * scale = smallest power of two with extent <= 255 * scale; min planes round down, max planes up, and every plane is
 * checked through the kernel's own decode, fmaf(q, scale, origin), so the decoded box always encloses the true one. */
static inline float decodePlane(uint8_t q, float scale, float origin) { return std::fmaf((float)q, scale, origin); }

/* float boxes relative to the node's min corner, rounded outwards (so that origin + rel encloses the true plane) */
static void widenNode(const Node4F &n, WideNode &out) {
    std::memset(&out, 0, sizeof(out));
    const float *mins[3] = { n.minx, n.miny, n.minz }, *maxs[3] = { n.maxx, n.maxy, n.maxz };
    float *rmins[3] = { out.rminx, out.rminy, out.rminz }, *rmaxs[3] = { out.rmaxx, out.rmaxy, out.rmaxz };
    float origin[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY;
        for (int k = 0; k < n.nk; ++k) lo = std::min(lo, mins[a][k]);
        origin[a] = lo;
        for (int k = 0; k < 4; ++k) {
            if (k >= n.nk) { rmins[a][k] = INFINITY; rmaxs[a][k] = INFINITY; continue; }
            float r0 = mins[a][k] - lo, r1 = maxs[a][k] - lo;
            while (lo + r0 > mins[a][k]) r0 = std::nextafter(r0, -INFINITY);
            while (lo + r1 < maxs[a][k]) r1 = std::nextafter(r1, INFINITY);
            rmins[a][k] = r0; rmaxs[a][k] = r1;
        }
    }
    out.ox = origin[0]; out.oy = origin[1]; out.oz = origin[2];
    for (int k = 0; k < 4; ++k) out.child[k] = k < n.nk ? n.child[k] : 0u;
}

static void quantizeNode(const Node4F &n, Node4 &out) {
    std::memset(&out, 0, sizeof(out));
    const float *mins[3] = { n.minx, n.miny, n.minz }, *maxs[3] = { n.maxx, n.maxy, n.maxz };
    uint8_t *qmins[3] = { out.qminx, out.qminy, out.qminz }, *qmaxs[3] = { out.qmaxx, out.qmaxy, out.qmaxz };
    float origin[3]; uint8_t expo[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < n.nk; ++k) { lo = std::min(lo, mins[a][k]); hi = std::max(hi, maxs[a][k]); }
        origin[a] = lo;
        const float extent = hi - lo;
        int e = 0;                                   /* scale = 2^e */
        if (extent > 0.f) { std::frexp(extent / 255.0f, &e); }   /* extent/255 = m * 2^e, m in [0.5,1) => 2^e >= extent/255 */
        else e = -126;
        e = std::max(-126, std::min(127, e));
        float scale = std::ldexp(1.0f, e);
        /* make sure the top of the range is reachable after float rounding of the decode */
        while (e < 127 && decodePlane(255, scale, lo) < hi) { ++e; scale = std::ldexp(1.0f, e); }
        expo[a] = (uint8_t)(e + 127);
        for (int k = 0; k < 4; ++k) {
            if (k >= n.nk) { qmins[a][k] = 255; qmaxs[a][k] = 0; continue; }
            int q0 = (int)std::floor((mins[a][k] - lo) / scale), q1 = (int)std::ceil((maxs[a][k] - lo) / scale);
            q0 = std::max(0, std::min(255, q0)); q1 = std::max(0, std::min(255, q1));
            while (q0 > 0 && decodePlane((uint8_t)q0, scale, lo) > mins[a][k]) --q0;
            while (q1 < 255 && decodePlane((uint8_t)q1, scale, lo) < maxs[a][k]) ++q1;
            qmins[a][k] = (uint8_t)q0; qmaxs[a][k] = (uint8_t)q1;
        }
    }
    out.ox = origin[0]; out.oy = origin[1]; out.oz = origin[2];
    out.ex = expo[0]; out.ey = expo[1]; out.ez = expo[2]; out.epad = 0;
    for (int k = 0; k < 4; ++k) out.child[k] = k < n.nk ? n.child[k] : 0u;
}

bool BVH4::build(const float *positions, const uint32_t *indices, uint64_t nTris, const uint32_t *shapeOfPrim, std::string &err) {
    nodes.clear(); this->wide.clear(); tris.clear(); info = Info();
    if (nTris == 0) { err = "scene has no triangles"; return false; }
    if (nTris > (uint64_t)kLeafStartMask) { err = "too many triangles for the 28-bit leaf slot field"; return false; }
    Builder b;
    b.prims.reserve(nTris);
    for (uint64_t p = 0; p < nTris; ++p) {
        Prim pr; pr.box.reset(); pr.id = (uint32_t)p;
        bool finite = true;
        for (int k = 0; k < 3; ++k) {
            const float *v = positions + 3 * (size_t)indices[3 * p + k];
            for (int a = 0; a < 3; ++a) if (!std::isfinite(v[a])) finite = false;
            pr.box.grow(v);
        }
        if (!finite) { err = "non-finite vertex position"; return false; }
        for (int a = 0; a < 3; ++a) pr.c[a] = 0.5f * (pr.box.mn[a] + pr.box.mx[a]);
        b.prims.push_back(pr);
    }
    b.nodes.resize(2 * nTris);
    unsigned hw = std::thread::hardware_concurrency();
    b.freeThreads = (int)std::min(16u, hw ? hw : 1u) - 1;
    const uint32_t root = b.build(0, (uint32_t)nTris, 0);
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
        c.emit(root, 0, 0);
    }
    nodes.resize(wide.size()); this->wide.resize(wide.size());
    for (size_t i = 0; i < wide.size(); ++i) { quantizeNode(wide[i], nodes[i]); widenNode(wide[i], this->wide[i]); }
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
