/*
 * ORACLE (test infrastructure only): CPU restatement of the reference's ray/scene intersection.
 *
 * Follows (relative to /root/reference/src):
 *   librender/skdtree.cpp:49-110   shape map, TriAccel table
 *   librender/skdtree.cpp:112-142  closest-hit entry (AABB clip, adaptive epsilon with Epsilon clamp)
 *   librender/skdtree.cpp:207-226  any-hit entry (adaptive epsilon WITHOUT the inner clamp)
 *   libcore/triangle.cpp:69-166    Sutherland-Hodgman clip in double precision (perfect splits)
 *   tests/test_kd.cpp:34-84        known answers for the clip
 *   tests/test_dgeom.cpp:35-120    known answers for the hit record
 * Bodies that live in the absent include/mitsuba headers (triaccel.h, gkdtree.h, sahkdtree3.h,
 * skdtree.h, aabb.h) are restated from the published Mitsuba 0.5 algorithm as recorded in
 * Wald's projection test, SAH kd-tree (traversal cost 15, query cost 20,
 * empty-space bonus 0.9, stop at 6 prims, depth 8+1.3 log2 N <= 48, exact sweep with perfect
 * splits below 65536 prims and 128-bin min-max binning above), Havran's TA^B_rec traversal.
 * Parity of those bodies is pinned only by test_dgeom / test_kd ("parity unpinned" otherwise).
 */
#pragma once
#include "ora_math.h"
#include <vector>
#include <cstdio>
#include <cassert>
#include <atomic>
#include <cstdlib>
#include <thread>

namespace ora {

struct AABB {
    V3 min, max;
    AABB() : min(std::numeric_limits<Float>::infinity()), max(-std::numeric_limits<Float>::infinity()) {}
    AABB(V3 a, V3 b) : min(a), max(b) {}
    void expandBy(V3 p) {
        for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], p[i]); max[i] = std::max(max[i], p[i]); }
    }
    void expandBy(const AABB &b) { expandBy(b.min); expandBy(b.max); }
    bool isValid() const { return max.x >= min.x && max.y >= min.y && max.z >= min.z; }
    void clip(const AABB &b) {
        for (int i = 0; i < 3; ++i) { min[i] = std::max(min[i], b.min[i]); max[i] = std::min(max[i], b.max[i]); }
    }
    Float surfaceArea() const {
        V3 d = max - min;
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    V3 center() const { return (max + min) * 0.5f; }
    /* aabb.h (absent): slab test returning the signed entry/exit distances */
    bool rayIntersect(V3 o, V3 d, V3 dRcp, Float &nearT, Float &farT) const {
        nearT = -std::numeric_limits<Float>::infinity();
        farT = std::numeric_limits<Float>::infinity();
        for (int i = 0; i < 3; ++i) {
            const Float origin = o[i], minVal = min[i], maxVal = max[i];
            if (d[i] == 0) {
                if (origin < minVal || origin > maxVal) return false;
            } else {
                Float t1 = (minVal - origin) * dRcp[i], t2 = (maxVal - origin) * dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT)) return false;
            }
        }
        return true;
    }
};

struct Ray {
    V3 o, d, dRcp; Float mint, maxt;
    /* RayDifferential (ray.h, absent): only the direction offsets are used on this path (envmap.cpp:394-407) */
    bool hasDifferentials = false; V3 rxDirection, ryDirection;
    Ray() {}
    Ray(V3 o_, V3 d_, Float mint_, Float maxt_) : o(o_), d(d_), mint(mint_), maxt(maxt_) {
        dRcp = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    }
    V3 at(Float t) const { return o + d * t; }
};

/* ---- triaccel.h (absent) ----------------------------------------------------- */
struct TriAccel {
    uint32_t k;
    Float n_u, n_v, n_d;
    Float a_u, a_v;
    Float b_nu, b_nv;
    Float c_nu, c_nv;
    uint32_t shapeIndex, primIndex;

    void load(V3 A, V3 B, V3 C) {
        static const int waldModulo[4] = { 1, 2, 0, 1 };
        V3 b = C - A, c = B - A, N = cross(c, b);
        k = 0;
        for (int j = 0; j < 3; ++j)
            if (std::fabs(N[j]) > std::fabs(N[(int)k])) k = (uint32_t)j;
        int u = waldModulo[k], v = waldModulo[k + 1];
        const Float n_k = N[(int)k], denom = b[u] * c[v] - b[v] * c[u];
        if (denom == 0) { k = 3; return; }
        n_u = N[u] / n_k; n_v = N[v] / n_k; n_d = dot(A, N) / n_k;
        b_nu = b[u] / denom; b_nv = -b[v] / denom;
        a_u = A[u]; a_v = A[v];
        c_nu = c[v] / denom; c_nv = -c[u] / denom;
    }
    bool rayIntersect(const Ray &ray, Float mint, Float maxt, Float &u, Float &v, Float &t) const {
        Float o_u, o_v, o_k, d_u, d_v, d_k;
        switch (k) {
            case 0: o_u = ray.o.y; o_v = ray.o.z; o_k = ray.o.x; d_u = ray.d.y; d_v = ray.d.z; d_k = ray.d.x; break;
            case 1: o_u = ray.o.z; o_v = ray.o.x; o_k = ray.o.y; d_u = ray.d.z; d_v = ray.d.x; d_k = ray.d.y; break;
            case 2: o_u = ray.o.x; o_v = ray.o.y; o_k = ray.o.z; d_u = ray.d.x; d_v = ray.d.y; d_k = ray.d.z; break;
            default: return false;
        }
        t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
        if (t < mint || t > maxt) return false;
        const Float hu = o_u + t * d_u - a_u, hv = o_v + t * d_v - a_v;
        u = hv * b_nu + hu * b_nv;
        v = hu * c_nu + hv * c_nv;
        return u >= 0 && v >= 0 && u + v <= 1.0f;
    }
};

/* ---- libcore/triangle.cpp:69-166: clipped AABB in double precision --------------------------- */
struct P3d { double v[3]; };
inline int sutherlandHodgman(const P3d *in, int inCount, P3d *out, int axis, double splitPos, bool isMinimum) {
    if (inCount < 3) return 0;
    P3d cur = in[0];
    double sign = isMinimum ? 1.0 : -1.0;
    double distance = sign * (cur.v[axis] - splitPos);
    bool curIsInside = (distance >= 0);
    int outCount = 0;
    for (int i = 0; i < inCount; ++i) {
        int nextIdx = i + 1; if (nextIdx == inCount) nextIdx = 0;
        P3d next = in[nextIdx];
        distance = sign * (next.v[axis] - splitPos);
        bool nextIsInside = (distance >= 0);
        if (curIsInside && nextIsInside) {
            out[outCount++] = next;
        } else if (curIsInside != nextIsInside) {
            double t = (splitPos - cur.v[axis]) / (next.v[axis] - cur.v[axis]);
            P3d p;
            for (int j = 0; j < 3; ++j) p.v[j] = cur.v[j] + (next.v[j] - cur.v[j]) * t;
            p.v[axis] = splitPos;
            out[outCount++] = p;
            if (nextIsInside) out[outCount++] = next;
        }
        cur = next; curIsInside = nextIsInside;
    }
    return outCount;
}
inline AABB triClippedAABB(V3 p0, V3 p1, V3 p2, const AABB &aabb) {
    P3d v1[10], v2[10];
    const V3 ps[3] = { p0, p1, p2 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v1[i].v[j] = (double)ps[i][j];
    int n = 3;
    for (int axis = 0; axis < 3; ++axis) {
        n = sutherlandHodgman(v1, n, v2, axis, (double)aabb.min[axis], true);
        n = sutherlandHodgman(v2, n, v1, axis, (double)aabb.max[axis], false);
    }
    AABB result;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j) {
            double pd = v1[i].v[j];
            float pf = (float)pd, lo, hi;
            if (pf < pd) { lo = pf; hi = std::nextafterf(pf, std::numeric_limits<float>::infinity()); }
            else if (pf > pd) { hi = pf; lo = std::nextafterf(pf, -std::numeric_limits<float>::infinity()); }
            else lo = hi = pf;
            result.min[j] = std::min(result.min[j], lo);
            result.max[j] = std::max(result.max[j], hi);
        }
    }
    result.clip(aabb);
    return result;
}

/* ---- the flattened geometry the accelerators index -------------------------------------------- */
struct MeshView {
    const float *pos; const float *nrm; const float *uv; const uint32_t *idx;
    uint64_t nVerts, nTris;
    V3 P(uint32_t i) const { return V3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); }
    V3 N(uint32_t i) const { return V3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]); }
    V2 UV(uint32_t i) const { return V2{ uv[2 * i], uv[2 * i + 1] }; }
};

struct Hit { Float t, u, v; uint32_t prim; };
static const uint32_t kNoPrim = 0xFFFFFFFFu;

/* ---- SAH kd-tree (gkdtree.h / sahkdtree3.h, absent) ------------------------------------------ */
struct KDNode {               /* 8 bytes, as the reference's KDNode */
    uint32_t combined;        /* leaf: kLeafFlag | primStart ; inner: (leftChildIndex << 2) | axis */
    union { Float split; uint32_t primEnd; };
    bool isLeaf() const { return (combined & 0x80000000u) != 0; }
};
static const uint32_t kLeafFlag = 0x80000000u;

class KDTree {
public:
    std::vector<KDNode> nodes;
    std::vector<uint32_t> indices;
    std::vector<TriAccel> accel;
    AABB aabb, tightAABB;
    int maxDepthSeen = 0;

    static constexpr Float kTraversalCost = 15.f, kQueryCost = 20.f, kEmptyBonus = 0.9f;
    static constexpr int kStopPrims = 6, kMaxDepthCap = 48, kMinMaxBins = 128;
    static constexpr uint32_t kExactThreshold = 65536;

    void build(const MeshView &m, const std::vector<uint32_t> &shapeOfPrim, const std::vector<uint32_t> &firstPrimOfShape) {
        mesh = m;
        const uint32_t n = (uint32_t)m.nTris;
        accel.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            accel[i].load(m.P(m.idx[3 * i]), m.P(m.idx[3 * i + 1]), m.P(m.idx[3 * i + 2]));
            accel[i].shapeIndex = shapeOfPrim[i];
            accel[i].primIndex = i - firstPrimOfShape[shapeOfPrim[i]];
        }
        std::vector<uint32_t> prims(n);
        std::vector<AABB> boxes(n);
        tightAABB = AABB();
        for (uint32_t i = 0; i < n; ++i) {
            prims[i] = i;
            AABB b; b.expandBy(m.P(m.idx[3 * i])); b.expandBy(m.P(m.idx[3 * i + 1])); b.expandBy(m.P(m.idx[3 * i + 2]));
            boxes[i] = b; tightAABB.expandBy(b);
        }
        /* gkdtree.h: "slightly enlarge the bounding box" (MTS_KD_AABB_EPSILON = 1e-3 in single precision) */
        aabb = tightAABB;
        const Float eps = 1e-3f;
        V3 ext = aabb.max - aabb.min;
        aabb.min = aabb.min - (ext * eps + V3(eps));
        aabb.max = aabb.max + (ext * eps + V3(eps));
        int maxDepth = (int)(8 + 1.3f * std::log2((Float)std::max(n, 1u)));
        if (maxDepth > kMaxDepthCap) maxDepth = kMaxDepthCap;
        nodes.clear(); indices.clear();
        nodes.reserve(n / 2 + 16);
        nodes.push_back(KDNode());
        /* Subtrees of at most kTaskPrims primitives are independent of each other (a subtree is a function of its cell, its
         * primitive list, their clipped boxes, depth and bad-refine count), so the top of the tree is built here and the
         * subtrees below it on worker threads into fragments that are spliced in afterwards -- the reference builds in
         * parallel too (kdParallelBuild, scene.cpp:66-69).  Same tree as the sequential build, other node order in memory. */
        deferBelow = buildThreads() > 1 ? kTaskPrims : 0;
        tasks.clear();
        buildRec(0, aabb, prims, boxes, 0, maxDepth, 0);
        deferBelow = 0;
        if (!tasks.empty()) runTasks(boxes, maxDepth);
    }
    static unsigned buildThreads() {
        const char *e = std::getenv("KKORACLE_BUILD_THREADS");
        unsigned t = e ? (unsigned)std::atoi(e) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1u, t);
    }

    /* stats (nullable): [0] += visited nodes, [1] += tested triangles (N_nodes / N_tris of SURVEY 8d) */
    template <bool shadowRay> bool traverse(const Ray &ray, Float mint, Float maxt, Hit &hit, uint64_t *stats) const;

    /* librender/skdtree.cpp:112-142 */
    bool rayIntersectClosest(const Ray &ray, Hit &hit, uint64_t *stats = nullptr) const {
        hit.t = std::numeric_limits<Float>::infinity(); hit.prim = kNoPrim; hit.u = hit.v = 0;
        Float mint, maxt;
        if (aabb.rayIntersect(ray.o, ray.d, ray.dRcp, mint, maxt)) {
            Float rayMinT = ray.mint;
            if (rayMinT == kEpsilon)
                rayMinT *= std::max(std::max(std::max(std::fabs(ray.o.x), std::fabs(ray.o.y)), std::fabs(ray.o.z)), kEpsilon);
            if (rayMinT > mint) mint = rayMinT;
            if (ray.maxt < maxt) maxt = ray.maxt;
            if (maxt > mint)
                return traverse<false>(ray, mint, maxt, hit, stats);
        }
        return false;
    }
    /* librender/skdtree.cpp:207-226 */
    bool rayIntersectAny(const Ray &ray, uint64_t *stats = nullptr) const {
        Float mint, maxt;
        Hit hit;
        if (aabb.rayIntersect(ray.o, ray.d, ray.dRcp, mint, maxt)) {
            Float rayMinT = ray.mint;
            if (rayMinT == kEpsilon)
                rayMinT *= std::max(std::max(std::fabs(ray.o.x), std::fabs(ray.o.y)), std::fabs(ray.o.z));
            if (rayMinT > mint) mint = rayMinT;
            if (ray.maxt < maxt) maxt = ray.maxt;
            if (maxt > mint)
                return traverse<true>(ray, mint, maxt, hit, stats);
        }
        return false;
    }
    /* Tree-independent ground truth: every triangle in primitive order with the same TriAccel
       arithmetic, the same interval handling and the same "equal t replaces" rule. */
    bool bruteForce(const Ray &ray, bool anyHit, Hit &hit) const {
        hit.t = std::numeric_limits<Float>::infinity(); hit.prim = kNoPrim; hit.u = hit.v = 0;
        Float mint, maxt;
        if (!aabb.rayIntersect(ray.o, ray.d, ray.dRcp, mint, maxt)) return false;
        Float rayMinT = ray.mint;
        if (rayMinT == kEpsilon) {
            Float m = std::max(std::max(std::fabs(ray.o.x), std::fabs(ray.o.y)), std::fabs(ray.o.z));
            if (!anyHit) m = std::max(m, kEpsilon);
            rayMinT *= m;
        }
        if (rayMinT > mint) mint = rayMinT;
        if (ray.maxt < maxt) maxt = ray.maxt;
        if (!(maxt > mint)) return false;
        bool found = false;
        for (uint32_t i = 0; i < (uint32_t)accel.size(); ++i) {
            Float tu, tv, tt;
            if (accel[i].rayIntersect(ray, mint, maxt, tu, tv, tt)) {
                found = true; maxt = tt;
                hit.t = tt; hit.u = tu; hit.v = tv; hit.prim = i;
                if (anyHit) return true;
            }
        }
        return found;
    }

private:
    MeshView mesh;
    struct Event { Float pos; uint32_t prim; uint8_t type; /* 0 = end, 1 = planar, 2 = start */ };
    static constexpr uint32_t kTaskPrims = 131072;
    struct Task { uint32_t nodeIdx; AABB cell; std::vector<uint32_t> prims; std::vector<AABB> primBoxes; int depth, badRefines; };
    std::vector<Task> tasks;
    uint32_t deferBelow = 0;      /* > 0 while the top of the tree is built: subtrees of <= that many primitives become tasks */

    void runTasks(const std::vector<AABB> &boxes, int maxDepth) {
        std::vector<KDTree> frags(tasks.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= tasks.size()) return;
                Task &t = tasks[k]; KDTree &f = frags[k];
                f.mesh = mesh;
                /* a private box table for the task's primitives (the recursion clips boxes in place), addressed by local id */
                std::vector<uint32_t> local(t.prims.size());
                for (uint32_t i = 0; i < (uint32_t)local.size(); ++i) local[i] = i;
                f.globalId = t.prims;
                f.nodes.push_back(KDNode());
                f.buildRec(0, t.cell, local, t.primBoxes, t.depth, maxDepth, t.badRefines);
                std::vector<AABB>().swap(t.primBoxes);
            }
        };
        std::vector<std::thread> pool;
        const unsigned nt = std::min<unsigned>(buildThreads(), (unsigned)tasks.size());
        for (unsigned i = 0; i < nt; ++i) pool.emplace_back(worker);
        for (auto &th : pool) th.join();
        for (size_t k = 0; k < tasks.size(); ++k) {      /* splice: fragment node 0 -> the task's node, the rest appended */
            const KDTree &f = frags[k];
            const uint32_t nodeOff = (uint32_t)nodes.size() - 1u, idxOff = (uint32_t)indices.size();
            auto fix = [&](KDNode nd) {
                if (nd.isLeaf()) { nd.combined = ((nd.combined & ~kLeafFlag) + idxOff) | kLeafFlag; nd.primEnd += idxOff; }
                else nd.combined = ((((nd.combined >> 2) + nodeOff)) << 2) | (nd.combined & 3u);
                return nd;
            };
            nodes[tasks[k].nodeIdx] = fix(f.nodes[0]);
            for (size_t j = 1; j < f.nodes.size(); ++j) nodes.push_back(fix(f.nodes[j]));
            indices.insert(indices.end(), f.indices.begin(), f.indices.end());
            maxDepthSeen = std::max(maxDepthSeen, f.maxDepthSeen);
        }
        tasks.clear();
    }
    std::vector<uint32_t> globalId;   /* fragment builds: local primitive id -> primitive; empty = identity */
    uint32_t gid(uint32_t p) const { return globalId.empty() ? p : globalId[p]; }

    void makeLeaf(uint32_t nodeIdx, const std::vector<uint32_t> &prims, int depth) {
        maxDepthSeen = std::max(maxDepthSeen, depth);
        KDNode &nd = nodes[nodeIdx];
        nd.combined = (uint32_t)indices.size() | kLeafFlag;
        for (uint32_t p : prims) indices.push_back(gid(p));
        nd.primEnd = (uint32_t)indices.size();
    }

    void buildRec(uint32_t nodeIdx, const AABB &box, std::vector<uint32_t> &prims, std::vector<AABB> &boxes,
                  int depth, int maxDepth, int badRefines) {
        const uint32_t n = (uint32_t)prims.size();
        if (n <= (uint32_t)kStopPrims || depth >= maxDepth) { makeLeaf(nodeIdx, prims, depth); return; }
        if (deferBelow && n <= deferBelow) {
            Task t; t.nodeIdx = nodeIdx; t.cell = box; t.depth = depth; t.badRefines = badRefines;
            t.primBoxes.resize(n);
            for (uint32_t i = 0; i < n; ++i) t.primBoxes[i] = boxes[prims[i]];
            t.prims.swap(prims);
            tasks.push_back(std::move(t));
            return;
        }
        const Float invSA = 1.0f / box.surfaceArea();
        const Float leafCost = kQueryCost * n;
        int bestAxis = -1; Float bestPos = 0, bestCost = std::numeric_limits<Float>::infinity();
        bool planarLeft = true;
        V3 ext = box.max - box.min;
        if (n > kExactThreshold) {
            /* min-max binning (Shevtsov et al.), kMinMaxBins bins per axis */
            for (int axis = 0; axis < 3; ++axis) {
                if (!(ext[axis] > 0)) continue;
                std::vector<uint32_t> minBin(kMinMaxBins, 0), maxBin(kMinMaxBins, 0);
                const Float invBin = kMinMaxBins / ext[axis];
                for (uint32_t i = 0; i < n; ++i) {
                    const AABB &b = boxes[prims[i]];
                    int lo = std::min(kMinMaxBins - 1, std::max(0, (int)((b.min[axis] - box.min[axis]) * invBin)));
                    int hi = std::min(kMinMaxBins - 1, std::max(0, (int)((b.max[axis] - box.min[axis]) * invBin)));
                    minBin[lo]++; maxBin[hi]++;
                }
                uint32_t nL = 0, nR = n;
                for (int b = 0; b < kMinMaxBins - 1; ++b) {
                    nL += minBin[b]; nR -= maxBin[b];
                    Float pos = box.min[axis] + (b + 1) * ext[axis] / kMinMaxBins;
                    Float c = sahCost(box, axis, pos, nL, nR, invSA);
                    if (c < bestCost) { bestCost = c; bestAxis = axis; bestPos = pos; }
                }
            }
        } else {
            /* exact sweep over clipped-AABB events */
            std::vector<Event> ev;
            for (int axis = 0; axis < 3; ++axis) {
                if (!(ext[axis] > 0)) continue;
                ev.clear(); ev.reserve(2 * n);
                for (uint32_t i = 0; i < n; ++i) {
                    const AABB &b = boxes[prims[i]];
                    if (b.min[axis] == b.max[axis]) ev.push_back(Event{ b.min[axis], prims[i], 1 });
                    else { ev.push_back(Event{ b.min[axis], prims[i], 2 }); ev.push_back(Event{ b.max[axis], prims[i], 0 }); }
                }
                std::sort(ev.begin(), ev.end(), [](const Event &a, const Event &b) {
                    if (a.pos != b.pos) return a.pos < b.pos;
                    return a.type < b.type;
                });
                uint32_t nL = 0, nR = n;
                size_t i = 0;
                while (i < ev.size()) {
                    Float pos = ev[i].pos;
                    uint32_t pEnd = 0, pPlanar = 0, pStart = 0;
                    while (i < ev.size() && ev[i].pos == pos && ev[i].type == 0) { ++pEnd; ++i; }
                    while (i < ev.size() && ev[i].pos == pos && ev[i].type == 1) { ++pPlanar; ++i; }
                    while (i < ev.size() && ev[i].pos == pos && ev[i].type == 2) { ++pStart; ++i; }
                    nR -= pPlanar + pEnd;
                    if (pos > box.min[axis] && pos < box.max[axis]) {
                        Float cL = sahCost(box, axis, pos, nL + pPlanar, nR, invSA);
                        Float cR = sahCost(box, axis, pos, nL, nR + pPlanar, invSA);
                        if (cL < bestCost) { bestCost = cL; bestAxis = axis; bestPos = pos; planarLeft = true; }
                        if (cR < bestCost) { bestCost = cR; bestAxis = axis; bestPos = pos; planarLeft = false; }
                    }
                    nL += pStart + pPlanar;
                }
            }
        }
        if (bestAxis < 0) { makeLeaf(nodeIdx, prims, depth); return; }
        if (bestCost >= leafCost) {
            /* "bad refine": keep splitting a few times (maxBadRefines = 3) unless hopeless */
            if ((bestCost > 4 * leafCost && n < 16) || badRefines >= 3) { makeLeaf(nodeIdx, prims, depth); return; }
            ++badRefines;
        }
        AABB lBox = box, rBox = box;
        lBox.max[bestAxis] = bestPos; rBox.min[bestAxis] = bestPos;
        std::vector<uint32_t> lp, rp;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t p = prims[i];
            const AABB &b = boxes[p];
            if (b.min[bestAxis] == b.max[bestAxis] && b.min[bestAxis] == bestPos) {
                (planarLeft ? lp : rp).push_back(p);
            } else if (b.max[bestAxis] <= bestPos) lp.push_back(p);
            else if (b.min[bestAxis] >= bestPos) rp.push_back(p);
            else { lp.push_back(p); rp.push_back(p); }
        }
        if (lp.size() == n && rp.size() == n) { makeLeaf(nodeIdx, prims, depth); return; }
        { std::vector<uint32_t>().swap(prims); }
        const uint32_t left = (uint32_t)nodes.size();
        nodes.push_back(KDNode()); nodes.push_back(KDNode());
        nodes[nodeIdx].split = bestPos;
        nodes[nodeIdx].combined = (left << 2) | (uint32_t)bestAxis;
        const bool exact = n <= kExactThreshold;
        /* perfect splits: re-clip straddling triangles to each child cell (sahkdtree3.h 'clip') */
        buildChild(left, lBox, lp, boxes, depth, maxDepth, badRefines, exact);
        buildChild(left + 1, rBox, rp, boxes, depth, maxDepth, badRefines, exact);
    }

    void buildChild(uint32_t nodeIdx, const AABB &cell, std::vector<uint32_t> &prims, std::vector<AABB> &boxes,
                    int depth, int maxDepth, int badRefines, bool exact) {
        std::vector<std::pair<uint32_t, AABB>> saved;
        std::vector<uint32_t> kept; kept.reserve(prims.size());
        for (uint32_t p : prims) {
            const AABB &b = boxes[p];
            bool inside = true;
            for (int a = 0; a < 3; ++a) if (b.min[a] < cell.min[a] || b.max[a] > cell.max[a]) inside = false;
            if (inside) { kept.push_back(p); continue; }
            AABB nb;
            if (exact) {
                const uint32_t g = gid(p);
                nb = triClippedAABB(mesh.P(mesh.idx[3 * g]), mesh.P(mesh.idx[3 * g + 1]), mesh.P(mesh.idx[3 * g + 2]), cell);
                if (!nb.isValid()) continue;      /* clipped away entirely */
            } else { nb = b; nb.clip(cell); }
            saved.push_back(std::make_pair(p, b));
            boxes[p] = nb;
            kept.push_back(p);
        }
        { std::vector<uint32_t>().swap(prims); }
        buildRec(nodeIdx, cell, kept, boxes, depth + 1, maxDepth, badRefines);
        for (auto &s : saved) boxes[s.first] = s.second;
    }

    static Float sahCost(const AABB &box, int axis, Float pos, uint32_t nL, uint32_t nR, Float invSA) {
        AABB l = box, r = box; l.max[axis] = pos; r.min[axis] = pos;
        Float pL = l.surfaceArea() * invSA, pR = r.surfaceArea() * invSA;
        Float c = kTraversalCost + kQueryCost * (pL * nL + pR * nR);
        if (nL == 0 || nR == 0) c *= kEmptyBonus;
        return c;
    }
};

/* Havran's TA^B_rec (gkdtree.h rayIntersectHavran, absent).  The leaf test uses the ray's
 * own [mint,maxt]; a hit shrinks maxt; traversal stops once the current cell's exit lies beyond maxt.
 * A later primitive with equal t replaces an earlier one (the triangle test rejects only t > maxt). */
template <bool shadowRay>
bool KDTree::traverse(const Ray &ray, Float mint, Float maxt, Hit &hit, uint64_t *stats) const {
    struct Entry { const KDNode *node; Float t; V3 p; uint32_t prev; };
    Entry stack[kMaxDepthCap + 4];
    uint32_t enPt = 0;
    stack[enPt].t = mint; stack[enPt].p = ray.at(mint);
    uint32_t exPt = 1;
    stack[exPt].t = maxt; stack[exPt].p = ray.at(maxt); stack[exPt].node = nullptr;
    bool found = false;
    const KDNode *base = nodes.data();
    const KDNode *cur = base;
    while (cur != nullptr) {
        while (!cur->isLeaf()) {
            if (stats) ++stats[0];
            const Float splitVal = cur->split;
            const int axis = (int)(cur->combined & 3u);
            const KDNode *left = base + (cur->combined >> 2);
            const KDNode *farChild;
            if (stack[enPt].p[axis] <= splitVal) {
                if (stack[exPt].p[axis] <= splitVal) { cur = left; continue; }
                if (stack[enPt].p[axis] == splitVal) { cur = left + 1; continue; }
                cur = left; farChild = left + 1;
            } else {
                if (splitVal < stack[exPt].p[axis]) { cur = left + 1; continue; }
                farChild = left; cur = left + 1;
            }
            Float distToSplit = (splitVal - ray.o[axis]) * ray.dRcp[axis];
            const uint32_t tmp = exPt++;
            if (exPt == enPt) ++exPt;
            stack[exPt].prev = tmp;
            stack[exPt].t = distToSplit;
            stack[exPt].node = farChild;
            stack[exPt].p = ray.at(distToSplit);
            stack[exPt].p[axis] = splitVal;
        }
        if (stats) ++stats[0];
        const uint32_t start = cur->combined & ~kLeafFlag, end = cur->primEnd;
        for (uint32_t e = start; e != end; ++e) {
            const uint32_t primIdx = indices[e];
            Float tu, tv, tt;
            if (stats) ++stats[1];
            if (accel[primIdx].rayIntersect(ray, mint, maxt, tu, tv, tt)) {
                if (shadowRay) return true;
                maxt = tt; found = true;
                hit.t = tt; hit.u = tu; hit.v = tv; hit.prim = primIdx;
            }
        }
        if (stack[exPt].t > maxt) break;
        enPt = exPt;
        cur = stack[exPt].node;
        exPt = stack[enPt].prev;
    }
    return found;
}

} /* namespace ora */
