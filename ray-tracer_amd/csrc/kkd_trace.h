/*
 * kkd_trace.h -- gfx950 ray traversal: one ray per lane, 4-wide BVH, per-lane stack in LDS,
 * watertight triangle test (Woop, Benthin, Wald 2013).
 *
 * Replaces ShapeKDTree::rayIntersect(ray, its) / rayIntersect(ray) of the reference
 * (src/librender/skdtree.cpp:112-142, 207-226: AABB clip, adaptive ray epsilon, Havran kd-tree walk,
 * TriAccel projection test).  Kept from the reference: the adaptive epsilon rule, the [mint, maxt]
 * acceptance interval, the hit record (t, barycentrics of vertices 1 and 2, primitive).
 *
 * Memory shape: a node visit is four 16-byte loads of one 64-byte record (Node4, 8-bit child boxes), a triangle test is
 * three 16-byte loads (LeafTri); the first STACK entries of a lane's stack live in LDS (index = slot * BLOCK +
 * thread, so the two 32-lane halves of a wave never collide on a bank) and only deeper entries, which the SAH
 * trees of the configs almost never reach, spill to a per-lane private array of kStackOverflow words.
 */
#pragma once
#include "kkd_types.h"

namespace kkd {

#ifndef KKD_STACK_LDS
#define KKD_STACK_LDS 16      /* 4 ... 16 (parked rays carry these words: TraceContinuation holds 16 per record) */
#endif
static constexpr int kStackLds = KKD_STACK_LDS;   /* LDS-resident stack entries per lane           */
/* Shadow rays stop at their first hit and seldom stack deep; every KB of LDS a workgroup does not take shows in the two-lane
 * job (profiles/r3_ab_lds_stack.txt: 12 entries: C4 shadow stage 306 -> 301 ms, C2 29.1 -> 27.7 ms; 8 entries: C4 334 ms, the private
 * part gets used; 12 for ALL walks: BSDF stage 433 -> 440 ms) */
#ifndef KKD_STACK_LDS_ANY
#define KKD_STACK_LDS_ANY 12
#endif
static constexpr int kStackLdsAny = KKD_STACK_LDS_ANY;   /* the same for the occlusion walk of the render path and of kkd_trace* */
static constexpr int kStackOverflow = 80;   /* further entries in private (scratch) memory   */

__device__ __forceinline__ float selc(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

/* Per-ray constants of the watertight test */
struct WoopRay {
    int kx, ky, kz; float Sx, Sy, Sz;
    __device__ __forceinline__ void init(float dx, float dy, float dz) {
        const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
        kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
        kx = kz + 1; if (kx == 3) kx = 0;
        ky = kx + 1; if (ky == 3) ky = 0;
        const float dkz = selc(dx, dy, dz, kz);
        if (dkz < 0.0f) { int t = kx; kx = ky; ky = t; }
        Sx = selc(dx, dy, dz, kx) / dkz;
        Sy = selc(dx, dy, dz, ky) / dkz;
        Sz = 1.0f / dkz;
    }
};

/* returns true and updates (t,u,v) if the triangle is hit with tmin <= t <= tmax.
 * No fused multiply-adds in the edge functions: shared edges must evaluate to exact negatives. */
__device__ __forceinline__ bool woopIntersect(const WoopRay &w, float ox, float oy, float oz,
                                              const float4 &a, const float4 &b, const float4 &c,
                                              float tmin, float tmax, float &tOut, float &uOut, float &vOut) {
#pragma clang fp contract(off)   /* __fmul_rn / __fsub_rn alone do not stop hipcc from fusing across them */
    const float Ax0 = a.x - ox, Ay0 = a.y - oy, Az0 = a.z - oz;
    const float Bx0 = b.x - ox, By0 = b.y - oy, Bz0 = b.z - oz;
    const float Cx0 = c.x - ox, Cy0 = c.y - oy, Cz0 = c.z - oz;
    const float Akz = selc(Ax0, Ay0, Az0, w.kz), Bkz = selc(Bx0, By0, Bz0, w.kz), Ckz = selc(Cx0, Cy0, Cz0, w.kz);
    const float Ax = __fsub_rn(selc(Ax0, Ay0, Az0, w.kx), __fmul_rn(w.Sx, Akz));
    const float Ay = __fsub_rn(selc(Ax0, Ay0, Az0, w.ky), __fmul_rn(w.Sy, Akz));
    const float Bx = __fsub_rn(selc(Bx0, By0, Bz0, w.kx), __fmul_rn(w.Sx, Bkz));
    const float By = __fsub_rn(selc(Bx0, By0, Bz0, w.ky), __fmul_rn(w.Sy, Bkz));
    const float Cx = __fsub_rn(selc(Cx0, Cy0, Cz0, w.kx), __fmul_rn(w.Sx, Ckz));
    const float Cy = __fsub_rn(selc(Cx0, Cy0, Cz0, w.ky), __fmul_rn(w.Sy, Ckz));
    float U = __fsub_rn(__fmul_rn(Cx, By), __fmul_rn(Cy, Bx));
    float V = __fsub_rn(__fmul_rn(Ax, Cy), __fmul_rn(Ay, Cx));
    float W = __fsub_rn(__fmul_rn(Bx, Ay), __fmul_rn(By, Ax));
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float T = U * (w.Sz * Akz) + V * (w.Sz * Bkz) + W * (w.Sz * Ckz);
    /* v_rcp_f32 (1 ulp) instead of a correctly rounded division (~10 VALU instructions in a kernel that is bound by VALU
     * issue): which triangle is hit was decided above by the signs of U, V, W; only the last bit of t, u, v can move */
    const float rcp = __builtin_amdgcn_rcpf(det);
    const float t = T * rcp;
    if (!(t >= tmin && t <= tmax)) return false;
    tOut = t; uOut = V * rcp; vOut = W * rcp;
    return true;
}

/* KKD_FLAG_PROJECTION_TEST: TriAccel::rayIntersect of the reference (Wald's projection test; triaccel.h, called from the
 * leaf loop skdtree.cpp:294-300 / gkdtree.h).  Same operations in the same order, each rounded on its own (no fma
 * contraction, IEEE division), so a triangle is hit or missed, and gets the (t, u, v), exactly as on the CPU. */
__device__ __forceinline__ bool projIntersect(float ox, float oy, float oz, float dx, float dy, float dz,
                                              const float4 &r0, const float4 &r1, const float4 &r2,
                                              float tmin, float tmax, float &tOut, float &uOut, float &vOut) {
#pragma clang fp contract(off)   /* __fmul_rn / __fadd_rn alone do not stop hipcc from fusing across them */
    const uint32_t k = __float_as_uint(r0.w);
    if (k > 2u) return false;
    const float o_u = k == 0u ? oy : (k == 1u ? oz : ox), o_v = k == 0u ? oz : (k == 1u ? ox : oy), o_k = k == 0u ? ox : (k == 1u ? oy : oz);
    const float d_u = k == 0u ? dy : (k == 1u ? dz : dx), d_v = k == 0u ? dz : (k == 1u ? dx : dy), d_k = k == 0u ? dx : (k == 1u ? dy : dz);
    const float n_u = r0.x, n_v = r0.y, n_d = r0.z, a_u = r1.x, a_v = r1.y, b_nu = r1.z, b_nv = r1.w, c_nu = r2.x, c_nv = r2.y;
    const float num = __fsub_rn(__fsub_rn(__fsub_rn(n_d, __fmul_rn(o_u, n_u)), __fmul_rn(o_v, n_v)), o_k);
    const float den = __fadd_rn(__fadd_rn(__fmul_rn(d_u, n_u), __fmul_rn(d_v, n_v)), d_k);
    const float t = __fdiv_rn(num, den);
    if (t < tmin || t > tmax) return false;
    const float hu = __fsub_rn(__fadd_rn(o_u, __fmul_rn(t, d_u)), a_u), hv = __fsub_rn(__fadd_rn(o_v, __fmul_rn(t, d_v)), a_v);
    const float u = __fadd_rn(__fmul_rn(hv, b_nu), __fmul_rn(hu, b_nv));
    const float v = __fadd_rn(__fmul_rn(hu, c_nu), __fmul_rn(hv, c_nv));
    if (!(u >= 0.f && v >= 0.f && __fadd_rn(u, v) <= 1.0f)) return false;
    tOut = t; uOut = u; vOut = v;
    return true;
}

#ifdef KKD_TRI_LDS
/* Experiment of round 3 ("LDS-rotated triangle test"): the ray's origin and direction sit in LDS as (x, y, z, x, y) per lane,
 * so the components the triangle's axis k selects -- (k, k + 1, k + 2) mod 3 -- are three reads at fixed offsets from one
 * address instead of twelve selects.  sr = the lane's column (stride BLOCK floats). */
template <int BLOCK>
__device__ __forceinline__ bool projIntersectLds(const float *sr, const float4 &r0, const float4 &r1, const float4 &r2,
                                                 float tmin, float tmax, float &tOut, float &uOut, float &vOut) {
#pragma clang fp contract(off)
    const uint32_t k = __float_as_uint(r0.w);
    if (k > 2u) return false;
    const float *o = sr + k * BLOCK, *d = sr + (5u + k) * BLOCK;
    const float o_k = o[0], o_u = o[BLOCK], o_v = o[2 * BLOCK], d_k = d[0], d_u = d[BLOCK], d_v = d[2 * BLOCK];
    const float n_u = r0.x, n_v = r0.y, n_d = r0.z, a_u = r1.x, a_v = r1.y, b_nu = r1.z, b_nv = r1.w, c_nu = r2.x, c_nv = r2.y;
    const float num = __fsub_rn(__fsub_rn(__fsub_rn(n_d, __fmul_rn(o_u, n_u)), __fmul_rn(o_v, n_v)), o_k);
    const float den = __fadd_rn(__fadd_rn(__fmul_rn(d_u, n_u), __fmul_rn(d_v, n_v)), d_k);
    const float t = __fdiv_rn(num, den);
    if (t < tmin || t > tmax) return false;
    const float hu = __fsub_rn(__fadd_rn(o_u, __fmul_rn(t, d_u)), a_u), hv = __fsub_rn(__fadd_rn(o_v, __fmul_rn(t, d_v)), a_v);
    const float u = __fadd_rn(__fmul_rn(hv, b_nu), __fmul_rn(hu, b_nv));
    const float v = __fadd_rn(__fmul_rn(hu, c_nu), __fmul_rn(hv, c_nv));
    if (!(u >= 0.f && v >= 0.f && __fadd_rn(u, v) <= 1.0f)) return false;
    tOut = t; uOut = u; vOut = v;
    return true;
}
#endif

/* adaptive ray epsilon, skdtree.cpp:126-129 (closest hit) and :213-214 (shadow rays: no inner clamp) */
template <bool ANY> __device__ __forceinline__ float adaptiveMinT(float mint, float ox, float oy, float oz) {
    if (mint == kEpsilon) {
        float m = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
        if (!ANY) m = fmaxf(m, kEpsilon);
        mint *= m;
    }
    return mint;
}

/* The per-lane walk as a resumable object: begin() derives the per-ray constants, run() advances every lane of the wave by
 * at most iterCap steps (all state -- current reference, stack pointer, clipped interval, best hit -- stays in the
 * object and in the lane's stack), so a kernel can stop a wave whose last few lanes are still walking, park those lanes
 * in memory and let a later, densely packed launch finish them (kTraceFirst / kTraceResume). */
/* A lane that reaches a leaf does not spend an iteration on it: the leaf becomes its PENDING leaf, the lane takes the next
 * entry of its stack at once, and in the following iterations it takes a node step AND tests one triangle of the pending leaf
 * (round 3: C4 one-lane BSDF / shadow stage 460 / 326 -> 449 / 313 ms).  A lane that reaches a second leaf while one is pending
 * waits for the triangle step.
 * KKD_TRI_GATE > 0 (developer switch): "speculative" traversal after Aila & Laine 2009 -- the triangle step of an iteration
 * is issued only when at least that many lanes hold a pending leaf (or KKD_TRI_GATE_BLOCKED lanes are blocked at a second
 * leaf, or no lane can take a node step), so that its ~90 instructions run for 20-30 lanes instead of ~10.  Built and
 * measured in round 3 (C4 one-lane BSDF / shadow ms): gate 12 / blocked 6: 459 / 317; 20 / 4: 457 / 316; 20 / 8: 467 / 321;
 * 28 / 12: 490 / 332, against 449 / 313 without a gate: a wave runs as long as its longest ray, every iteration that ray waits
 * for the gate lengthens the wave, and that costs more than the triangle steps saved.  Default 0. */
#ifndef KKD_TRI_GATE
#define KKD_TRI_GATE 0
#endif
#ifndef KKD_TRI_GATE_BLOCKED
#define KKD_TRI_GATE_BLOCKED 8
#endif
template <bool ANY, bool COUNT, bool PROJ, int STACK, int BLOCK>
struct Walk {
    /* (the private part of the stack, kStackOverflow words, is the caller's array: an array member would pin the whole
     * object in scratch memory) */
    DHit hit; uint32_t hprim;                        /* hprim: primitive id of the best hit (equal-distance ties) */
    float ox, oy, oz, dx, dy, dz, tmin, tmax, idx, idy, idz; WoopRay wr; uint32_t cur, pend; int sp;
    int col;                                         /* the lane's column of the LDS stack: its thread index, unless the ray was moved to another lane (KKD_REPACK_AT) */

    __device__ __forceinline__ void begin(const DRay &ray) {
        hit.t = kInf; hit.u = 0.f; hit.v = 0.f; hit.id = kNoHit; hprim = 0u;
        ox = ray.ox; oy = ray.oy; oz = ray.oz; dx = ray.dx; dy = ray.dy; dz = ray.dz;
        tmin = adaptiveMinT<ANY>(ray.mint, ox, oy, oz);
        tmax = ray.maxt;
        /* A zero direction component would give 1/d = inf and 0 * inf = NaN for an origin lying exactly on a slab plane
         * (rays aimed along an axis at mesh vertices do that); a huge finite reciprocal keeps such a slab "inside". */
        const float kBig = 0x1p100f, kTiny = 0x1p-100f;
        idx = fabsf(dx) < kTiny ? copysignf(kBig, dx) : 1.0f / dx;
        idy = fabsf(dy) < kTiny ? copysignf(kBig, dy) : 1.0f / dy;
        idz = fabsf(dz) < kTiny ? copysignf(kBig, dz) : 1.0f / dz;
        if (!PROJ) wr.init(dx, dy, dz);
        sp = 0; pend = 0u; col = (int)threadIdx.x;
        cur = (tmax > tmin) ? 0u : kTraversalDone;      /* cur: inner node index | leaf reference (kLeafBit) | kTraversalDone */
    }
    /* pend: the leaf whose triangles are still to be tested (a leaf reference, 0 = none) */
    __device__ __forceinline__ bool finished() const { return cur == kTraversalDone && pend == 0u; }
    __device__ __forceinline__ void stop() { cur = kTraversalDone; pend = 0u; }
    /* the state a parked ray was left in (its stack words are the caller's to restore); the best hit's distance is tmax */
    __device__ __forceinline__ void resume(uint32_t cur_, uint32_t pend_, int sp_, float tmax_, float u, float v, uint32_t id, uint32_t prim) {
        cur = cur_; pend = pend_; sp = sp_; tmax = tmax_;
        hit.t = id != kNoHit ? tmax_ : kInf; hit.u = u; hit.v = v; hit.id = id; hprim = prim;
    }

    __device__ __forceinline__ void run(const Node4 *__restrict__ nodes, const void *__restrict__ tris, uint32_t *sstack, uint32_t *ovf,
                                        uint32_t &nNodes, uint32_t &nTris, int iterCap) {
        const bool negx = idx < 0.f, negy = idy < 0.f, negz = idz < 0.f;
        const uint4 *nodes4 = reinterpret_cast<const uint4 *>(nodes);
        const float4 *tris4 = reinterpret_cast<const float4 *>(tris);
        const int tid = col;
#ifdef KKD_TRI_LDS
        __shared__ float sray[10 * BLOCK];
        if (PROJ) {   /* only this lane reads its column: no barrier */
            float *sr = sray + tid;
            sr[0] = ox; sr[BLOCK] = oy; sr[2 * BLOCK] = oz; sr[3 * BLOCK] = ox; sr[4 * BLOCK] = oy;
            sr[5 * BLOCK] = dx; sr[6 * BLOCK] = dy; sr[7 * BLOCK] = dz; sr[8 * BLOCK] = dx; sr[9 * BLOCK] = dy;
        }
#endif
#define KKD_PUSH(V) { if (sp < STACK) sstack[sp * BLOCK + tid] = (V); else ovf[sp - STACK] = (V); ++sp; }
/* the LDS read is unconditional and the private read a separate, rarely taken branch: one select between the two
 * pointers would turn every pop into a flat load, which goes the long way round through the texture addresser */
#define KKD_POP() { if (sp == 0) cur = kTraversalDone; else { --sp; cur = sstack[(sp < STACK ? sp : STACK - 1) * BLOCK + tid]; \
                    if (sp >= STACK) cur = ovf[sp - STACK]; } }
        /* One step per iteration and lane: a node visit, and -- when the wave takes a triangle step -- ONE triangle test of the
         * lane's pending leaf, so the iteration costs at most one node visit plus one triangle test, not the longest leaf of the
         * wave.  Every wave leaves this loop: the walk is finite on a tree, a wave in which no lane can take a node step takes
         * triangle steps until a lane can, and the step bound ends it on a malformed tree. */
        for (int guard = iterCap; guard > 0; --guard) {
            const bool atNode = !(cur & kLeafBit);
            if (__ballot(atNode || pend != 0u || cur != kTraversalDone) == 0ull) break;
            if (atNode) {
                if (COUNT) ++nNodes;
#ifndef KKD_OFS64   /* 64-bit record addresses: developer switch (round 3 measured the 32-bit form 1 % faster in the traversal stages) */
    /* 32-bit byte offsets from a scalar base (global_load ... v_off, s[base]): one shift per record address instead of 64-bit
     * arithmetic; x * 48 and x * 80 as shift-adds (a 32-bit multiply is quarter rate).  The upload refuses trees whose node or
     * triangle array exceeds 4 GiB (kMaxAccelBytes: 67 M nodes, 89 M triangles). */
    #define KKD_REC(BASE, INDEX, BYTES) reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(BASE) + (size_t)( \
        (BYTES) == 48 ? ((((uint32_t)(INDEX) << 1) + (uint32_t)(INDEX)) << 4) : (BYTES) == 80 ? ((((uint32_t)(INDEX) << 2) + (uint32_t)(INDEX)) << 4) : ((uint32_t)(INDEX) * (uint32_t)(BYTES))))
#else
    #define KKD_REC(BASE, INDEX, BYTES) ((BASE) + (size_t)(INDEX) * ((BYTES) / 16))
#endif
#ifdef KKD_NODE_FP16
                const uint4 *n = KKD_REC(nodes4, cur, 80);
                const uint4 hd = n[0], ch = n[1], ha = n[2], hb = n[3], hc = n[4];
#else
                const uint4 *n = KKD_REC(nodes4, cur, 64);
                const uint4 hd = n[0], ch = n[1], qa = n[2], qb = n[3];
#endif
                const float gx = __uint_as_float(hd.x), gy = __uint_as_float(hd.y), gz = __uint_as_float(hd.z);
                const float sx = __uint_as_float((hd.w & 0xffu) << 23), sy = __uint_as_float(((hd.w >> 8) & 0xffu) << 23),
                            sz = __uint_as_float(((hd.w >> 16) & 0xffu) << 23);
                float d0, d1, d2, d3;
                /* Slab test in the node's grid: t(plane q) = (origin + q * scale - o) / d = q * ta + tb, so a plane costs one
                 * byte->float conversion and one fma.  The near / far plane of each axis is picked by the sign of the ray
                 * direction for all four children at once (one select of the packed byte quadruple).  The 2e-6 slack on the
                 * far side covers the rounding of ta, tb (the builder's enclosure check covers the grid itself; one multiply
                 * instead of one on each side: tn >= 0 on this path). */
                const float tax = sx * idx, tay = sy * idy, taz = sz * idz;
                const float tbx = (gx - ox) * idx, tby = (gy - oy) * idy, tbz = (gz - oz) * idz;
#ifdef KKD_NODE_FP16
                /* ha = minx | miny, hb = minz | maxx, hc = maxy | maxz: two words of two halves per axis and side */
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                const uint32_t nxA = negx ? hb.z : ha.x, nxB = negx ? hb.w : ha.y, fxA = negx ? ha.x : hb.z, fxB = negx ? ha.y : hb.w;
                const uint32_t nyA = negy ? hc.x : ha.z, nyB = negy ? hc.y : ha.w, fyA = negy ? ha.z : hc.x, fyB = negy ? ha.w : hc.y;
                const uint32_t nzA = negz ? hc.z : hb.x, nzB = negz ? hc.w : hb.y, fzA = negz ? hb.x : hc.z, fzB = negz ? hb.y : hc.w;
    #define KKD_H(W, I) ((float)__builtin_bit_cast(half2v, W)[I])
    #define KKD_SLAB(D, NX, FX, NY, FY, NZ, FZ, I, C) { \
                const float tnx = fmaf(KKD_H(NX, I), tax, tbx), tfx = fmaf(KKD_H(FX, I), tax, tbx); \
                const float tny = fmaf(KKD_H(NY, I), tay, tby), tfy = fmaf(KKD_H(FY, I), tay, tby); \
                const float tnz = fmaf(KKD_H(NZ, I), taz, tbz), tfz = fmaf(KKD_H(FZ, I), taz, tbz); \
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin)); \
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax)); \
                D = (tn <= tf * 1.000002f && (C) != 0u) ? tn : kInf; }
                KKD_SLAB(d0, nxA, fxA, nyA, fyA, nzA, fzA, 0, ch.x)
                KKD_SLAB(d1, nxA, fxA, nyA, fyA, nzA, fzA, 1, ch.y)
                KKD_SLAB(d2, nxB, fxB, nyB, fyB, nzB, fzB, 0, ch.z)
                KKD_SLAB(d3, nxB, fxB, nyB, fyB, nzB, fzB, 1, ch.w)
    #undef KKD_SLAB
    #undef KKD_H
#else
                const uint32_t nx4 = negx ? qa.w : qa.x, fx4 = negx ? qa.x : qa.w;
                const uint32_t ny4 = negy ? qb.x : qa.y, fy4 = negy ? qa.y : qb.x;
                const uint32_t nz4 = negz ? qb.y : qa.z, fz4 = negz ? qa.z : qb.y;
    #define KKD_SLAB(D, K, C) { \
                const float tnx = fmaf((float)((nx4 >> (8 * K)) & 0xffu), tax, tbx), tfx = fmaf((float)((fx4 >> (8 * K)) & 0xffu), tax, tbx); \
                const float tny = fmaf((float)((ny4 >> (8 * K)) & 0xffu), tay, tby), tfy = fmaf((float)((fy4 >> (8 * K)) & 0xffu), tay, tby); \
                const float tnz = fmaf((float)((nz4 >> (8 * K)) & 0xffu), taz, tbz), tfz = fmaf((float)((fz4 >> (8 * K)) & 0xffu), taz, tbz); \
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin)); \
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax)); \
                D = (tn <= tf * 1.000002f && (C) != 0u) ? tn : kInf; }
                KKD_SLAB(d0, 0, ch.x)
                KKD_SLAB(d1, 1, ch.y)
                KKD_SLAB(d2, 2, ch.z)
                KKD_SLAB(d3, 3, ch.w)
    #undef KKD_SLAB
#endif
                uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
                if (ANY) {
                    /* occlusion does not depend on the order of the walk, and a ray that reaches its light visits every box it
                     * pierces anyway: no sorting network for shadow rays, enter the first child that is hit and push the others
                     * (C4: shadow rays 127 -> 118 ms per 128 spp; C2: unchanged) */
                    const bool v0 = d0 < kInf, v1 = d1 < kInf, v2 = d2 < kInf, v3 = d3 < kInf;
                    if (v0 || v1 || v2 || v3) {
                        if ((v0 || v1 || v2) && v3) KKD_PUSH(c3)
                        if ((v0 || v1) && v2) KKD_PUSH(c2)
                        if (v0 && v1) KKD_PUSH(c1)
                        cur = v0 ? c0 : (v1 ? c1 : (v2 ? c2 : c3));
                    } else KKD_POP()
                } else {
    #define KKD_CSWAP(DA, CA, DB, CB) { const bool s = DB < DA; const float td = s ? DB : DA; DB = s ? DA : DB; DA = td; \
                                        const uint32_t tc = s ? CB : CA; CB = s ? CA : CB; CA = tc; }
                /* (sorting only as far as the nearest child -- 3 or 4 of the 5 comparators -- was measured: C2 BSDF rays 71.3 -> 73.7 /
                 * 72.8 ms, C4 166 -> 163 / 164 ms: the full order is kept) */
                KKD_CSWAP(d0, c0, d1, c1) KKD_CSWAP(d2, c2, d3, c3) KKD_CSWAP(d0, c0, d2, c2) KKD_CSWAP(d1, c1, d3, c3) KKD_CSWAP(d1, c1, d2, c2)
    #undef KKD_CSWAP
                if (d0 < kInf) {
                    if (d3 < kInf) KKD_PUSH(c3)
                    if (d2 < kInf) KKD_PUSH(c2)
                    if (d1 < kInf) KKD_PUSH(c1)
                    cur = c0;
                } else KKD_POP()
                }
            }
            /* a lane at a leaf with nothing pending makes the leaf its pending one and takes the next entry of its stack: it
             * walks on while the leaf waits for the wave's next triangle step */
            if ((cur & kLeafBit) && cur != kTraversalDone && pend == 0u) { pend = cur; KKD_POP() }
            {
                const unsigned long long pendLanes = __ballot(pend != 0u);
                const unsigned long long blocked = __ballot((cur & kLeafBit) && cur != kTraversalDone);       /* a second leaf while one is pending */
                const bool canWalk = __ballot(!(cur & kLeafBit)) != 0ull;
                if (pendLanes != 0ull && (KKD_TRI_GATE == 0 || !canWalk || __popcll(pendLanes) >= KKD_TRI_GATE || __popcll(blocked) >= KKD_TRI_GATE_BLOCKED)) {
                if (pend != 0u) {
                    const uint32_t slot = pend & kLeafStartMask;
                    if (COUNT) ++nTris;
                    const float4 *tp = reinterpret_cast<const float4 *>(KKD_REC(reinterpret_cast<const uint4 *>(tris4), slot, 48));   /* 32-bit shift-add, not a quarter-rate 64-bit multiply */
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    float t, u, v;
                    /* of two triangles hit at the same distance the one with the lower primitive id is reported, whatever the order
                     * in which the walk reaches them: the packet walk's order depends on the other rays of the wave, a render cut
                     * into tiles or sample ranges groups the rays differently (coplanar emitters made 4 pixels of C3 differ), and
                     * slot numbers depend on which builder made the tree */
                    const uint32_t prim = PROJ ? __float_as_uint(c.z) : __float_as_uint(a.w);
#if defined(KKD_TRI_LDS) && KKD_TRI_LDS != 2     /* (2: the LDS is taken and written, the test reads registers -- what the occupancy alone costs) */
                    if ((PROJ ? projIntersectLds<BLOCK>(sray + tid, a, b, c, tmin, tmax, t, u, v)
#else
                    if ((PROJ ? projIntersect(ox, oy, oz, dx, dy, dz, a, b, c, tmin, tmax, t, u, v)
#endif
                              : woopIntersect(wr, ox, oy, oz, a, b, c, tmin, tmax, t, u, v))
                        && (ANY || t < tmax || hit.id == kNoHit || prim < hprim)) {
                        tmax = t; hit.t = t; hit.u = u; hit.v = v; hit.id = slot; hprim = prim;
                        if (ANY) stop();
                    }
                    if (pend & (7u << 28)) pend = pend - (1u << 28) + 1u;   /* next triangle of this leaf */
                    else pend = 0u;
                }
                }
            }
        }
#undef KKD_PUSH
#undef KKD_POP
#undef KKD_REC
    }
};

/*
 * ANY      : stop at the first hit (ShapeKDTree::rayIntersect(ray)); else closest hit.
 * COUNT    : accumulate visited nodes / tested triangles (the N_nodes, N_tris of the roofline formula).
 * PROJ     : tris is the ProjTri table and the leaf test is the reference's projection test; else LeafTri + watertight.
 * Returns hit.id = LeafTri slot or kNoHit.
 */
template <bool ANY, bool COUNT, bool PROJ, int STACK, int BLOCK>
__device__ __forceinline__ DHit traverse(const Node4 *__restrict__ nodes, const void *__restrict__ tris,
                                         const DRay &ray, uint32_t *sstack, uint32_t &nNodes, uint32_t &nTris) {
    Walk<ANY, COUNT, PROJ, STACK, BLOCK> w;
    uint32_t ovf[kStackOverflow];
    w.begin(ray);
    w.run(nodes, tris, sstack, ovf, nNodes, nTris, 1 << 24);
    return w.hit;
}

/*
 * Packet traversal for coherent rays (primary rays: the 64 lanes of a wave hold samples of one pixel or of neighbouring
 * pixels).  The wave walks the tree ONCE: the current node and the stack are wave-uniform, node (WideNode, float boxes)
 * and triangle records are fetched through the scalar cache, a child is visited if ANY lane's ray hits its box within
 * that lane's current [tmin, tmax], children are ordered by the entry distance of their first hitting lane.  Every lane
 * still tests its own ray against every visited triangle, so the result per ray is that of the per-lane walk (ties
 * aside).  No divergent control flow, no per-lane stack.
 * wstack: LDS, kMaxTraversalStack words owned by this wave.  Counters: a lane counts a child it hits (node or the
 * triangles of a leaf), i.e. what its own walk would have visited.
 */
/* OCT: the sign octant of the ray directions when it is the same for every live lane of the wave (bit 0 / 1 / 2: x / y / z
 * negative), 8 otherwise.  With a known octant the near and the far plane of every slab are known at compile time, so a
 * child costs 6 fma + 2 max/max3 + 2 min/min3 instead of 6 more min / max -- same values, same walk, a third fewer VALU
 * instructions in a kernel that is VALU-issue bound.  Primary rays of one pixel share their octant except across an axis. */
template <bool COUNT, bool PROJ, int OCT, int R>
__device__ __forceinline__ void packetWalk(const WideNode *__restrict__ wnodes, const float4 *__restrict__ tris4, const DRay *ray, const WoopRay *wr,
                                           const float *idx, const float *idy, const float *idz, const float *tmin, float *tmax,
                                           uint32_t *wstack, DHit *hit, uint32_t &nNodes, uint32_t &nTris) {
    int sp = 0;
    uint32_t cur = 0;
    uint32_t hprim[R];
#pragma unroll
    for (int r = 0; r < R; ++r) hprim[r] = 0u;
    /* every wave leaves this loop: each iteration pops or descends in a finite tree; the step bound ends a malformed one */
    for (int guard = 1 << 24; guard > 0; --guard) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        if (!(cur & kLeafBit)) {
            /* (reading the record as eight 16-byte words up front instead was measured slower: 30.1 vs 28.9 ms; the scalar side
             * of this loop, ~85 SALU instructions per node visit, bounds it together with the VALU side) */
            const WideNode *n = wnodes + cur;
            float tbx[R], tby[R], tbz[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { tbx[r] = (n->ox - ray[r].ox) * idx[r]; tby[r] = (n->oy - ray[r].oy) * idy[r]; tbz[r] = (n->oz - ray[r].oz) * idz[r]; }
            uint32_t dk[4], ck[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ck[k] = n->child[k]; dk[k] = 0xFFFFFFFFu;
                if (ck[k] == 0u) continue;         /* empty slot (wave-uniform) */
                bool h = false; float tnMin = kInf;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float tn, tf;
                    if (OCT < 8) {
                        const float nx = (OCT & 1) ? n->rmaxx[k] : n->rminx[k], fx = (OCT & 1) ? n->rminx[k] : n->rmaxx[k];
                        const float ny = (OCT & 2) ? n->rmaxy[k] : n->rminy[k], fy = (OCT & 2) ? n->rminy[k] : n->rmaxy[k];
                        const float nz = (OCT & 4) ? n->rmaxz[k] : n->rminz[k], fz = (OCT & 4) ? n->rminz[k] : n->rmaxz[k];
                        tn = fmaxf(fmaxf(fmaf(nx, idx[r], tbx[r]), fmaf(ny, idy[r], tby[r])), fmaxf(fmaf(nz, idz[r], tbz[r]), tmin[r]));
                        tf = fminf(fminf(fmaf(fx, idx[r], tbx[r]), fmaf(fy, idy[r], tby[r])), fminf(fmaf(fz, idz[r], tbz[r]), tmax[r]));
                    } else {
                        const float t0x = fmaf(n->rminx[k], idx[r], tbx[r]), t1x = fmaf(n->rmaxx[k], idx[r], tbx[r]);
                        const float t0y = fmaf(n->rminy[k], idy[r], tby[r]), t1y = fmaf(n->rmaxy[k], idy[r], tby[r]);
                        const float t0z = fmaf(n->rminz[k], idz[r], tbz[r]), t1z = fmaf(n->rmaxz[k], idz[r], tbz[r]);
                        tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin[r]));
                        tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax[r]));
                    }
                    const bool hr = tn <= tf * 1.000002f;
                    if (COUNT && hr) { if (ck[k] & kLeafBit) nTris += ((ck[k] >> 28) & 7u) + 1u; else ++nNodes; }
                    if (R == 1) { h = hr; tnMin = tn; }
                    else { h = h || hr; tnMin = hr ? fminf(tnMin, tn) : tnMin; }
                }
                const unsigned long long m = __ballot(h);
                if (m) dk[k] = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tnMin), __ffsll((long long)m) - 1);   /* tn >= 0: orders as uint */
            }
#define KKD_SSWAP(A, B) { if (dk[B] < dk[A]) { const uint32_t td = dk[A]; dk[A] = dk[B]; dk[B] = td; const uint32_t tc = ck[A]; ck[A] = ck[B]; ck[B] = tc; } }
            KKD_SSWAP(0, 1) KKD_SSWAP(2, 3) KKD_SSWAP(0, 2) KKD_SSWAP(1, 3) KKD_SSWAP(1, 2)
#undef KKD_SSWAP
            if (dk[3] != 0xFFFFFFFFu) wstack[sp++] = ck[3];
            if (dk[2] != 0xFFFFFFFFu) wstack[sp++] = ck[2];
            if (dk[1] != 0xFFFFFFFFu) wstack[sp++] = ck[1];
            if (dk[0] != 0xFFFFFFFFu) { cur = ck[0]; continue; }
        } else {
            const uint32_t start = cur & kLeafStartMask, cnt = ((cur >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float4 *tp = tris4 + 3u * (start + i);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const uint32_t prim = PROJ ? __float_as_uint(c.z) : __float_as_uint(a.w);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float t, u, v;
                    if ((PROJ ? projIntersect(ray[r].ox, ray[r].oy, ray[r].oz, ray[r].dx, ray[r].dy, ray[r].dz, a, b, c, tmin[r], tmax[r], t, u, v)
                              : woopIntersect(wr[r], ray[r].ox, ray[r].oy, ray[r].oz, a, b, c, tmin[r], tmax[r], t, u, v))
                        && (t < tmax[r] || hit[r].id == kNoHit || prim < hprim[r])) {      /* equal distances: the lower primitive id, as in Walk::run */
                        tmax[r] = t; hit[r].t = t; hit[r].u = u; hit[r].v = v; hit[r].id = start + i; hprim[r] = prim;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = wstack[--sp];
    }
}

/* R rays per lane (1 or 2): with two, a wave walks for 128 rays at once -- the scalar work of the walk (node fetches, ballots,
 * sorting network, wave stack) is paid per wave, so it halves per ray. */
template <bool COUNT, bool PROJ, int R>
__device__ __forceinline__ void traversePacket(const WideNode *__restrict__ wnodes, const void *__restrict__ tris, const DRay *ray,
                                               uint32_t *wstack, DHit *hit, uint32_t &nNodes, uint32_t &nTris) {
    float tmin[R], tmax[R], idx[R], idy[R], idz[R];
    WoopRay wr[R];
    bool anyLive = false;
    const float kBig = 0x1p100f, kTiny = 0x1p-100f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        hit[r].t = kInf; hit[r].u = 0.f; hit[r].v = 0.f; hit[r].id = kNoHit;
        tmin[r] = adaptiveMinT<false>(ray[r].mint, ray[r].ox, ray[r].oy, ray[r].oz);
        tmax[r] = ray[r].maxt;
        if (!(tmax[r] > tmin[r])) tmax[r] = -1.0f;              /* dead ray: no box and no triangle is ever hit */
        anyLive = anyLive || tmax[r] > tmin[r];
    }
    const unsigned long long liveLanes = __ballot(anyLive);
    if (liveLanes == 0ull) return;
    /* sign octant of the live rays (a dead ray hits nothing whatever planes it is given) */
    unsigned long long mx = 0, my = 0, mz = 0, px = 0, py = 0, pz = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        idx[r] = fabsf(ray[r].dx) < kTiny ? copysignf(kBig, ray[r].dx) : 1.0f / ray[r].dx;
        idy[r] = fabsf(ray[r].dy) < kTiny ? copysignf(kBig, ray[r].dy) : 1.0f / ray[r].dy;
        idz[r] = fabsf(ray[r].dz) < kTiny ? copysignf(kBig, ray[r].dz) : 1.0f / ray[r].dz;
        wr[r].init(ray[r].dx, ray[r].dy, ray[r].dz);
        const bool live = tmax[r] > tmin[r];
        mx |= __ballot(live && idx[r] < 0.f); px |= __ballot(live && !(idx[r] < 0.f));
        my |= __ballot(live && idy[r] < 0.f); py |= __ballot(live && !(idy[r] < 0.f));
        mz |= __ballot(live && idz[r] < 0.f); pz |= __ballot(live && !(idz[r] < 0.f));
    }
    const float4 *tris4 = reinterpret_cast<const float4 *>(tris);
    if (COUNT) nNodes += (uint32_t)R;              /* the root */
    const bool uniform = (mx == 0ull || px == 0ull) && (my == 0ull || py == 0ull) && (mz == 0ull || pz == 0ull);
    const int oct = uniform ? ((mx ? 1 : 0) | (my ? 2 : 0) | (mz ? 4 : 0)) : 8;
#define KKD_PW(O) packetWalk<COUNT, PROJ, O, R>(wnodes, tris4, ray, wr, idx, idy, idz, tmin, tmax, wstack, hit, nNodes, nTris)
    switch (oct) {
        case 0: KKD_PW(0); break; case 1: KKD_PW(1); break; case 2: KKD_PW(2); break; case 3: KKD_PW(3); break;
        case 4: KKD_PW(4); break; case 5: KKD_PW(5); break; case 6: KKD_PW(6); break; case 7: KKD_PW(7); break;
        default: KKD_PW(8); break;
    }
#undef KKD_PW
}

} /* namespace kkd */
