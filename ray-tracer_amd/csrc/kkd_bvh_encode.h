/*
 * kkd_bvh_encode.h -- encoding of one 4-wide node into the two records the kernels read (Node4: 8-bit grid boxes, WideNode:
 * float boxes relative to the node's min corner; layouts in kkd_types.h).  Shared by the host builder (kkd_bvh.cpp) and the
 * device builder (kkd_gpu_build.hip), so both produce byte-identical records from the same boxes.
 */
#pragma once
#include "kkd_types.h"
#include <cmath>
#include <cstring>
#if defined(__HIPCC__)
#define KKD_ENC_HD __host__ __device__ inline
#else
#define KKD_ENC_HD inline
#endif

namespace kkd {

/* builder-internal uncompressed 4-wide node */
struct Node4F { float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4]; uint32_t child[4]; int nk; };

/* 8-bit grid encoding of a node's child boxes (layout: kkd_types.h).  Per axis: origin = min over the children,
 * scale = smallest power of two with extent <= 255 * scale; min planes round down, max planes up, and every plane is
 * checked through the kernel's own decode, fmaf(q, scale, origin), so the decoded box always encloses the true one. */
KKD_ENC_HD float decodePlane(uint8_t q, float scale, float origin) { return fmaf((float)q, scale, origin); }

/* float boxes relative to the node's min corner, rounded outwards (so that origin + rel encloses the true plane) */
KKD_ENC_HD void widenNode(const Node4F &n, WideNode &out) {
    memset(&out, 0, sizeof(out));
    const float *mins[3] = { n.minx, n.miny, n.minz }, *maxs[3] = { n.maxx, n.maxy, n.maxz };
    float *rmins[3] = { out.rminx, out.rminy, out.rminz }, *rmaxs[3] = { out.rmaxx, out.rmaxy, out.rmaxz };
    float origin[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY;
        for (int k = 0; k < n.nk; ++k) lo = fminf(lo, mins[a][k]);
        origin[a] = lo;
        for (int k = 0; k < 4; ++k) {
            if (k >= n.nk) { rmins[a][k] = INFINITY; rmaxs[a][k] = INFINITY; continue; }
            float r0 = mins[a][k] - lo, r1 = maxs[a][k] - lo;
            while (lo + r0 > mins[a][k]) r0 = nextafterf(r0, -INFINITY);
            while (lo + r1 < maxs[a][k]) r1 = nextafterf(r1, INFINITY);
            rmins[a][k] = r0; rmaxs[a][k] = r1;
        }
    }
    out.ox = origin[0]; out.oy = origin[1]; out.oz = origin[2];
    for (int k = 0; k < 4; ++k) out.child[k] = k < n.nk ? n.child[k] : 0u;
}

#ifdef KKD_NODE_FP16
/* half <-> float for the non-negative plane coordinates of the fp16 node layout, with directed rounding (bit arithmetic, the
 * same on host and device).  Values below the smallest normal half (2^-14) round to 0 / up to 2^-14. */
KKD_ENC_HD float halfBitsToFloat(uint16_t h) {
    if ((h & 0x7fffu) == 0u) return 0.f;
    const uint32_t u = ((((uint32_t)h >> 10) & 0x1fu) - 15u + 127u) << 23 | ((uint32_t)h & 0x3ffu) << 13;
    float f; memcpy(&f, &u, 4); return f;
}
KKD_ENC_HD uint16_t halfFloorBits(float x) {
    if (!(x >= 6.103515625e-05f)) return 0;
    if (x >= 65504.f) return 0x7bff;
    uint32_t u; memcpy(&u, &x, 4);
    return (uint16_t)((((u >> 23) - 127u + 15u) << 10) | ((u >> 13) & 0x3ffu));
}
KKD_ENC_HD uint16_t halfCeilBits(float x) {
    if (!(x > 0.f)) return 0;
    if (x <= 6.103515625e-05f) return 0x0400;
    uint32_t u; memcpy(&u, &x, 4);
    uint16_t h = halfFloorBits(x);
    if ((u & 0x1fffu) != 0u && h < 0x7bff) ++h;
    return h;
}
KKD_ENC_HD float decodePlaneH(uint16_t h, float scale, float origin) { return fmaf(halfBitsToFloat(h), scale, origin); }

/* fp16 encoding of a node's child boxes: per axis origin = min over the children, scale = smallest power of two with extent <=
 * 2047 * scale; min planes round down, max planes up, each checked through the kernel's decode. */
KKD_ENC_HD void quantizeNode(const Node4F &n, Node4 &out) {
    memset(&out, 0, sizeof(out));
    const float *mins[3] = { n.minx, n.miny, n.minz }, *maxs[3] = { n.maxx, n.maxy, n.maxz };
    uint16_t *hmins[3] = { out.hminx, out.hminy, out.hminz }, *hmaxs[3] = { out.hmaxx, out.hmaxy, out.hmaxz };
    float origin[3]; uint8_t expo[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < n.nk; ++k) { lo = fminf(lo, mins[a][k]); hi = fmaxf(hi, maxs[a][k]); }
        origin[a] = lo;
        const float extent = hi - lo;
        int e = 0;
        if (extent > 0.f) { frexpf(extent / 2047.0f, &e); }
        else e = -126;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        float scale = ldexpf(1.0f, e);
        while (e < 127 && decodePlaneH(0x6bff /* 4094 */, scale, lo) < hi) { ++e; scale = ldexpf(1.0f, e); }
        expo[a] = (uint8_t)(e + 127);
        for (int k = 0; k < 4; ++k) {
            if (k >= n.nk) { hmins[a][k] = 0x7bff; hmaxs[a][k] = 0; continue; }
            uint16_t h0 = halfFloorBits((mins[a][k] - lo) / scale), h1 = halfCeilBits((maxs[a][k] - lo) / scale);
            while (h0 > 0 && decodePlaneH(h0, scale, lo) > mins[a][k]) h0 = h0 == 0x0400 ? 0 : (uint16_t)(h0 - 1);
            while (h1 < 0x7bff && decodePlaneH(h1, scale, lo) < maxs[a][k]) h1 = h1 == 0 ? 0x0400 : (uint16_t)(h1 + 1);
            hmins[a][k] = h0; hmaxs[a][k] = h1;
        }
    }
    out.ox = origin[0]; out.oy = origin[1]; out.oz = origin[2];
    out.ex = expo[0]; out.ey = expo[1]; out.ez = expo[2]; out.epad = 0;
    for (int k = 0; k < 4; ++k) out.child[k] = k < n.nk ? n.child[k] : 0u;
}
#else
KKD_ENC_HD void quantizeNode(const Node4F &n, Node4 &out) {
    memset(&out, 0, sizeof(out));
    const float *mins[3] = { n.minx, n.miny, n.minz }, *maxs[3] = { n.maxx, n.maxy, n.maxz };
    uint8_t *qmins[3] = { out.qminx, out.qminy, out.qminz }, *qmaxs[3] = { out.qmaxx, out.qmaxy, out.qmaxz };
    float origin[3]; uint8_t expo[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < n.nk; ++k) { lo = fminf(lo, mins[a][k]); hi = fmaxf(hi, maxs[a][k]); }
        origin[a] = lo;
        const float extent = hi - lo;
        int e = 0;                                   /* scale = 2^e */
        if (extent > 0.f) { frexpf(extent / 255.0f, &e); }   /* extent/255 = m * 2^e, m in [0.5,1) => 2^e >= extent/255 */
        else e = -126;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        float scale = ldexpf(1.0f, e);
        /* make sure the top of the range is reachable after float rounding of the decode */
        while (e < 127 && decodePlane(255, scale, lo) < hi) { ++e; scale = ldexpf(1.0f, e); }
        expo[a] = (uint8_t)(e + 127);
        for (int k = 0; k < 4; ++k) {
            if (k >= n.nk) { qmins[a][k] = 255; qmaxs[a][k] = 0; continue; }
            int q0 = (int)floorf((mins[a][k] - lo) / scale), q1 = (int)ceilf((maxs[a][k] - lo) / scale);
            q0 = q0 < 0 ? 0 : (q0 > 255 ? 255 : q0); q1 = q1 < 0 ? 0 : (q1 > 255 ? 255 : q1);
            while (q0 > 0 && decodePlane((uint8_t)q0, scale, lo) > mins[a][k]) --q0;
            while (q1 < 255 && decodePlane((uint8_t)q1, scale, lo) < maxs[a][k]) ++q1;
            qmins[a][k] = (uint8_t)q0; qmaxs[a][k] = (uint8_t)q1;
        }
    }
    out.ox = origin[0]; out.oy = origin[1]; out.oz = origin[2];
    out.ex = expo[0]; out.ey = expo[1]; out.ez = expo[2]; out.epad = 0;
    for (int k = 0; k < 4; ++k) out.child[k] = k < n.nk ? n.child[k] : 0u;
}
#endif


} /* namespace kkd */
