/*
 * kkd_shade.h -- device-side scene queries of the kk direct path: hit record, sensor rays, emitter
 * sampling / pdf / evaluation (area, point, environment map).
 *
 * Semantics follow the reference (paths relative to /root/reference/src):
 *   sensors/perspective.cpp:271-298            sampleRayDifferential (differentials are not used on this path)
 *   skdtree.h fillIntersectionRecord<true> (absent header; behaviour pinned by tests/test_dgeom.cpp:35-120,
 *     corroborated by librender/skdtree.cpp:164-186), librender/trimesh.cpp:687-747 (uv tangents)
 *   librender/scene.cpp:828-852, 949-952       sampleEmitterDirect / pdfEmitterDirect
 *   librender/shape.cpp:99-123, librender/trimesh.cpp:416-427, libcore/triangle.cpp:24-59
 *   emitters/area.cpp:104-109,158-183 ; emitters/point.cpp:131-151
 *   emitters/envmap.cpp:358-410, 516-633, 653-658
 */
#pragma once
#include "kkd_types.h"
#include <hip/hip_fp16.h>

namespace kkd {

struct Its {
    f3 p, geoN, wi; Frame sh; float t;
    int shape, material, emitter; uint32_t shapeFlags;
};

struct DirectRec { f3 ref, refN, p, n, d; float dist, pdf; int measure /* 0 solid angle, 1 discrete */; int emitter; };

/* element offsets are computed in 32 bits (3u * i is a shift-add; a 64-bit index multiply is a quarter-rate v_mad_u64_u32):
 * vertex, index and triangle tables stay below 2^32 elements, which kkd_scene_upload checks */
__device__ __forceinline__ f3 ldv(const float *base, uint32_t i) { const float *p = base + 3u * i; return f3(p[0], p[1], p[2]); }

/* perspective.cpp:271-298 */
__device__ __forceinline__ DRay sensorRay(const DSensor &s, float px, float py) {
    const float x = px * s.invResX, y = py * s.invResY;
    const float *m = s.s2c;
    float cx = m[0] * x + m[1] * y + m[3], cy = m[4] * x + m[5] * y + m[7], cz = m[8] * x + m[9] * y + m[11], cw = m[12] * x + m[13] * y + m[15];
    f3 nearP = (cw == 1.0f) ? f3(cx, cy, cz) : f3(cx, cy, cz) / cw;
    f3 d = normalize(nearP);
    const float invZ = frcp(d.z);
    const float *w = s.toWorld;
    DRay r;
    r.ox = w[3]; r.oy = w[7]; r.oz = w[11];
    r.dx = w[0] * d.x + w[1] * d.y + w[2] * d.z;
    r.dy = w[4] * d.x + w[5] * d.y + w[6] * d.z;
    r.dz = w[8] * d.x + w[9] * d.y + w[10] * d.z;
    r.mint = s.nearClip * invZ; r.maxt = s.farClip * invZ;
    return r;
}

/* Hit record from (slot, u, v).  full=false skips the shading frame tangents and wi (emitter lookups only
 * need p / shading normal). */
template <bool FULL>
__device__ __forceinline__ void fillIntersection(const DScene &sc, f3 rayD, const DHit &hit, Its &its) {
    const float4 *tp = reinterpret_cast<const float4 *>(sc.tris) + 3u * hit.id;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const uint32_t prim = __float_as_uint(a.w), sh = __float_as_uint(b.w);
    const DShape shape = sc.shapes[sh];
    const f3 p0(a.x, a.y, a.z), p1(b.x, b.y, b.z), p2(c.x, c.y, c.z);
    const float b0 = 1.f - hit.u - hit.v, b1 = hit.u, b2 = hit.v;
    its.t = hit.t; its.shape = (int)sh; its.material = shape.material; its.emitter = shape.emitter; its.shapeFlags = shape.flags;
    its.p = p0 * b0 + p1 * b1 + p2 * b2;
    const f3 side1 = p1 - p0, side2 = p2 - p0;
    f3 faceNormal = cross(side1, side2);
    const float len = length(faceNormal);
    if (len != 0.f) faceNormal = faceNormal / len;
    uint32_t i0 = 0, i1 = 0, i2 = 0;
    if (shape.flags & (KKD_SHAPE_HAS_NORMALS | KKD_SHAPE_HAS_TEXCOORDS)) {
        const uint32_t *ip = sc.indices + 3u * prim;
        i0 = ip[0]; i1 = ip[1]; i2 = ip[2];
    }
    f3 n;
    if (shape.flags & KKD_SHAPE_HAS_NORMALS) {
        n = normalize(ldv(sc.normals, i0) * b0 + ldv(sc.normals, i1) * b1 + ldv(sc.normals, i2) * b2);
        if (dot(faceNormal, n) < 0.f) faceNormal = -faceNormal;
    } else n = faceNormal;
    its.geoN = faceNormal;
    its.sh.n = n;
    if (FULL) {
        f3 dpdu = side1;
        if (shape.flags & KKD_SHAPE_HAS_TEXCOORDS) {   /* trimesh.cpp:687-747 */
            const float *t0 = sc.texcoords + 2u * i0, *t1 = sc.texcoords + 2u * i1, *t2 = sc.texcoords + 2u * i2;
            const float du1 = t1[0] - t0[0], dv1 = t1[1] - t0[1], du2 = t2[0] - t0[0], dv2 = t2[1] - t0[1];
            const float det = du1 * dv2 - dv1 * du2;
            if (len == 0.f) dpdu = f3(0.f);
            else if (det == 0.f) { f3 tv; coordinateSystem(cross(side1, side2) / len, dpdu, tv); }
            else dpdu = (side1 * dv2 - side2 * dv1) * frcp(det);
        }
        its.sh.s = normalize(dpdu - n * dot(n, dpdu));
        its.sh.t = cross(n, its.sh.s);
        its.wi = its.sh.toLocal(-rayD);
    }
}

/* pmf.h DiscreteDistribution::sample: lower_bound over n+1 entries, skip zero-probability entries */
template <uint32_t G = 0>
__device__ __forceinline__ uint32_t cdfSample(const float *cdf, uint32_t n, float v, const uint32_t *guide = nullptr) {
    uint32_t lo = 0, cnt = n + 1;
    if (G != 0u && guide) {   /* bracket of the sample's bucket (kkd_types.h): same lower_bound, ~4 instead of ~14 dependent loads for 10 k entries */
        const uint32_t b = min((uint32_t)(v * (float)G), G - 1u);
        lo = guide[b]; cnt = guide[b + 1u] - lo + 1u;
    }
    while (cnt > 0) {
        const uint32_t step = cnt >> 1, mid = lo + step;
        if (cdf[mid] < v) { lo = mid + 1; cnt -= step + 1; } else cnt = step;
    }
    uint32_t index = lo > 0 ? lo - 1 : 0;
    index = min(index, n - 1);
    while (index + 1 < n && cdf[index + 1] - cdf[index] == 0.f) ++index;
    return index;
}
template <uint32_t G = 0>
__device__ __forceinline__ uint32_t cdfSampleReuse(const float *cdf, uint32_t n, float &v, float &pdf, const uint32_t *guide = nullptr) {
    const uint32_t index = cdfSample<G>(cdf, n, v, guide);
    const float c0 = cdf[index], c1 = cdf[index + 1];
    pdf = c1 - c0;
    v = fdiv(v - c0, c1 - c0);
    return index;
}
/* envmap.cpp:653-658 */
__device__ __forceinline__ uint32_t envSampleReuse(const float *cdf, uint32_t size, float &sample, const uint32_t *guide, uint32_t G) {
    /* std::lower_bound over the size + 1 entries, started on the bracket the guide table gives for the sample's bucket */
    const uint32_t b = min((uint32_t)(sample * (float)G), G - 1u);
    uint32_t lo = guide[b], cnt = guide[b + 1u] - lo + 1u;
    while (cnt > 0) {
        const uint32_t step = cnt >> 1, mid = lo + step;
        if (cdf[mid] < sample) { lo = mid + 1; cnt -= step + 1; } else cnt = step;
    }
    const uint32_t index = min(lo > 0 ? lo - 1 : 0u, size - 1);
    const float c0 = cdf[index], c1 = cdf[index + 1];
    sample = fdiv(sample - c0, c1 - c0);
    return index;
}

/* the same through the bucket's guide record of W words (kkd_types.h), loaded whole: the CDF does not decrease, so the index
 * std::lower_bound returns is the bracket's start plus the number of entries below the sample, counted in registers */
template <uint32_t W>
__device__ __forceinline__ uint32_t envSampleReuseRec(const float *cdf, uint32_t size, float &sample, const uint32_t *guide, const uint32_t *recs, uint32_t G) {
    static_assert(W % 4 == 0 && W >= 4, "records are read as 16-byte words");
    const uint32_t b = min((uint32_t)(sample * (float)G), G - 1u);
    uint32_t w[W];
#pragma unroll
    for (uint32_t q = 0; q < W / 4; ++q) {
        const uint4 v = reinterpret_cast<const uint4 *>(recs + (size_t)b * W)[q];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    if ((w[0] >> 24) == 0xFFu) return envSampleReuse(cdf, size, sample, guide, G);
    const uint32_t first = w[0] & 0xFFFFFFu;
    uint32_t lo = 0;                                                /* w[2 + i] = cdf[first + i]; 2.0f past the bracket */
#pragma unroll
    for (uint32_t i = 2; i < W; ++i) lo += __uint_as_float(w[i]) < sample ? 1u : 0u;
    const uint32_t abs = first + lo;                                /* what std::lower_bound returns */
    const uint32_t index = min(abs > 0 ? abs - 1 : 0u, size - 1);
    const uint32_t sel = index + 2u - first;                        /* w[sel] = cdf[index]; index >= first - 1 */
    float c0 = __uint_as_float(w[1]), c1 = __uint_as_float(w[2]);
#pragma unroll
    for (uint32_t j = 2; j + 1 < W; ++j)
        if (sel == j) { c0 = __uint_as_float(w[j]); c1 = __uint_as_float(w[j + 1]); }
    sample = fdiv(sample - c0, c1 - c0);
    return index;
}

__device__ __forceinline__ f3 rot3(const float *R, f3 v) {
    return f3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
/* mipmap.h evalTexel: repeat in U, clamp in V (envmap.cpp:145-146); texels are stored as half */
__device__ __forceinline__ f3 envTexel(const DEnvMap &e, int x, int y) {
    x = x % e.w; if (x < 0) x += e.w;
    y = min(max(y, 0), e.h - 1);
    const uint2 raw = reinterpret_cast<const uint2 *>(e.texels)[(size_t)y * e.w + x];
    const __half2 rg = *reinterpret_cast<const __half2 *>(&raw.x), ba = *reinterpret_cast<const __half2 *>(&raw.y);
    const float2 frg = __half22float2(rg), fba = __half22float2(ba);
    return f3(frg.x, frg.y, fba.x);
}
__device__ __forceinline__ void envBilinearRows(const DEnvMap &e, float u, float v, f3 &value1, f3 &value2, int &yPos) {
    const int xPos = (int)floorf(u); yPos = (int)floorf(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    value1 = envTexel(e, xPos, yPos) * dx2 * dy2 + envTexel(e, xPos + 1, yPos) * dx1 * dy2;
    value2 = envTexel(e, xPos, yPos + 1) * dx2 * dy1 + envTexel(e, xPos + 1, yPos + 1) * dx1 * dy1;
}
/* envmap.cpp:380-393,408-409 without ray differentials: bilinear level-0 lookup */
__device__ __forceinline__ f3 envEval(const DEnvMap &e, f3 worldDir) {
    const f3 v = rot3(e.toLocal, worldDir);
    const float uu = atan2f(v.x, -v.z) * kInvTwoPi, vv = safeAcos(v.y) * kInvPi;
    if (!isfinite(uu) || !isfinite(vv)) return f3(0.f);
    f3 a, b; int yPos;
    envBilinearRows(e, uu * e.w - 0.5f, vv * e.h - 0.5f, a, b, yPos);
    return (a + b) * e.scale;
}
/* ---- filtered lookup for rays with differentials (camera rays): envmap.cpp:394-407 -> TMIPMap::eval, EWA filter, maxAnisotropy 10.
 * mipmap.h is not in the checkout: restated from the published Mitsuba 0.5 header (the oracle holds the same restatement),
 * parity unpinned.  Runs in a kernel of its own (kEnvMiss): inside kShade it cost that kernel 35 spilled registers.  Everything
 * is inlined: one out-of-line call gave the kernel an 800-byte private segment, and a dispatch whose scratch demand exceeds the
 * queue's standing allocation has it allocated and freed around every launch (2.3 ms per launch on C4). */
__device__ __forceinline__ f3 envTexelL(const DEnvMap &e, int level, int x, int y) {
    const int lw = e.levelW[level], lh = e.levelH[level];
    x = x % lw; if (x < 0) x += lw;
    y = min(max(y, 0), lh - 1);
    const uint2 raw = reinterpret_cast<const uint2 *>(e.texels)[(size_t)e.levelOfs[level] + (size_t)y * lw + x];
    const __half2 rg = *reinterpret_cast<const __half2 *>(&raw.x), ba = *reinterpret_cast<const __half2 *>(&raw.y);
    const float2 frg = __half22float2(rg), fba = __half22float2(ba);
    return f3(frg.x, frg.y, fba.x);
}
__device__ __forceinline__ f3 envBilinearL(const DEnvMap &e, int level, float uvx, float uvy) {
    if (!isfinite(uvx) || !isfinite(uvy)) return f3(0.f);
    if (level >= e.nLevels) {                          /* evalBox on the coarsest level */
        const int l = e.nLevels - 1;
        return envTexelL(e, l, (int)floorf(uvx * (float)e.levelW[l]), (int)floorf(uvy * (float)e.levelH[l]));
    }
    const float u = uvx * (float)e.levelW[level] - 0.5f, v = uvy * (float)e.levelH[level] - 0.5f;
    const int xPos = (int)floorf(u), yPos = (int)floorf(v);
    const float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
    return envTexelL(e, level, xPos, yPos) * dx2 * dy2 + envTexelL(e, level, xPos, yPos + 1) * dx2 * dy1
         + envTexelL(e, level, xPos + 1, yPos) * dx1 * dy2 + envTexelL(e, level, xPos + 1, yPos + 1) * dx1 * dy1;
}
__device__ __forceinline__ f3 envEwaL(const DEnvMap &e, int level, float uvx, float uvy, float A, float B, float C) {
#pragma clang fp contract(off)
    if (!isfinite(A + B + C + uvx + uvy)) return f3(0.f);
    if (level >= e.nLevels) return envBilinearL(e, level, uvx, uvy);
    const float lw = (float)e.levelW[level], lh = (float)e.levelH[level];
    const float u = uvx * lw - 0.5f, v = uvy * lh - 0.5f;
    const float ratio0 = fdiv(lw, (float)e.w), ratio1 = fdiv(lh, (float)e.h);
    A = fdiv(A, ratio0 * ratio0); B = fdiv(B, ratio0 * ratio1); C = fdiv(C, ratio1 * ratio1);
    const float invDet = fdiv(1.0f, -B * B + 4.0f * A * C), deltaU = 2.0f * sqrtf(C * invDet), deltaV = 2.0f * sqrtf(A * invDet);
    const int u0 = (int)ceilf(u - deltaU), u1 = (int)floorf(u + deltaU), v0 = (int)ceilf(v - deltaV), v1 = (int)floorf(v + deltaV);
    const float As = A * 64.f, Bs = B * 64.f, Cs = C * 64.f;
    f3 result(0.f); float denominator = 0.0f;
    const float ddq = 2 * As, uu0 = (float)u0 - u;
    /* the footprint is bounded: minor radius in [1, 2) texels of this level (or the next finer one), major at most 10 x that */
    for (int vt = v0; vt <= v1; ++vt) {
        const float vv = (float)vt - v;
        float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
        float dq = As * (2 * uu0 + 1) + Bs * vv;
        for (int ut = u0; ut <= u1; ++ut) {
            if (q < 64.f) {
                const uint32_t qi = (uint32_t)q;
                if (qi < 64u) {
                    const float weight = e.ewaLut[(int)q];
                    result = result + envTexelL(e, level, ut, vt) * weight;
                    denominator += weight;
                }
            }
            q += dq; dq += ddq;
        }
    }
    if (denominator == 0.f) return envBilinearL(e, level, uvx, uvy);
    return result / denominator;
}
__device__ __forceinline__ f3 envEvalFiltered(const DEnvMap &e, f3 worldDir, f3 rxDir, f3 ryDir) {
#pragma clang fp contract(off)
    const f3 v = rot3(e.toLocal, worldDir);
    const float uvx = atan2f(v.x, -v.z) * kInvTwoPi, uvy = safeAcos(v.y) * kInvPi;
    const f3 dvdx = rot3(e.toLocal, rxDir) - v, dvdy = rot3(e.toLocal, ryDir) - v;
    const float t1 = fdiv(kInvTwoPi, v.x * v.x + v.z * v.z), t2 = fdiv(-kInvPi, fmaxf(safeSqrt(1.0f - v.y * v.y), kEpsilon));
    const float d0x = t1 * (dvdx.z * v.x - dvdx.x * v.z), d0y = t2 * dvdx.y, d1x = t1 * (dvdy.z * v.x - dvdy.x * v.z), d1y = t2 * dvdy.y;
    const float maxAnisotropy = 10.0f;
    const float du0 = d0x * (float)e.w, dv0 = d0y * (float)e.h, du1 = d1x * (float)e.w, dv1 = d1y * (float)e.h;
    float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
    const float root = hypotf(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root);
    float majorRadius = Aprime != 0.f ? sqrtf(fdiv(F, Aprime)) : 0.f, minorRadius = Cprime != 0.f ? sqrtf(fdiv(F, Cprime)) : 0.f;
    f3 value;
    if (!(minorRadius > 0.f) || !(majorRadius > 0.f) || F < 0.f) {
        const float level = log2f(fmaxf(majorRadius, kEpsilon));
        const int ilevel = (int)floorf(level);
        if (ilevel < 0) value = envBilinearL(e, 0, uvx, uvy);
        else { const float a = level - ilevel; value = envBilinearL(e, ilevel, uvx, uvy) * (1.0f - a) + envBilinearL(e, ilevel + 1, uvx, uvy) * a; }
    } else {
        if (minorRadius * maxAnisotropy < majorRadius) {
            minorRadius = fdiv(majorRadius, maxAnisotropy);
            const float theta = 0.5f * atanf(fdiv(B, A - C));
            float sinTheta, cosTheta; sincosf(theta, &sinTheta, &cosTheta);
            const float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta,
                        sin2Theta = 2 * sinTheta * cosTheta;
            A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
        }
        const float sc = fdiv(1.0f, F);
        A *= sc; B *= sc; C *= sc;
        const float level = fmaxf(0.0f, log2f(minorRadius));
        const int ilevel = (int)level;
        const float a = level - ilevel;
        if (majorRadius < 1.f || !(A > 0.f && C > 0.f)) value = envBilinearL(e, ilevel, uvx, uvy);
        else value = envEwaL(e, ilevel, uvx, uvy, A, B, C) * (1.0f - a) + envEwaL(e, ilevel + 1, uvx, uvy, A, B, C) * a;
    }
    return value * e.scale;
}
/* perspective.cpp:294-295 + RayDifferential::scaleDifferential (integrator.cpp:181): directions of the rays through the
 * neighbouring pixels, pulled towards the ray by 1 / sqrt(spp) */
__device__ __forceinline__ void sensorDifferentials(const DSensor &s, float px, float py, float scale, f3 d, f3 &rxD, f3 &ryD) {
    const float x = px * s.invResX, y = py * s.invResY;
    const float *m = s.s2c;
    float cx = m[0] * x + m[1] * y + m[3], cy = m[4] * x + m[5] * y + m[7], cz = m[8] * x + m[9] * y + m[11], cw = m[12] * x + m[13] * y + m[15];
    const f3 nearP = (cw == 1.0f) ? f3(cx, cy, cz) : f3(cx, cy, cz) / cw;
    const f3 dx = normalize(nearP + f3(s.dxCam[0], s.dxCam[1], s.dxCam[2])), dy = normalize(nearP + f3(s.dyCam[0], s.dyCam[1], s.dyCam[2]));
    const float *w = s.toWorld;
    rxD = f3(w[0] * dx.x + w[1] * dx.y + w[2] * dx.z, w[4] * dx.x + w[5] * dx.y + w[6] * dx.z, w[8] * dx.x + w[9] * dx.y + w[10] * dx.z);
    ryD = f3(w[0] * dy.x + w[1] * dy.y + w[2] * dy.z, w[4] * dy.x + w[5] * dy.y + w[6] * dy.z, w[8] * dy.x + w[9] * dy.y + w[10] * dy.z);
    rxD = d + (rxD - d) * scale; ryD = d + (ryD - d) * scale;
}
/* envmap.cpp:603-633 */
__device__ __forceinline__ float envPdfDirection(const DEnvMap &e, f3 d) {
    const float uu = atan2f(d.x, -d.z) * kInvTwoPi, vv = safeAcos(d.y) * kInvPi;
    if (!isfinite(uu) || !isfinite(vv)) return 0.f;
    f3 a, b; int yPos;
    envBilinearRows(e, uu * e.w - 0.5f, vv * e.h - 0.5f, a, b, yPos);
    const float sinTheta = safeSqrt(1.f - d.y * d.y);
    return (luminance(a) * e.rowWeights[min(max(yPos, 0), e.h - 1)] + luminance(b) * e.rowWeights[min(max(yPos + 1, 0), e.h - 1)])
           * e.normalization * frcp(fmaxf(fabsf(sinTheta), kEpsilon));
}
/* envEval and envPdfDirection of ONE direction (a BSDF ray that left the scene needs both for its MIS weight): the rotation,
 * the two inverse trigonometric functions, the four texels and the bilinear rows are computed once; every expression is the
 * one of the two functions above, so the results are theirs bit for bit */
__device__ __forceinline__ void envEvalWithPdf(const DEnvMap &e, f3 worldDir, f3 &value, float &pdf) {
    const f3 v = rot3(e.toLocal, worldDir);
    const float uu = atan2f(v.x, -v.z) * kInvTwoPi, vv = safeAcos(v.y) * kInvPi;
    value = f3(0.f); pdf = 0.f;
    if (!isfinite(uu) || !isfinite(vv)) return;
    f3 a, b; int yPos;
    envBilinearRows(e, uu * e.w - 0.5f, vv * e.h - 0.5f, a, b, yPos);
    value = (a + b) * e.scale;
    const float sinTheta = safeSqrt(1.f - v.y * v.y);
    pdf = (luminance(a) * e.rowWeights[min(max(yPos, 0), e.h - 1)] + luminance(b) * e.rowWeights[min(max(yPos + 1, 0), e.h - 1)])
          * e.normalization * frcp(fmaxf(fabsf(sinTheta), kEpsilon));
}
/* envmap.cpp:567-600 */
__device__ __forceinline__ void envSampleDirection(const DEnvMap &e, f2 sample, f3 &d, f3 &value, float &pdf) {
#if KKD_ENV_REC_ROWS == 0
    const uint32_t row = envSampleReuse(e.cdfRows, (uint32_t)e.h, sample.y, e.guideRows, kEnvGuideRows);
#else
    const uint32_t row = envSampleReuseRec<kEnvRecRowWords>(e.cdfRows, (uint32_t)e.h, sample.y, e.guideRows, e.recRows, kEnvGuideRows);
#endif
#if KKD_ENV_REC_COLS == 0
    const uint32_t col = envSampleReuse(e.cdfCols + (size_t)row * (e.w + 1), (uint32_t)e.w, sample.x, e.guideCols + row * (kEnvGuideCols + 1u), kEnvGuideCols);
#else
    const uint32_t col = envSampleReuseRec<kEnvRecColWords>(e.cdfCols + (size_t)row * (e.w + 1), (uint32_t)e.w, sample.x, e.guideCols + row * (kEnvGuideCols + 1u),
                                                            e.recCols + (size_t)row * kEnvGuideCols * kEnvRecColWords, kEnvGuideCols);
#endif
    const float px = (float)col + intervalToTent(sample.x), py = (float)row + intervalToTent(sample.y);
    f3 a, b; int yPos;
    envBilinearRows(e, px, py, a, b, yPos);
    value = (a + b) * e.scale;
    pdf = (luminance(a) * e.rowWeights[min(max(yPos, 0), e.h - 1)] + luminance(b) * e.rowWeights[min(max(yPos + 1, 0), e.h - 1)]) * e.normalization;
    float sinPhi, cosPhi, sinTheta, cosTheta;
    sincosf(e.pixelSizeX * (px + 0.5f), &sinPhi, &cosPhi);
    sincosf(e.pixelSizeY * (py + 0.5f), &sinTheta, &cosTheta);
    d = f3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
    pdf *= frcp(fmaxf(fabsf(sinTheta), kEpsilon));
}

/* area.cpp:104-109 */
__device__ __forceinline__ f3 areaEval(const DEmitter &e, f3 shN, f3 d) {
    if (dot(shN, d) <= 0.f) return f3(0.f);
    return f3(e.radiance[0], e.radiance[1], e.radiance[2]);
}

/* Scene::sampleEmitterDirect WITHOUT the visibility test (the shadow ray is queued by the caller):
 * returns value = Le / pdf with pdf including the emitter-selection probability; zero if not sampleable. */
__device__ __forceinline__ f3 sampleEmitterDirectUnoccluded(const DScene &sc, DirectRec &dRec, f2 sample) {
    float emPdf;
    const uint32_t index = cdfSampleReuse<kEmitterGuide>(sc.emitterCdf, sc.nEmitters, sample.x, emPdf, sc.emitterGuide);
    const DEmitter e = sc.emitters[index];
    f3 value(0.f);
    dRec.pdf = 0.f;
    if (e.type == KKD_EMITTER_AREA) {
        float dummy;
        const uint32_t local = cdfSampleReuse<kAreaGuide>(sc.areaCdf + e.cdfOffset, e.nPrims, sample.y, dummy,
                                                             e.guideOffset ? sc.areaGuide + (e.guideOffset - 1u) : nullptr);
        const uint32_t prim = e.firstPrim + local;
        const uint32_t *ip = sc.indices + 3u * prim;
        const uint32_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
        const f3 p0 = ldv(sc.positions, i0), p1 = ldv(sc.positions, i1), p2 = ldv(sc.positions, i2);
        const f2 bary = squareToUniformTriangle(sample);
        const f3 sideA = p1 - p0, sideB = p2 - p0;
        dRec.p = p0 + (sideA * bary.x) + (sideB * bary.y);
        if (e.shapeFlags & KKD_SHAPE_HAS_NORMALS)
            dRec.n = normalize(ldv(sc.normals, i0) * (1.0f - bary.x - bary.y) + ldv(sc.normals, i1) * bary.x + ldv(sc.normals, i2) * bary.y);
        else
            dRec.n = normalize(cross(sideA, sideB));
        dRec.pdf = e.invArea;
        dRec.d = dRec.p - dRec.ref;
        const float distSquared = dot(dRec.d, dRec.d);
        dRec.dist = fsqrt(distSquared);
        dRec.d = dRec.d / dRec.dist;
        const float dp = fabsf(dot(dRec.d, dRec.n));
        dRec.pdf *= dp != 0.f ? fdiv(distSquared, dp) : 0.0f;
        dRec.measure = 0;
        if (dot(dRec.d, dRec.refN) >= 0.f && dot(dRec.d, dRec.n) < 0.f && dRec.pdf != 0.f)
            value = f3(e.radiance[0], e.radiance[1], e.radiance[2]) / dRec.pdf;
        else dRec.pdf = 0.0f;
    } else if (e.type == KKD_EMITTER_POINT) {
        dRec.p = f3(e.position[0], e.position[1], e.position[2]); dRec.pdf = 1.0f; dRec.measure = 1;
        dRec.d = dRec.p - dRec.ref;
        dRec.dist = length(dRec.d);
        const float invDist = frcp(dRec.dist);
        dRec.d = dRec.d * invDist;
        dRec.n = f3(0.f);
        value = f3(e.radiance[0], e.radiance[1], e.radiance[2]) * (invDist * invDist);
    } else {
        f3 d, v; float pdf;
        envSampleDirection(sc.env, sample, d, v, pdf);
        const f3 wd = rot3(sc.env.toWorld, d);
        const f3 center(sc.env.center[0], sc.env.center[1], sc.env.center[2]);
        float nearT, farT;
        if (isZero(v) || pdf == 0.f || !sphereRayIntersect(center, sc.env.radius, dRec.ref, wd, nearT, farT) || nearT >= 0.f || farT <= 0.f) {
            dRec.pdf = 0.0f;
        } else {
            dRec.pdf = pdf; dRec.p = dRec.ref + wd * farT; dRec.n = normalize(center - dRec.p);
            dRec.dist = farT; dRec.d = wd; dRec.measure = 0;
            value = v / pdf;
        }
    }
    if (dRec.pdf != 0.f) {
        dRec.emitter = (int)index;
        dRec.pdf *= emPdf;
        return value / emPdf;
    }
    return f3(0.f);
}

/* scene.cpp:949-952 */
__device__ __forceinline__ float pdfEmitterDirect(const DScene &sc, const DirectRec &dRec) {
    const DEmitter e = sc.emitters[dRec.emitter];
    float pdf = 0.f;
    if (e.type == KKD_EMITTER_AREA) {
        if (dot(dRec.d, dRec.refN) >= 0.f && dot(dRec.d, dRec.n) < 0.f)
            pdf = fdiv(e.invArea * (dRec.dist * dRec.dist), fabsf(dot(dRec.d, dRec.n)));
    } else if (e.type == KKD_EMITTER_POINT) {
        pdf = dRec.measure == 1 ? 1.0f : 0.0f;
    } else {
        pdf = envPdfDirection(sc.env, rot3(sc.env.toLocal, dRec.d));
    }
    return pdf * (e.weight * sc.emitterNormalization);
}

} /* namespace kkd */
