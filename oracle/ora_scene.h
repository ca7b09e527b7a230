/*
 * ORACLE (test infrastructure only): CPU restatement of scene queries, emitters, sensor, film and
 * the kk direct integrator itself.
 *
 * Follows (relative to /root/reference/src):
 *   integrators/kk/direct.cpp:50-178           Li / integrateBrdf / integrateLight / computeMIS
 *   librender/integrator.cpp:140-188           SamplingIntegrator::renderBlock
 *   librender/imageproc.cpp:28-78, renderproc.cpp:158-176   tile decomposition (crop-relative offsets)
 *   librender/scene.cpp:376-380,828-852,949-952   emitter PDF, sampleEmitterDirect, pdfEmitterDirect
 *   librender/shape.cpp:99-123                 Shape::sampleDirect / pdfDirect
 *   librender/trimesh.cpp:392-427              area CDF, samplePosition
 *   libcore/triangle.cpp:24-67                 Triangle::sample / surfaceArea
 *   emitters/area.cpp:104-109,158-183          AreaLight
 *   emitters/point.cpp:131-151                 PointEmitter
 *   emitters/envmap.cpp:260-329,358-410,516-633,653-658   EnvironmentMap
 *   sensors/perspective.cpp:126-180,271-298 ; librender/sensor.cpp:84-108 ; libcore/transform.cpp:99-123
 *   libcore/rfilter.cpp:37-55 ; rfilters/{gaussian,box,tent,mitchell,catmullrom,lanczos}.cpp
 * Absent-header bodies (skdtree.h fillIntersectionRecord, imageblock.h put, pmf.h, records.inl,
 * mipmap.h evalBilinear/evalTexel) are restated from SURVEY.md 9.3-9.5: parity unpinned except for
 * the hit-record vectors of tests/test_dgeom.cpp:35-120.
 *
 * Primary-ray environment lookups are the EWA-filtered MIP lookup with ray differentials (envmap.cpp:394-407); TMIPMap and
 * Resampler live in headers the checkout does not hold (mipmap.h, rfilter.h): restated from the published Mitsuba 0.5 sources,
 * parity unpinned.
 */
#pragma once
#include "ora_accel.h"
#include "ora_bsdf.h"
#include "../include/kkdirect.h"
#include <thread>
#include <atomic>

namespace ora {

/* ---- half precision round trip (envmap.cpp:102-103 stores texels as half) ------------------- */
inline uint16_t floatToHalfBits(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t mant = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7C00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        mant |= 0x800000u;
        int shift = 14 - exp;
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}
inline float halfBitsToFloat(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F, mant = h & 0x3FFu, x;
    if (exp == 0) {
        if (mant == 0) x = sign;
        else {
            int e = -1;
            do { ++e; mant <<= 1; } while (!(mant & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7F800000u | (mant << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f; std::memcpy(&f, &x, 4); return f;
}
inline float roundToHalf(float f) { return halfBitsToFloat(floatToHalfBits(f)); }

/* ---- pmf.h (absent) --------------------------------------------------------------- */
struct DiscreteDistribution {
    std::vector<Float> cdf; Float sum = 0, normalization = 0;
    DiscreteDistribution() { cdf.push_back(0.0f); }
    void append(Float v) { cdf.push_back(cdf.back() + v); }
    size_t size() const { return cdf.size() - 1; }
    Float operator[](size_t i) const { return cdf[i + 1] - cdf[i]; }
    Float normalize() {
        sum = cdf.back();
        if (sum > 0) {
            normalization = 1.0f / sum;
            for (size_t i = 1; i < cdf.size(); ++i) cdf[i] *= normalization;
            cdf.back() = 1.0f;
        } else normalization = 0.0f;
        return sum;
    }
    size_t sample(Float v) const {
        auto entry = std::lower_bound(cdf.begin(), cdf.end(), v);
        size_t index = (size_t)std::max((ptrdiff_t)0, entry - cdf.begin() - 1);
        index = std::min(index, cdf.size() - 2);
        while ((*this)[index] == 0 && index < cdf.size() - 1) ++index;
        return index;
    }
    size_t sampleReuse(Float &v, Float &pdf) const {
        size_t index = sample(v);
        pdf = (*this)[index];
        v = (v - cdf[index]) / (cdf[index + 1] - cdf[index]);
        return index;
    }
    size_t sampleReuse(Float &v) const { Float p; return sampleReuse(v, p); }
};

/* ---- 4x4 helpers (libcore/transform.cpp) ------------------------------------------------------- */
struct M44 {
    Float m[4][4];
    static M44 identity() { M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = i == j; return r; }
    M44 operator*(const M44 &b) const {
        M44 r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            Float s = 0; for (int k = 0; k < 4; ++k) s += m[i][k] * b.m[k][j]; r.m[i][j] = s;
        }
        return r;
    }
    static M44 scale(Float x, Float y, Float z) { M44 r = identity(); r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; return r; }
    static M44 translate(Float x, Float y, Float z) { M44 r = identity(); r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z; return r; }
    /* transform.cpp:99-123 */
    static M44 perspective(Float fov, Float n, Float f) {
        Float recip = 1.0f / (f - n);
        Float cot = 1.0f / std::tan((fov / 2.0f) * (kPi / 180.0f));
        M44 r; std::memset(&r, 0, sizeof(r));
        r.m[0][0] = cot; r.m[1][1] = cot; r.m[2][2] = f * recip; r.m[2][3] = -n * f * recip; r.m[3][2] = 1;
        return r;
    }
    M44 inverse() const {
        double a[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[i][j]; a[i][j + 4] = i == j; }
        for (int c = 0; c < 4; ++c) {
            int piv = c;
            for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
            for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
            double d = a[c][c];
            for (int j = 0; j < 8; ++j) a[c][j] /= d;
            for (int r = 0; r < 4; ++r) if (r != c) { double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
        }
        M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (Float)a[i][j + 4];
        return r;
    }
    V3 point(V3 p) const {   /* Transform::operator()(Point) with homogeneous divide */
        Float x = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
        Float y = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
        Float z = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
        Float w = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
        if (w == 1.0f) return V3(x, y, z);
        return V3(x, y, z) / w;
    }
    V3 pointAffine(V3 p) const {
        return V3(m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3],
                  m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
                  m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]);
    }
    V3 vector(V3 v) const {
        return V3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
                  m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                  m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
    }
};

/* ---- reconstruction filter table (libcore/rfilter.cpp:37-55) ----------------------------------- */
struct RFilter {
    static const int RES = 31;   /* MTS_FILTER_RESOLUTION */
    Float values[RES + 1], radius, scaleFactor; int borderSize;
    static Float evalKind(int type, Float radius, const Float *param, Float x) {
        switch (type) {
            case KKD_RFILTER_GAUSSIAN: { /* rfilters/gaussian.cpp:52-57 */
                Float stddev = param[0];
                Float alpha = -1.0f / (2.0f * stddev * stddev);
                return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
            }
            case KKD_RFILTER_BOX: /* rfilters/box.cpp:44-46 */
                return std::fabs(x) <= radius ? 1.0f : 0.0f;
            case KKD_RFILTER_TENT: /* rfilters/tent.cpp */
                return std::max(0.0f, 1.0f - std::fabs(x / radius));
            case KKD_RFILTER_MITCHELL: { /* rfilters/mitchell.cpp */
                Float B = param[0], C = param[1];
                x = std::fabs(x);
                Float x2 = x * x, x3 = x2 * x;
                if (x < 1) return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
                else if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
                return 0.0f;
            }
            case KKD_RFILTER_CATMULLROM: { /* rfilters/catmullrom.cpp */
                x = std::fabs(x);
                Float x2 = x * x, x3 = x2 * x, B = 0.0f, C = 0.5f;
                if (x < 1) return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
                else if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
                return 0.0f;
            }
            case KKD_RFILTER_LANCZOS: { /* rfilters/lanczos.cpp */
                x = std::fabs(x);
                if (x < kEpsilon) return 1.0f;
                else if (x > radius) return 0.0f;
                Float x1 = kPi * x, x2 = x1 / radius;
                return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
            }
        }
        return 0.0f;
    }
    void configure(int type, Float r, const Float *param) {
        radius = r;
        Float sum = 0.0f;
        for (int i = 0; i < RES; ++i) { Float v = evalKind(type, r, param, (r * i) / RES); values[i] = v; sum += v; }
        values[RES] = 0.0f;
        scaleFactor = RES / r;
        borderSize = (int)std::ceil(r - 0.5f);
        sum *= 2 * r / RES;
        Float normalization = 1.0f / sum;
        for (int i = 0; i < RES; ++i) values[i] *= normalization;
    }
    Float evalDiscretized(Float x) const { return values[std::min((int)std::fabs(x * scaleFactor), RES)]; }
};

/* ---- intersection record (skdtree.h fillIntersectionRecord<true>, absent) ---------- */
struct Intersection {
    Float t; V3 p; Frame geoFrame, shFrame; V2 uv; V3 dpdu, dpdv, wi;
    uint32_t prim, shape;
    V3 toWorld(V3 v) const { return shFrame.toWorld(v); }
    V3 toLocal(V3 v) const { return shFrame.toLocal(v); }
};

struct ShapeRec { uint32_t firstPrim, nPrims; int material, emitter; uint32_t flags; DiscreteDistribution areaDistr; Float surfaceArea = -1, invSurfaceArea = 0; };
struct EmitterRec { int type, shape; Float weight; V3 radiance, position; };

/* DirectSamplingRecord (records.inl, absent) */
struct DirectRec { V3 ref, refN, p, n, d; Float dist, pdf; int measure; /* 0 solid angle, 1 discrete */ int emitter; };

/* Resampler (rfilter.h, absent from the checkout: restated from the published Mitsuba 0.5 header; parity unpinned) with the
 * 2-lobed Lanczos filter of rfilters/lanczos.cpp, as envmap.cpp:166-172 configures it.  repeat: ERepeat, else EClamp. */
struct Resampler {
    int sourceRes, targetRes, taps; std::vector<int> start; std::vector<Float> weights; bool repeat;
    static Float lanczos2(Float x) {
        x = std::fabs(x);
        if (x < kEpsilon) return 1.0f; else if (x > 2.0f) return 0.0f;
        Float x1 = kPi * x, x2 = x1 / 2.0f;
        return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
    }
    Resampler(bool repeat_, int src, int trg) : sourceRes(src), targetRes(trg), repeat(repeat_) {
        Float filterRadius = 2.0f, scale = 1.0f, invScale = 1.0f;
        if (trg < src) { scale = (Float)src / (Float)trg; invScale = 1 / scale; filterRadius *= scale; }
        taps = (int)std::ceil(filterRadius * 2);
        start.resize(trg); weights.resize((size_t)taps * trg);
        for (int i = 0; i < trg; ++i) {
            Float center = (i + (Float)0.5f) / trg * src;
            start[i] = (int)std::floor(center - filterRadius + (Float)0.5f);
            Float sum = 0;
            for (int j = 0; j < taps; ++j) {
                Float pos = start[i] + j + (Float)0.5f - center;
                Float weight = lanczos2(pos * invScale);
                weights[(size_t)i * taps + j] = weight; sum += weight;
            }
            Float normalization = 1.0f / sum;
            for (int j = 0; j < taps; ++j) weights[(size_t)i * taps + j] *= normalization;
        }
    }
    /* resampleAndClamp over one line of V3 with the given strides, clamped to [0, inf) */
    void run(const V3 *src, size_t srcStride, V3 *trg, size_t trgStride) const {
        for (int i = 0; i < targetRes; ++i) {
            V3 result(0.f);
            for (int j = 0; j < taps; ++j) {
                int pos = start[i] + j;
                if (pos < 0 || pos >= sourceRes) {
                    if (repeat) { pos %= sourceRes; if (pos < 0) pos += sourceRes; }
                    else pos = std::min(std::max(pos, 0), sourceRes - 1);
                }
                result += src[(size_t)pos * srcStride] * weights[(size_t)i * taps + j];
            }
            trg[(size_t)i * trgStride] = V3(std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f));
        }
    }
};

struct EnvMap {
    int w = 0, h = 0; std::vector<V3> texels; Float scale = 1;
    /* TMIPMap (mipmap.h, absent: restated from the published Mitsuba 0.5 header; parity unpinned): level 0 = texels, every
     * further level half the size (rounded up), resampled from the unquantised level before and stored as half */
    struct Level { int w, h; std::vector<V3> texels; };
    std::vector<Level> pyramid;                /* levels 1 ... */
    Float weightLut[64];
    int levels() const { return 1 + (int)pyramid.size(); }
    void buildPyramid(const std::vector<V3> &level0Float) {
        for (int i = 0; i < 64; ++i) { Float r2 = (Float)i / (Float)63; weightLut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f); }
        std::vector<V3> cur = level0Float; int cw = w, ch = h;
        while (cw > 1 || ch > 1) {
            const int nw = std::max(1, (cw + 1) / 2), nh = std::max(1, (ch + 1) / 2);
            std::vector<V3> tmp((size_t)nw * ch), nxt((size_t)nw * nh);
            if (nw != cw) { Resampler r(true, cw, nw); for (int y = 0; y < ch; ++y) r.run(&cur[(size_t)y * cw], 1, &tmp[(size_t)y * nw], 1); }
            else tmp = cur;
            if (nh != ch) { Resampler r(false, ch, nh); for (int x = 0; x < nw; ++x) r.run(&tmp[x], nw, &nxt[x], nw); }
            else nxt = tmp;
            Level L; L.w = nw; L.h = nh; L.texels.resize(nxt.size());
            for (size_t i = 0; i < nxt.size(); ++i) L.texels[i] = V3(roundToHalf(nxt[i].x), roundToHalf(nxt[i].y), roundToHalf(nxt[i].z));
            pyramid.push_back(std::move(L));
            cur.swap(nxt); cw = nw; ch = nh;
        }
    }
    V3 texelL(int level, int x, int y) const {
        if (level == 0) return texel(x, y);
        const Level &L = pyramid[level - 1];
        x = x % L.w; if (x < 0) x += L.w;
        y = std::min(std::max(y, 0), L.h - 1);
        return L.texels[(size_t)y * L.w + x];
    }
    void sizeL(int level, int &lw, int &lh) const { if (level == 0) { lw = w; lh = h; } else { lw = pyramid[level - 1].w; lh = pyramid[level - 1].h; } }
    /* mipmap.h evalBilinear(level, uv) */
    V3 evalBilinear(int level, V2 uv) const {
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return V3(0.f);
        int lw, lh;
        if (level >= levels()) {                      /* evalBox on the coarsest level */
            sizeL(levels() - 1, lw, lh);
            return texelL(levels() - 1, (int)std::floor(uv.x * lw), (int)std::floor(uv.y * lh));
        }
        sizeL(level, lw, lh);
        Float u = uv.x * lw - 0.5f, v = uv.y * lh - 0.5f;
        int xPos = (int)std::floor(u), yPos = (int)std::floor(v);
        Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        return texelL(level, xPos, yPos) * dx2 * dy2 + texelL(level, xPos, yPos + 1) * dx2 * dy1
             + texelL(level, xPos + 1, yPos) * dx1 * dy2 + texelL(level, xPos + 1, yPos + 1) * dx1 * dy1;
    }
    /* mipmap.h evalEWA */
    V3 evalEWA(int level, V2 uv, Float A, Float B, Float C) const {
        if (!std::isfinite(A + B + C + uv.x + uv.y)) return V3(0.f);
        if (level >= levels()) return evalBilinear(level, uv);
        int lw, lh; sizeL(level, lw, lh);
        Float u = uv.x * lw - 0.5f, v = uv.y * lh - 0.5f;
        Float ratio0 = lw / (Float)w, ratio1 = lh / (Float)h;
        A /= ratio0 * ratio0; B /= ratio0 * ratio1; C /= ratio1 * ratio1;
        Float invDet = 1.0f / (-B * B + 4.0f * A * C), deltaU = 2.0f * std::sqrt(C * invDet), deltaV = 2.0f * std::sqrt(A * invDet);
        int u0 = (int)std::ceil(u - deltaU), u1 = (int)std::floor(u + deltaU);
        int v0 = (int)std::ceil(v - deltaV), v1 = (int)std::floor(v + deltaV);
        Float As = A * 64, Bs = B * 64, Cs = C * 64;
        V3 result(0.f); Float denominator = 0.0f;
        Float ddq = 2 * As, uu0 = (Float)u0 - u;
        for (int vt = v0; vt <= v1; ++vt) {
            const Float vv = (Float)vt - v;
            Float q = As * uu0 * uu0 + (Bs * uu0 + Cs * vv) * vv;
            Float dq = As * (2 * uu0 + 1) + Bs * vv;
            for (int ut = u0; ut <= u1; ++ut) {
                if (q < (Float)64) {
                    uint32_t qi = (uint32_t)q;
                    if (qi < 64) {
                        const Float weight = weightLut[(int)q];
                        result += texelL(level, ut, vt) * weight;
                        denominator += weight;
                    }
                }
                q += dq; dq += ddq;
            }
        }
        if (denominator == 0) return evalBilinear(level, uv);
        return result / denominator;
    }
    /* mipmap.h eval(uv, d0, d1): EWA filter type, maxAnisotropy 10 (envmap.cpp:141-142) */
    V3 evalFiltered(V2 uv, V2 d0, V2 d1) const {
        const Float maxAnisotropy = 10.0f;
        Float du0 = d0.x * w, dv0 = d0.y * h, du1 = d1.x * w, dv1 = d1.y * h;
        Float A = dv0 * dv0 + dv1 * dv1, B = -2.0f * (du0 * dv0 + du1 * dv1), C = du0 * du0 + du1 * du1, F = A * C - B * B * 0.25f;
        Float root = std::hypot(A - C, B), Aprime = 0.5f * (A + C - root), Cprime = 0.5f * (A + C + root),
              majorRadius = Aprime != 0 ? std::sqrt(F / Aprime) : 0, minorRadius = Cprime != 0 ? std::sqrt(F / Cprime) : 0;
        if (!(minorRadius > 0) || !(majorRadius > 0) || F < 0) {
            Float level = std::log2(std::max(majorRadius, kEpsilon));
            int ilevel = (int)std::floor(level);
            if (ilevel < 0) return evalBilinear(0, uv);
            Float a = level - ilevel;
            return evalBilinear(ilevel, uv) * (1.0f - a) + evalBilinear(ilevel + 1, uv) * a;
        }
        if (minorRadius * maxAnisotropy < majorRadius) {
            minorRadius = majorRadius / maxAnisotropy;
            Float theta = 0.5f * std::atan(B / (A - C)), sinTheta = std::sin(theta), cosTheta = std::cos(theta);
            Float a2 = majorRadius * majorRadius, b2 = minorRadius * minorRadius, sinTheta2 = sinTheta * sinTheta, cosTheta2 = cosTheta * cosTheta,
                  sin2Theta = 2 * sinTheta * cosTheta;
            A = a2 * cosTheta2 + b2 * sinTheta2; B = (a2 - b2) * sin2Theta; C = a2 * sinTheta2 + b2 * cosTheta2; F = a2 * b2;
        }
        Float sc = 1.0f / F;
        A *= sc; B *= sc; C *= sc;
        Float level = std::max((Float)0.0f, std::log2(minorRadius));
        int ilevel = (int)level;
        Float a = level - ilevel;
        if (majorRadius < 1 || !(A > 0 && C > 0)) return evalBilinear(ilevel, uv);
        return evalEWA(ilevel, uv, A, B, C) * (1.0f - a) + evalEWA(ilevel + 1, uv, A, B, C) * a;
    }
    Float toWorld[9], toLocal[9];
    V3 center; Float radius = 0;
    std::vector<float> cdfRows, cdfCols; std::vector<Float> rowWeights; Float normalization = 0; V2 pixelSize;

    V3 rot(const Float *R, V3 v) const { return V3(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z); }
    /* mipmap.h evalTexel: repeat in U, clamp in V (envmap.cpp:145-146) */
    V3 texel(int x, int y) const {
        x = x % w; if (x < 0) x += w;
        y = std::min(std::max(y, 0), h - 1);
        return texels[(size_t)y * w + x];
    }
    /* envmap.cpp:260-329 */
    void configure() {
        cdfCols.assign((size_t)(w + 1) * h, 0.f); cdfRows.assign(h + 1, 0.f); rowWeights.assign(h, 0.f);
        size_t colPos = 0, rowPos = 0; Float rowSum = 0.0f;
        cdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            Float colSum = 0;
            cdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) { colSum += luminance(texels[(size_t)y * w + x]); cdfCols[colPos++] = (float)colSum; }
            float normalization = 1.0f / (float)colSum;
            for (int x = 1; x < w; ++x) cdfCols[colPos - x - 1] *= normalization;
            cdfCols[colPos - 1] = 1.0f;
            Float weight = std::sin((y + 0.5f) * kPi / h);
            rowWeights[y] = weight;
            rowSum += colSum * weight;
            cdfRows[rowPos++] = (float)rowSum;
        }
        float norm = 1.0f / (float)rowSum;
        for (int y = 1; y < h; ++y) cdfRows[rowPos - y - 1] *= norm;
        cdfRows[rowPos - 1] = 1.0f;
        normalization = 1.0f / (rowSum * (2 * kPi / w) * (kPi / h));
        pixelSize = V2{ 2 * kPi / w, kPi / h };
    }
    /* envmap.cpp:653-658 */
    static uint32_t sampleReuse(const float *cdf, uint32_t size, Float &sample) {
        const float *entry = std::lower_bound(cdf, cdf + size + 1, (float)sample);
        uint32_t index = std::min((uint32_t)std::max((ptrdiff_t)0, entry - cdf - 1), size - 1);
        sample = (sample - (Float)cdf[index]) / (Float)(cdf[index + 1] - cdf[index]);
        return index;
    }
    void bilinearAt(Float u, Float v, V3 &value1, V3 &value2, int &yPos) const {
        int xPos = (int)std::floor(u); yPos = (int)std::floor(v);
        Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = v - yPos, dy2 = 1.0f - dy1;
        value1 = texel(xPos, yPos) * dx2 * dy2 + texel(xPos + 1, yPos) * dx1 * dy2;
        value2 = texel(xPos, yPos + 1) * dx2 * dy1 + texel(xPos + 1, yPos + 1) * dx1 * dy1;
    }
    /* envmap.cpp:567-600 */
    void internalSampleDirection(V2 sample, V3 &d, V3 &value, Float &pdf) const {
        uint32_t row = sampleReuse(cdfRows.data(), (uint32_t)h, sample.y),
                 col = sampleReuse(cdfCols.data() + (size_t)row * (w + 1), (uint32_t)w, sample.x);
        V2 tent = squareToTent(sample);
        V2 pos{ (Float)col + tent.x, (Float)row + tent.y };
        V3 value1, value2; int yPos;
        bilinearAt(pos.x, pos.y, value1, value2, yPos);
        value = (value1 + value2) * scale;
        pdf = (luminance(value1) * rowWeights[std::min(std::max(yPos, 0), h - 1)] +
               luminance(value2) * rowWeights[std::min(std::max(yPos + 1, 0), h - 1)]) * normalization;
        Float a = pixelSize.x * (pos.x + 0.5f), b = pixelSize.y * (pos.y + 0.5f);
        Float sinPhi = std::sin(a), cosPhi = std::cos(a), sinTheta = std::sin(b), cosTheta = std::cos(b);
        d = V3(sinPhi * sinTheta, cosTheta, -cosPhi * sinTheta);
        pdf /= std::max(std::fabs(sinTheta), kEpsilon);
    }
    /* envmap.cpp:603-633 */
    Float internalPdfDirection(V3 d) const {
        V2 uv{ std::atan2(d.x, -d.z) * kInvTwoPi, safe_acos(d.y) * kInvPi };
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return 0.0f;
        Float u = uv.x * w - 0.5f, v = uv.y * h - 0.5f;
        V3 value1, value2; int yPos;
        bilinearAt(u, v, value1, value2, yPos);
        Float sinTheta = safe_sqrt(1 - d.y * d.y);
        return (luminance(value1) * rowWeights[std::min(std::max(yPos, 0), h - 1)] +
                luminance(value2) * rowWeights[std::min(std::max(yPos + 1, 0), h - 1)]) * normalization / std::max(std::fabs(sinTheta), kEpsilon);
    }
    /* envmap.cpp:380-393,408-409 with hasDifferentials=false (mipmap.h evalBilinear) */
    V3 evalEnvironment(V3 worldDir) const {
        V3 v = rot(toLocal, worldDir);
        V2 uv{ std::atan2(v.x, -v.z) * kInvTwoPi, safe_acos(v.y) * kInvPi };
        if (!std::isfinite(uv.x) || !std::isfinite(uv.y)) return V3(0.f);
        Float u = uv.x * w - 0.5f, vv = uv.y * h - 0.5f;
        int xPos = (int)std::floor(u), yPos = (int)std::floor(vv);
        Float dx1 = u - xPos, dx2 = 1.0f - dx1, dy1 = vv - yPos, dy2 = 1.0f - dy1;
        V3 value = texel(xPos, yPos) * dx2 * dy2 + texel(xPos, yPos + 1) * dx2 * dy1
                 + texel(xPos + 1, yPos) * dx1 * dy2 + texel(xPos + 1, yPos + 1) * dx1 * dy1;
        return value * scale;
    }
    /* envmap.cpp:379-409: the filtered lookup for a ray with differentials (the primary ray) */
    V3 evalEnvironment(const Ray &ray) const {
        if (!ray.hasDifferentials) return evalEnvironment(ray.d);
        V3 v = rot(toLocal, ray.d);
        V2 uv{ std::atan2(v.x, -v.z) * kInvTwoPi, safe_acos(v.y) * kInvPi };
        V3 dvdx = rot(toLocal, ray.rxDirection) - v, dvdy = rot(toLocal, ray.ryDirection) - v;
        Float t1 = kInvTwoPi / (v.x * v.x + v.z * v.z), t2 = -kInvPi / std::max(safe_sqrt(1.0f - v.y * v.y), kEpsilon);
        V2 dudx{ t1 * (dvdx.z * v.x - dvdx.x * v.z), t2 * dvdx.y }, dudy{ t1 * (dvdy.z * v.x - dvdy.x * v.z), t2 * dvdy.y };
        return evalFiltered(uv, dudx, dudy) * scale;
    }
};

struct Sensor {
    M44 toWorld, sampleToCamera; Float nearClip, farClip; V3 dx, dy; V2 invRes; int cropW, cropH;
    /* perspective.cpp:126-163, sensor.cpp:101-107 */
    void configure(const kkd_sensor &s) {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) toWorld.m[i][j] = s.to_world[4 * i + j];
        nearClip = s.near_clip; farClip = s.far_clip; cropW = (int)s.crop_width; cropH = (int)s.crop_height;
        Float aspect = s.film_width / (Float)s.film_height;
        V2 relSize{ (Float)s.crop_width / (Float)s.film_width, (Float)s.crop_height / (Float)s.film_height };
        V2 relOffset{ (Float)s.crop_offset_x / (Float)s.film_width, (Float)s.crop_offset_y / (Float)s.film_height };
        M44 cameraToSample = M44::scale(1.0f / relSize.x, 1.0f / relSize.y, 1.0f)
            * M44::translate(-relOffset.x, -relOffset.y, 0.0f)
            * M44::scale(-0.5f, -0.5f * aspect, 1.0f)
            * M44::translate(-1.0f, -1.0f / aspect, 0.0f)
            * M44::perspective(s.xfov_deg, nearClip, farClip);
        sampleToCamera = cameraToSample.inverse();
        invRes = V2{ 1.0f / (Float)s.crop_width, 1.0f / (Float)s.crop_height };
        dx = sampleToCamera.point(V3(invRes.x, 0, 0)) - sampleToCamera.point(V3(0.f));
        dy = sampleToCamera.point(V3(0, invRes.y, 0)) - sampleToCamera.point(V3(0.f));
    }
    /* perspective.cpp:271-298 + RayDifferential::scaleDifferential (integrator.cpp:181; ray.h absent: d + (rxDirection - d) * amount) */
    Ray sampleRay(V2 pixelSample, Float diffScale = 1.0f) const {
        V3 nearP = sampleToCamera.point(V3(pixelSample.x * invRes.x, pixelSample.y * invRes.y, 0.0f));
        V3 d = normalize(nearP);
        Float invZ = 1.0f / d.z;
        Ray ray(toWorld.pointAffine(V3(0.f)), toWorld.vector(d), nearClip * invZ, farClip * invZ);
        ray.rxDirection = toWorld.vector(normalize(nearP + dx)); ray.ryDirection = toWorld.vector(normalize(nearP + dy));
        ray.rxDirection = ray.d + (ray.rxDirection - ray.d) * diffScale; ray.ryDirection = ray.d + (ray.ryDirection - ray.d) * diffScale;
        ray.hasDifferentials = true;
        return ray;
    }
};

struct RenderParams { uint32_t spp, nB, nL; bool hideEmitters; uint32_t blockSize, tileIndex, tileCount, sampleFirst; uint64_t seed;
                      uint32_t integrator = 0; bool strictNormals = false; int maxDepth = -1, rrDepth = 5;
                      uint32_t sampler = 0, samplerDimension = 0, sampleCountTotal = 0;
                      uint32_t total() const { return sampleCountTotal ? sampleCountTotal : spp; }                /* Sampler::getSampleCount() */
                      static uint32_t pow2ceil(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }
                      HaltonCfg halton;                                                                          /* filled by prepare() */
                      SobolCfg sobol;
                      void prepare(int cropW, int cropH) {                                                       /* Integrator::configureSampler, integrator.cpp:37-41 */
                          if (sampler == KKD_SAMPLER_HAMMERSLEY || sampler == KKD_SAMPLER_HAMMERSLEY_PLAIN)
                              halton.setFilmResolutionHammersley(cropW, cropH, sampler == KKD_SAMPLER_HAMMERSLEY, total());
                          if (sampler == KKD_SAMPLER_HALTON || sampler == KKD_SAMPLER_HALTON_PLAIN)
                              halton.setFilmResolution(cropW, cropH, sampler == KKD_SAMPLER_HALTON, integrator == KKD_INTEGRATOR_PATH ? 0u : (nB > 1 ? 1u : 0u) + (nL > 1 ? 1u : 0u));
                          if (sampler == KKD_SAMPLER_SOBOL)
                              sobol.setFilmResolution(cropW, cropH, integrator == KKD_INTEGRATOR_PATH ? 0u : (nB > 1 ? 1u : 0u) + (nL > 1 ? 1u : 0u));
                      }
                      void configure(SampleStream &rng, int X, int Y) const {                                    /* ldsampler.cpp:76-87 */
                          if (sampler == KKD_SAMPLER_LDSAMPLER) rng.useLdSampler(pow2ceil(total()), samplerDimension ? samplerDimension : 4u);
                          if (sampler == KKD_SAMPLER_STRATIFIED) {                                               /* stratified.cpp:62-79 */
                              uint32_t res = 1; while ((uint64_t)res * res < total()) ++res;
                              rng.useStratified(res, samplerDimension ? samplerDimension : 4u);
                          }
                          if (sampler >= KKD_SAMPLER_HALTON && sampler <= KKD_SAMPLER_HAMMERSLEY_PLAIN) rng.useHalton(&halton, X, Y);
                          if (sampler == KKD_SAMPLER_SOBOL) rng.useSobol(&sobol, X, Y);
                      } };
struct Stats { std::atomic<uint64_t> samples{0}, raysClosest{0}, raysShadow{0}, nodes{0}, tris{0}; };

class Scene {
public:
    MeshView mesh;
    std::vector<float> posStore, nrmStore, uvStore; std::vector<uint32_t> idxStore;
    std::vector<ShapeRec> shapes; std::vector<uint32_t> shapeOfPrim, firstPrimOfShape;
    std::vector<V3> tangU, tangV; std::vector<uint8_t> hasTang;   /* TriMesh::computeUVTangents */
    BSDFSet bsdfs; std::vector<EmitterRec> emitters; DiscreteDistribution emitterPDF;
    EnvMap env; int envEmitter = -1;
    Sensor sensor; RFilter filter; bool filmHasAlpha = false, hqEdges = false;
    KDTree kd; bool useBruteForce = false;

    bool load(const kkd_scene_desc *sd, std::string &err);

    /* ---- ray queries ---- */
    bool rayIntersect(const Ray &ray, Hit &hit, Stats *st) const {
        if (st) st->raysClosest++;
        if (useBruteForce) return kd.bruteForce(ray, false, hit);
        return kd.rayIntersectClosest(ray, hit);
    }
    bool rayOccluded(const Ray &ray, Stats *st) const {
        if (st) st->raysShadow++;
        if (useBruteForce) { Hit h; return kd.bruteForce(ray, true, h); }
        return kd.rayIntersectAny(ray);
    }
    void fillIntersection(const Ray &ray, const Hit &hit, Intersection &its) const;

    /* ---- emitters ---- */
    V3 emitterEvalArea(const EmitterRec &e, const Intersection &its, V3 d) const { /* area.cpp:104-109 */
        if (dot(its.shFrame.n, d) <= 0) return V3(0.f);
        return e.radiance;
    }
    V3 sampleEmitterDirect(DirectRec &dRec, V2 sample, Stats *st) const;
    Float pdfEmitterDirect(const DirectRec &dRec) const;
    Float pdfEmitterDiscrete(int e) const { return emitters[e].weight * emitterPDF.normalization; }
    void meshSamplePosition(const ShapeRec &s, V2 sample, V3 &p, V3 &n) const;
    bool envFillDirectRec(DirectRec &dRec, const Ray &ray) const { /* envmap.cpp:358-374 */
        Float nearT, farT;
        if (!bsphereRayIntersect(env.center, env.radius, ray.o, ray.d, nearT, farT) || nearT > 0 || farT < 0) return false;
        dRec.p = ray.at(farT); dRec.n = normalize(env.center - dRec.p); dRec.measure = 0; dRec.emitter = envEmitter;
        dRec.d = ray.d; dRec.dist = farT;
        return true;
    }

    /* ---- integrator ---- */
    V3 Li(const Ray &ray, SampleStream &rng, const RenderParams &rp, Float &alpha, Stats *st) const;
    V3 LiDirectPlugin(const Ray &ray, const Intersection &its, int bsdf, uint32_t btype, V3 refN, SampleStream &rng, const RenderParams &rp, Stats *st) const;
    V3 LiPath(const Ray &ray, SampleStream &rng, const RenderParams &rp, Float &alpha, Stats *st) const;
    void render(const RenderParams &rp, float *film, Stats *st, int nThreads) const;
};

inline void Scene::fillIntersection(const Ray &ray, const Hit &hit, Intersection &its) const {
    const uint32_t prim = hit.prim, sh = shapeOfPrim[prim];
    const ShapeRec &s = shapes[sh];
    const V3 b(1 - hit.u - hit.v, hit.u, hit.v);
    const uint32_t i0 = mesh.idx[3 * prim], i1 = mesh.idx[3 * prim + 1], i2 = mesh.idx[3 * prim + 2];
    const V3 p0 = mesh.P(i0), p1 = mesh.P(i1), p2 = mesh.P(i2);
    its.t = hit.t; its.prim = prim; its.shape = sh;
    its.p = p0 * b.x + p1 * b.y + p2 * b.z;
    V3 side1 = p1 - p0, side2 = p2 - p0;
    V3 faceNormal = cross(side1, side2);
    Float len = length(faceNormal);
    if (!isZero(faceNormal)) faceNormal = faceNormal / len;
    if (!hasTang.empty() && hasTang[prim]) { its.dpdu = tangU[prim]; its.dpdv = tangV[prim]; }
    else { its.dpdu = side1; its.dpdv = side2; }
    if (s.flags & KKD_SHAPE_HAS_NORMALS) {
        its.shFrame.n = normalize(mesh.N(i0) * b.x + mesh.N(i1) * b.y + mesh.N(i2) * b.z);
        if (dot(faceNormal, its.shFrame.n) < 0) faceNormal = -faceNormal;
    } else its.shFrame.n = faceNormal;
    its.geoFrame = Frame(faceNormal);
    if (s.flags & KKD_SHAPE_HAS_TEXCOORDS) {
        V2 t0 = mesh.UV(i0), t1 = mesh.UV(i1), t2 = mesh.UV(i2);
        its.uv = V2{ t0.x * b.x + t1.x * b.y + t2.x * b.z, t0.y * b.x + t1.y * b.y + t2.y * b.z };
    } else its.uv = V2{ b.y, b.z };
    /* computeShadingFrame (shape.h, absent): Gram-Schmidt of dpdu against n (cf. librender/bsdf.cpp:66-75) */
    its.shFrame.s = normalize(its.dpdu - its.shFrame.n * dot(its.shFrame.n, its.dpdu));
    its.shFrame.t = cross(its.shFrame.n, its.shFrame.s);
    its.wi = its.toLocal(-ray.d);
}

/* trimesh.cpp:416-427 + triangle.cpp:24-59 */
inline void Scene::meshSamplePosition(const ShapeRec &s, V2 sample, V3 &p, V3 &n) const {
    size_t local = s.areaDistr.sampleReuse(sample.y);
    uint32_t prim = s.firstPrim + (uint32_t)local;
    const uint32_t i0 = mesh.idx[3 * prim], i1 = mesh.idx[3 * prim + 1], i2 = mesh.idx[3 * prim + 2];
    const V3 p0 = mesh.P(i0), p1 = mesh.P(i1), p2 = mesh.P(i2);
    V2 bary = squareToUniformTriangle(sample);
    V3 sideA = p1 - p0, sideB = p2 - p0;
    p = p0 + (sideA * bary.x) + (sideB * bary.y);
    if (s.flags & KKD_SHAPE_HAS_NORMALS)
        n = normalize(mesh.N(i0) * (1.0f - bary.x - bary.y) + mesh.N(i1) * bary.x + mesh.N(i2) * bary.y);
    else
        n = normalize(cross(sideA, sideB));
}

/* scene.cpp:828-852 */
inline V3 Scene::sampleEmitterDirect(DirectRec &dRec, V2 sample, Stats *st) const {
    Float emPdf;
    size_t index = emitterPDF.sampleReuse(sample.x, emPdf);
    const EmitterRec &e = emitters[index];
    V3 value(0.f);
    dRec.pdf = 0;
    if (e.type == KKD_EMITTER_AREA) {
        /* area.cpp:158-173 -> shape.cpp:99-112 */
        const ShapeRec &s = shapes[e.shape];
        meshSamplePosition(s, sample, dRec.p, dRec.n);
        dRec.pdf = s.invSurfaceArea;
        dRec.d = dRec.p - dRec.ref;
        Float distSquared = lengthSquared(dRec.d);
        dRec.dist = std::sqrt(distSquared);
        dRec.d = dRec.d / dRec.dist;
        Float dp = absDot(dRec.d, dRec.n);
        dRec.pdf *= dp != 0 ? (distSquared / dp) : 0.0f;
        dRec.measure = 0;
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0 && dRec.pdf != 0) value = e.radiance / dRec.pdf;
        else dRec.pdf = 0.0f;
    } else if (e.type == KKD_EMITTER_POINT) {
        /* point.cpp:131-147 */
        dRec.p = e.position; dRec.pdf = 1.0f; dRec.measure = 1;
        dRec.d = dRec.p - dRec.ref;
        dRec.dist = length(dRec.d);
        Float invDist = 1.0f / dRec.dist;
        dRec.d *= invDist;
        dRec.n = V3(0.f);
        value = e.radiance * (invDist * invDist);
    } else {
        /* envmap.cpp:516-543 */
        V3 d, v; Float pdf;
        env.internalSampleDirection(sample, d, v, pdf);
        V3 wd = env.rot(env.toWorld, d);
        Float nearT, farT;
        if (isZero(v) || pdf == 0 || !bsphereRayIntersect(env.center, env.radius, dRec.ref, wd, nearT, farT) || nearT >= 0 || farT <= 0) {
            dRec.pdf = 0.0f;
        } else {
            dRec.pdf = pdf; dRec.p = dRec.ref + wd * farT; dRec.n = normalize(env.center - dRec.p);
            dRec.dist = farT; dRec.d = wd; dRec.measure = 0;
            value = v / pdf;
        }
    }
    if (dRec.pdf != 0) {
        Ray ray(dRec.ref, dRec.d, kEpsilon, dRec.dist * (1 - kShadowEpsilon));
        if (rayOccluded(ray, st)) return V3(0.f);
        dRec.emitter = (int)index;
        dRec.pdf *= emPdf;
        value = value / emPdf;
        return value;
    }
    return V3(0.f);
}

/* scene.cpp:949-952 */
inline Float Scene::pdfEmitterDirect(const DirectRec &dRec) const {
    const EmitterRec &e = emitters[dRec.emitter];
    Float pdf = 0;
    if (e.type == KKD_EMITTER_AREA) {        /* area.cpp:175-183 -> shape.cpp:114-123 */
        if (dot(dRec.d, dRec.refN) >= 0 && dot(dRec.d, dRec.n) < 0)
            pdf = shapes[e.shape].invSurfaceArea * (dRec.dist * dRec.dist) / absDot(dRec.d, dRec.n);
    } else if (e.type == KKD_EMITTER_POINT) {  /* point.cpp:149-151 */
        pdf = dRec.measure == 1 ? 1.0f : 0.0f;
    } else {                                  /* envmap.cpp:545-556 */
        pdf = env.internalPdfDirection(env.rot(env.toLocal, dRec.d));
    }
    return pdf * pdfEmitterDiscrete(dRec.emitter);
}

/* integrators/kk/direct.cpp:50-178 */
inline V3 Scene::Li(const Ray &ray, SampleStream &rng, const RenderParams &rp, Float &alpha, Stats *st) const {
    if (rp.integrator == KKD_INTEGRATOR_PATH) return LiPath(ray, rng, rp, alpha, st);
    struct SV { V3 value, bsdfValue; Float pdf, otherPdf; };
    V3 L(0.f);
    Hit hit;
    if (!rayIntersect(ray, hit, st)) {                     /* :55-60 */
        if (filmHasAlpha) alpha = 0;
        if (!rp.hideEmitters && envEmitter >= 0) return env.evalEnvironment(ray);
        return V3(0.f);
    }
    Intersection its; fillIntersection(ray, hit, its);
    const ShapeRec &shape = shapes[its.shape];
    const int bsdf = shape.material;
    if (shape.emitter >= 0 && !rp.hideEmitters)            /* :66-67 */
        L += emitterEvalArea(emitters[shape.emitter], its, -ray.d);

    /* DirectSamplingRecord(its): refN = 0 for transmissive / two-sided BSDFs (records.inl) */
    const uint32_t btype = bsdfs.getType(bsdf);
    V3 refN = (btype & (ETransmission | EBackSide)) ? V3(0.f) : its.shFrame.n;

    if (rp.integrator == KKD_INTEGRATOR_DIRECT) return L + LiDirectPlugin(ray, its, bsdf, btype, refN, rng, rp, st);

    std::vector<SV> brdfSamples, lightSamples;
    /* integrateBrdf :74-125 */
    for (uint32_t i = 0; i < rp.nB; ++i) {
        V2 u = rp.nB > 1 ? rng.next2DArray(0, rp.nB, i) : rng.next2D();                  /* :80-85; arrays requested at :44-47 */
        V3 woLocal; Float bsdfPdf; uint32_t sampledType;
        V3 bsdfVal = bsdfs.sample(bsdf, its.wi, u, woLocal, bsdfPdf, sampledType);
        if (isZero(bsdfVal)) continue;
        const V3 wo = its.toWorld(woLocal);
        if (dot(its.geoFrame.n, wo) * woLocal.z <= 0) continue;
        Ray bsdfRay(its.p, wo, kEpsilon, std::numeric_limits<Float>::infinity());
        DirectRec dRec; dRec.ref = its.p; dRec.refN = refN;
        V3 value; Hit bh;
        if (rayIntersect(bsdfRay, bh, st)) {
            Intersection bIts; fillIntersection(bsdfRay, bh, bIts);
            const ShapeRec &bs = shapes[bIts.shape];
            if (bs.emitter < 0) continue;
            value = emitterEvalArea(emitters[bs.emitter], bIts, -bsdfRay.d);
            /* setQuery (records.inl) */
            dRec.p = bIts.p; dRec.n = bIts.shFrame.n; dRec.d = bsdfRay.d; dRec.dist = bIts.t; dRec.measure = 0; dRec.emitter = bs.emitter;
        } else {
            if (envEmitter < 0) continue;
            value = env.evalEnvironment(bsdfRay.d);
            if (!envFillDirectRec(dRec, bsdfRay)) continue;
        }
        Float lightPdf = !(sampledType & EDelta) ? pdfEmitterDirect(dRec) : 0;
        brdfSamples.push_back(SV{ value, bsdfVal, bsdfPdf, lightPdf });
    }
    /* integrateLight :127-160 */
    {
        DirectRec dRec; dRec.ref = its.p; dRec.refN = refN;
        for (uint32_t i = 0; i < rp.nL; ++i) {
            V2 u = rp.nL > 1 ? rng.next2DArray(rp.nB > 1 ? 1 : 0, rp.nL, i) : rng.next2D();
            if (!btype) continue;                          /* :140 (samples are still consumed, :132-137) */
            V3 value = sampleEmitterDirect(dRec, u, st);
            if (isZero(value)) continue;
            const EmitterRec &e = emitters[dRec.emitter];
            V3 woLocal = its.toLocal(dRec.d);
            const V3 bsdfVal = bsdfs.eval(bsdf, its.wi, woLocal);
            const bool onSurface = e.type != KKD_EMITTER_POINT;   /* area.cpp:70, envmap.cpp:107 */
            Float bsdfPdf = onSurface ? bsdfs.pdf(bsdf, its.wi, woLocal) : 0;
            if (!isZero(bsdfVal) && (dot(its.geoFrame.n, dRec.d) * woLocal.z > 0))
                lightSamples.push_back(SV{ value, bsdfVal, dRec.pdf, bsdfPdf });
        }
    }
    /* computeMIS :162-178 */
    Float cBrdf = rp.nB / (Float)(rp.nB + rp.nL), cLight = rp.nL / (Float)(rp.nB + rp.nL);
    Float bsdfFrag = 1 / (Float)rp.nB, lightFrag = 1 / (Float)rp.nL;
    for (const SV &s : brdfSamples) {
        Float weight = cBrdf * s.pdf / (cBrdf * s.pdf + cLight * s.otherPdf);
        L += weight * s.value * s.bsdfValue * bsdfFrag;
    }
    for (const SV &s : lightSamples) {
        Float weight = cLight * s.pdf / (cBrdf * s.otherPdf + cLight * s.pdf);
        L += weight * s.value * s.bsdfValue * lightFrag;
    }
    return L;
}

/* MIDirectIntegrator::Li after the emitted-radiance terms (reference src/integrators/direct/direct.cpp:176-311, depth 1,
 * no adaptive query): emitter sampling first, then BSDF sampling, power heuristic with the sample-count fractions. */
inline V3 Scene::LiDirectPlugin(const Ray &ray, const Intersection &its, int bsdf, uint32_t btype, V3 refN, SampleStream &rng,
                                const RenderParams &rp, Stats *st) const {
    V3 L(0.f);
    if (rp.strictNormals && dot(ray.d, its.geoFrame.n) * its.wi.z >= 0) return L;      /* :177-190 */
    const Float sum = (Float)(rp.nB + rp.nL);
    const Float weightBSDF = 1 / (Float)rp.nB, weightLum = 1 / (Float)rp.nL, fracBSDF = rp.nB / sum, fracLum = rp.nL / sum;   /* :129-136 */
    auto miWeight = [](Float a, Float b) { a *= a; b *= b; return a / (a + b); };        /* :313-316 */
    DirectRec dRec; dRec.ref = its.p; dRec.refN = refN;
    for (uint32_t i = 0; i < rp.nL; ++i) {                                               /* :219-250 */
        V2 u = rp.nL > 1 ? rng.next2DArray(0, rp.nL, i) : rng.next2D();                  /* :139-143, 210-214 */
        if (!(btype & ESmooth)) continue;
        V3 value = sampleEmitterDirect(dRec, u, st);
        if (isZero(value)) continue;
        const EmitterRec &e = emitters[dRec.emitter];
        V3 woLocal = its.toLocal(dRec.d);
        const V3 bsdfVal = bsdfs.eval(bsdf, its.wi, woLocal);
        if (!isZero(bsdfVal) && (!rp.strictNormals || dot(its.geoFrame.n, dRec.d) * woLocal.z > 0)) {
            Float bsdfPdf = (e.type != KKD_EMITTER_POINT) ? bsdfs.pdf(bsdf, its.wi, woLocal) : 0;
            const Float weight = miWeight(dRec.pdf * fracLum, bsdfPdf * fracBSDF) * weightLum;
            L += value * bsdfVal * weight;
        }
    }
    for (uint32_t i = 0; i < rp.nB; ++i) {                                               /* :256-309 */
        V2 u = rp.nB > 1 ? rng.next2DArray(rp.nL > 1 ? 1 : 0, rp.nB, i) : rng.next2D();
        V3 woLocal; Float bsdfPdf; uint32_t sampledType;
        V3 bsdfVal = bsdfs.sample(bsdf, its.wi, u, woLocal, bsdfPdf, sampledType);
        if (isZero(bsdfVal)) continue;
        const V3 wo = its.toWorld(woLocal);
        if (rp.strictNormals && dot(its.geoFrame.n, wo) * woLocal.z <= 0) continue;
        Ray bsdfRay(its.p, wo, kEpsilon, std::numeric_limits<Float>::infinity());
        DirectRec bRec; bRec.ref = its.p; bRec.refN = refN;
        V3 value; Hit bh;
        if (rayIntersect(bsdfRay, bh, st)) {
            Intersection bIts; fillIntersection(bsdfRay, bh, bIts);
            const ShapeRec &bs = shapes[bIts.shape];
            if (bs.emitter < 0) continue;
            value = emitterEvalArea(emitters[bs.emitter], bIts, -bsdfRay.d);
            bRec.p = bIts.p; bRec.n = bIts.shFrame.n; bRec.d = bsdfRay.d; bRec.dist = bIts.t; bRec.measure = 0; bRec.emitter = bs.emitter;
        } else {
            if (envEmitter < 0) continue;        /* sampledType == ENull does not occur with the BSDFs of this path */
            value = env.evalEnvironment(bsdfRay.d);
            if (!envFillDirectRec(bRec, bsdfRay)) continue;
        }
        const Float lumPdf = !(sampledType & EDelta) ? pdfEmitterDirect(bRec) : 0;
        const Float weight = miWeight(bsdfPdf * fracBSDF, lumPdf * fracLum) * weightBSDF;
        L += value * bsdfVal * weight;
    }
    return L;
}

/* MIPathTracer::Li, integrators/path/path.cpp:119-294 (parameters: MonteCarloIntegrator, librender/integrator.cpp:190-233).
 * Sample dimensions (counter-based stream instead of the sequential sampler): the vertex at depth d uses dimensions
 * 2 + 6 (d - 1) ... : emitter sample (2), BSDF sample (2), Russian roulette (1), one unused -- for d = 1 these are the
 * dimensions of the direct plugin with one emitter and one BSDF sample, so maxDepth = 2 reproduces it sample by sample. */
inline V3 Scene::LiPath(const Ray &r, SampleStream &rng, const RenderParams &rp, Float &alpha, Stats *st) const {
    auto miWeight = [](Float a, Float b) { a *= a; b *= b; return a / (a + b); };        /* :296-300 */
    Ray ray = r;
    V3 Li(0.f), throughput(1.f);
    Float eta = 1.0f;
    bool scattered = false;
    int depth = 1;                                                                       /* RadianceQueryRecord::newQuery */
    bool emitted = true;                                                                 /* rRec.type & EEmittedRadiance */
    Hit hit; Intersection its;
    bool valid = rayIntersect(ray, hit, st);                                             /* :129 */
    if (valid) fillIntersection(ray, hit, its);
    else if (filmHasAlpha) alpha = 0;                                                    /* integrator.cpp: rRec.alpha */
    ray.mint = kEpsilon;
    while (depth <= rp.maxDepth || rp.maxDepth < 0) {                                    /* :135 */
        if (!valid) {                                                                    /* :136-143 */
            if (emitted && (!rp.hideEmitters || scattered) && envEmitter >= 0) Li += throughput * env.evalEnvironment(ray);   /* differentials: the primary ray only */
            break;
        }
        const ShapeRec &shape = shapes[its.shape];
        const int bsdf = shape.material;
        if (shape.emitter >= 0 && emitted && (!rp.hideEmitters || scattered))            /* :148-150 */
            Li += throughput * emitterEvalArea(emitters[shape.emitter], its, -ray.d);
        if ((depth >= rp.maxDepth && rp.maxDepth > 0) || (rp.strictNormals && dot(ray.d, its.geoFrame.n) * its.wi.z >= 0)) break;   /* :156-165 */
        rng.dim = 2u + 6u * (uint32_t)(depth - 1); rng.hdim = rng.dim; rng.dim2D = 1u + 2u * (uint32_t)(depth - 1);   /* request numbers are fixed per vertex */
        const uint32_t btype = bsdfs.getType(bsdf);
        DirectRec dRec; dRec.ref = its.p; dRec.refN = (btype & (ETransmission | EBackSide)) ? V3(0.f) : its.shFrame.n;
        {                                                                                /* :172-200 */
            V2 u = rng.next2D();
            if (btype & ESmooth) {
                V3 value = sampleEmitterDirect(dRec, u, st);
                if (!isZero(value)) {
                    const EmitterRec &e = emitters[dRec.emitter];
                    V3 woLocal = its.toLocal(dRec.d);
                    const V3 bsdfVal = bsdfs.eval(bsdf, its.wi, woLocal);
                    if (!isZero(bsdfVal) && (!rp.strictNormals || dot(its.geoFrame.n, dRec.d) * woLocal.z > 0)) {
                        Float bsdfPdf = (e.type != KKD_EMITTER_POINT) ? bsdfs.pdf(bsdf, its.wi, woLocal) : 0;
                        Li += throughput * value * bsdfVal * miWeight(dRec.pdf, bsdfPdf);
                    }
                }
            }
        }
        V2 u = rng.next2D();                                                             /* :206-219 */
        V3 woLocal; Float bsdfPdf; uint32_t sampledType;
        V3 bsdfWeight = bsdfs.sample(bsdf, its.wi, u, woLocal, bsdfPdf, sampledType);
        if (isZero(bsdfWeight)) break;
        scattered = true;                                                                /* no ENull component among the BSDFs of this path */
        const V3 wo = its.toWorld(woLocal);
        if (rp.strictNormals && dot(its.geoFrame.n, wo) * woLocal.z <= 0) break;
        /* bRec.eta: relative index along the sampled direction (dielectric.cpp:300-308), 1 for reflection */
        Float bRecEta = 1.0f;
        if (sampledType == EDeltaTransmission) { const Float e = bsdfs.mats[bsdf].eta; bRecEta = its.wi.z > 0 ? e : 1 / e; }
        bool hitEmitter = false;
        V3 value(0.f);
        ray = Ray(its.p, wo, kEpsilon, std::numeric_limits<Float>::infinity());          /* :225 */
        valid = rayIntersect(ray, hit, st);
        if (valid) {                                                                     /* :226-232 */
            fillIntersection(ray, hit, its);
            const ShapeRec &bs = shapes[its.shape];
            if (bs.emitter >= 0) {
                value = emitterEvalArea(emitters[bs.emitter], its, -ray.d);
                dRec.p = its.p; dRec.n = its.shFrame.n; dRec.d = ray.d; dRec.dist = its.t; dRec.measure = 0; dRec.emitter = bs.emitter;
                hitEmitter = true;
            }
        } else {                                                                         /* :233-248 */
            if (envEmitter < 0) break;
            value = env.evalEnvironment(ray.d);
            if (!envFillDirectRec(dRec, ray)) break;
            hitEmitter = true;
        }
        throughput = throughput * bsdfWeight;                                            /* :252-253 */
        eta *= bRecEta;
        if (hitEmitter) {                                                                /* :257-264 */
            const Float lumPdf = !(sampledType & EDelta) ? pdfEmitterDirect(dRec) : 0;
            Li += throughput * value * miWeight(bsdfPdf, lumPdf);
        }
        if (!valid) break;                                                               /* :272-274 */
        emitted = false;
        if (depth++ >= rp.rrDepth) {                                                     /* :276-286 */
            Float q = std::min(maxComp(throughput) * eta * eta, (Float)0.95f);
            if (rng.next1D() >= q) break;
            throughput = throughput / q;
        }
    }
    return Li;
}

/* integrator.cpp:140-188 over all tiles of imageproc.cpp:28-78; the film receives every block as
 * Film::put(block) does (hdrfilm.cpp:391-393): here samples are splatted into per-thread tile blocks
 * with border and merged under a lock (renderproc.cpp:142-149). */
inline void Scene::render(const RenderParams &rp, float *film, Stats *st, int nThreads) const {
    const int W = sensor.cropW, H = sensor.cropH, bs = rp.blockSize ? (int)rp.blockSize : 32;
    const int border = filter.borderSize;
    /* Film::hasHighQualityEdges: the image process covers offset (-border, -border), size + 2 border (renderproc.cpp:163-171) */
    const int edge = hqEdges ? border : 0, EW = W + 2 * edge, EH = H + 2 * edge;
    const int nbx = (EW + bs - 1) / bs, nby = (EH + bs - 1) / bs, nTiles = nbx * nby;
    std::memset(film, 0, sizeof(float) * 5 * (size_t)W * H);
    std::atomic<int> next{0};
    std::mutex filmMutex;
    auto worker = [&]() {
        const int bw = bs + 2 * border;
        std::vector<float> block((size_t)bw * bw * 5);
        Float wx[16], wy[16];
        for (;;) {
            int t = next.fetch_add(1);
            if (t >= nTiles) break;
            if (rp.tileCount > 1 && (uint32_t)t % rp.tileCount != rp.tileIndex) continue;
            const int ox = (t % nbx) * bs - edge, oy = (t / nbx) * bs - edge;
            const int sx = std::min(bs, W + edge - ox), sy = std::min(bs, H + edge - oy);
            const int sizeX = sx + 2 * border, sizeY = sy + 2 * border;
            std::fill(block.begin(), block.end(), 0.f);
            for (int py = 0; py < sy; ++py) for (int px = 0; px < sx; ++px) {
                const int X = ox + px, Y = oy + py;
                for (uint32_t j = 0; j < rp.spp; ++j) {
                    SampleStream rng(rp.seed, (uint32_t)((Y + edge) * EW + (X + edge)), rp.sampleFirst + j);
                    rp.configure(rng, X, Y);
                    V2 jit = rng.next2D();
                    V2 samplePos{ (Float)X + jit.x, (Float)Y + jit.y };
                    Ray ray = sensor.sampleRay(samplePos, 1.0f / std::sqrt((Float)rp.total()));   /* integrator.cpp:144-145, 181 */
                    Float alpha = 1.0f;
                    V3 spec = Li(ray, rng, rp, alpha, st);
                    if (st) st->samples++;
                    /* ImageBlock::put (imageblock.h, absent) */
                    Float value[5] = { spec.x, spec.y, spec.z, alpha, 1.0f };
                    bool bad = false;
                    for (int c = 0; c < 5; ++c) if (!std::isfinite(value[c]) || value[c] < 0) bad = true;
                    if (bad) continue;
                    const Float posx = samplePos.x - 0.5f - (ox - border), posy = samplePos.y - 0.5f - (oy - border);
                    const int minx = std::max((int)std::ceil(posx - filter.radius), 0), miny = std::max((int)std::ceil(posy - filter.radius), 0);
                    const int maxx = std::min((int)std::floor(posx + filter.radius), sizeX - 1), maxy = std::min((int)std::floor(posy + filter.radius), sizeY - 1);
                    for (int x = minx, i = 0; x <= maxx; ++x) wx[i++] = filter.evalDiscretized(x - posx);
                    for (int y = miny, i = 0; y <= maxy; ++y) wy[i++] = filter.evalDiscretized(y - posy);
                    for (int y = miny, yr = 0; y <= maxy; ++y, ++yr)
                        for (int x = minx, xr = 0; x <= maxx; ++x, ++xr) {
                            const Float w = wx[xr] * wy[yr];
                            float *dst = &block[((size_t)y * bw + x) * 5];
                            for (int c = 0; c < 5; ++c) dst[c] += w * value[c];
                        }
                }
            }
            std::lock_guard<std::mutex> g(filmMutex);
            for (int y = 0; y < sizeY; ++y) {
                const int Y = oy - border + y; if (Y < 0 || Y >= H) continue;
                for (int x = 0; x < sizeX; ++x) {
                    const int X = ox - border + x; if (X < 0 || X >= W) continue;
                    const float *src = &block[((size_t)y * bw + x) * 5];
                    float *dst = &film[((size_t)Y * W + X) * 5];
                    for (int c = 0; c < 5; ++c) dst[c] += src[c];
                }
            }
        }
    };
    if (nThreads <= 1) worker();
    else { std::vector<std::thread> th; for (int i = 0; i < nThreads; ++i) th.emplace_back(worker); for (auto &t : th) t.join(); }
}

inline bool Scene::load(const kkd_scene_desc *sd, std::string &err) {
    if (!sd || sd->abi_version != KKD_ABI_VERSION) { err = "bad scene description / ABI version"; return false; }
    posStore.assign(sd->positions, sd->positions + 3 * sd->n_verts);
    idxStore.assign(sd->indices, sd->indices + 3 * sd->n_tris);
    if (sd->normals) nrmStore.assign(sd->normals, sd->normals + 3 * sd->n_verts);
    if (sd->texcoords) uvStore.assign(sd->texcoords, sd->texcoords + 2 * sd->n_verts);
    mesh.pos = posStore.data(); mesh.idx = idxStore.data();
    mesh.nrm = nrmStore.empty() ? nullptr : nrmStore.data(); mesh.uv = uvStore.empty() ? nullptr : uvStore.data();
    mesh.nVerts = sd->n_verts; mesh.nTris = sd->n_tris;
    for (uint64_t i = 0; i < 3 * sd->n_tris; ++i) if (idxStore[i] >= sd->n_verts) { err = "vertex index out of range"; return false; }
    shapes.resize(sd->n_shapes); shapeOfPrim.assign(sd->n_tris, 0); firstPrimOfShape.resize(sd->n_shapes);
    uint64_t expect = 0;
    for (uint32_t s = 0; s < sd->n_shapes; ++s) {
        const kkd_shape &k = sd->shapes[s];
        if (k.first_prim != expect) { err = "shapes must tile the triangle array in order"; return false; }
        expect += k.n_prims;
        if (k.material_id < 0 || (uint32_t)k.material_id >= sd->n_materials) { err = "material id out of range"; return false; }
        ShapeRec &r = shapes[s];
        r.firstPrim = k.first_prim; r.nPrims = k.n_prims; r.material = k.material_id; r.emitter = k.emitter_id; r.flags = k.flags;
        if ((r.flags & KKD_SHAPE_HAS_NORMALS) && !mesh.nrm) { err = "shape wants normals but none given"; return false; }
        if ((r.flags & KKD_SHAPE_HAS_TEXCOORDS) && !mesh.uv) { err = "shape wants texcoords but none given"; return false; }
        firstPrimOfShape[s] = k.first_prim;
        for (uint32_t p = 0; p < k.n_prims; ++p) shapeOfPrim[k.first_prim + p] = s;
    }
    if (expect != sd->n_tris) { err = "shapes do not cover all triangles"; return false; }
    /* TriMesh::computeUVTangents (trimesh.cpp:687-747), always attempted (trimesh.cpp:389) */
    if (mesh.uv) {
        tangU.assign(sd->n_tris, V3(0.f)); tangV.assign(sd->n_tris, V3(0.f)); hasTang.assign(sd->n_tris, 0);
        for (uint32_t s = 0; s < sd->n_shapes; ++s) if (shapes[s].flags & KKD_SHAPE_HAS_TEXCOORDS)
            for (uint32_t p = shapes[s].firstPrim; p < shapes[s].firstPrim + shapes[s].nPrims; ++p) {
                uint32_t i0 = mesh.idx[3 * p], i1 = mesh.idx[3 * p + 1], i2 = mesh.idx[3 * p + 2];
                V3 dP1 = mesh.P(i1) - mesh.P(i0), dP2 = mesh.P(i2) - mesh.P(i0);
                V2 a = mesh.UV(i0), b = mesh.UV(i1), c = mesh.UV(i2);
                V2 dUV1{ b.x - a.x, b.y - a.y }, dUV2{ c.x - a.x, c.y - a.y };
                V3 n = cross(dP1, dP2); Float len = length(n);
                hasTang[p] = 1;
                if (len == 0) continue;
                Float det = dUV1.x * dUV2.y - dUV1.y * dUV2.x;
                if (det == 0) coordinateSystem(n / len, tangU[p], tangV[p]);
                else {
                    Float invDet = 1.0f / det;
                    tangU[p] = (dP1 * dUV2.y - dP2 * dUV1.y) * invDet;
                    tangV[p] = (dP1 * (-dUV2.x) + dP2 * dUV1.x) * invDet;
                }
            }
    }
    /* materials */
    bsdfs.mats.resize(sd->n_materials);
    for (uint32_t i = 0; i < sd->n_materials; ++i) {
        const kkd_material &k = sd->materials[i]; Material &m = bsdfs.mats[i];
        m.type = k.type; m.distr = k.distribution; m.sampleVisible = k.sample_visible != 0; m.nonlinear = k.nonlinear != 0;
        m.nested[0] = k.nested[0]; m.nested[1] = k.nested[1];
        m.alphaU = k.alpha_u; m.alphaV = k.alpha_v; m.eta = k.eta;
        m.reflectance = V3(k.reflectance[0], k.reflectance[1], k.reflectance[2]);
        m.specRefl = V3(k.specular_reflectance[0], k.specular_reflectance[1], k.specular_reflectance[2]);
        m.specTrans = V3(k.specular_transmittance[0], k.specular_transmittance[1], k.specular_transmittance[2]);
        m.condEta = V3(k.cond_eta[0], k.cond_eta[1], k.cond_eta[2]); m.condK = V3(k.cond_k[0], k.cond_k[1], k.cond_k[2]);
        if (m.type < 0 || m.type > 4) { err = "unknown BSDF type"; return false; }
        if (m.type == KKD_BSDF_ROUGHPLASTIC && m.distr == KKD_DISTR_PHONG) { err = "roughplastic/phong unsupported"; return false; }
        if (m.type == KKD_BSDF_TWOSIDED && (m.nested[0] < 0 || (uint32_t)m.nested[0] >= sd->n_materials || (m.nested[1] >= 0 && (uint32_t)m.nested[1] >= sd->n_materials))) { err = "twosided: bad nested id"; return false; }
    }
    bsdfs.configure();
    /* emitters + scene emitter PDF (scene.cpp:376-380) */
    emitters.resize(sd->n_emitters); emitterPDF = DiscreteDistribution(); envEmitter = -1;
    for (uint32_t i = 0; i < sd->n_emitters; ++i) {
        const kkd_emitter &k = sd->emitters[i]; EmitterRec &e = emitters[i];
        e.type = k.type; e.shape = k.shape_id; e.weight = k.sampling_weight;
        e.radiance = V3(k.radiance[0], k.radiance[1], k.radiance[2]); e.position = V3(k.position[0], k.position[1], k.position[2]);
        if (e.type == KKD_EMITTER_AREA) {
            if (e.shape < 0 || (uint32_t)e.shape >= sd->n_shapes) { err = "area emitter: bad shape id"; return false; }
            ShapeRec &s = shapes[e.shape];
            if (s.surfaceArea < 0) {   /* trimesh.cpp:392-407, triangle.cpp:61-67 */
                for (uint32_t p = s.firstPrim; p < s.firstPrim + s.nPrims; ++p) {
                    V3 p0 = mesh.P(mesh.idx[3 * p]), p1 = mesh.P(mesh.idx[3 * p + 1]), p2 = mesh.P(mesh.idx[3 * p + 2]);
                    s.areaDistr.append(0.5f * length(cross(p1 - p0, p2 - p0)));
                }
                s.surfaceArea = s.areaDistr.normalize(); s.invSurfaceArea = 1.0f / s.surfaceArea;
            }
        } else if (e.type == KKD_EMITTER_ENVMAP) envEmitter = (int)i;
        emitterPDF.append(e.weight);
    }
    if (sd->n_emitters) emitterPDF.normalize();
    if (envEmitter >= 0) {
        const kkd_envmap &k = sd->envmap;
        if (!k.width || !k.height || !k.texels) { err = "envmap emitter without texels"; return false; }
        env.w = (int)k.width; env.h = (int)k.height; env.scale = k.scale;
        env.texels.resize((size_t)env.w * env.h);
        std::vector<V3> level0(env.texels.size());
        for (size_t i = 0; i < env.texels.size(); ++i) {
            level0[i] = V3(k.texels[3 * i], k.texels[3 * i + 1], k.texels[3 * i + 2]);
            env.texels[i] = V3(roundToHalf(k.texels[3 * i]), roundToHalf(k.texels[3 * i + 1]), roundToHalf(k.texels[3 * i + 2]));
        }
        env.buildPyramid(level0);
        for (int i = 0; i < 9; ++i) env.toWorld[i] = k.to_world[i];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) env.toLocal[3 * i + j] = k.to_world[3 * j + i];
        env.center = V3(k.bsphere_center[0], k.bsphere_center[1], k.bsphere_center[2]); env.radius = k.bsphere_radius;
        env.configure();
    }
    sensor.configure(sd->sensor);
    filmHasAlpha = sd->sensor.film_has_alpha != 0; hqEdges = sd->sensor.film_high_quality_edges != 0;
    filter.configure(sd->sensor.rfilter_type, sd->sensor.rfilter_radius, sd->sensor.rfilter_param);
    if (filter.borderSize > 6) { err = "filter radius too large"; return false; }
    kd.build(mesh, shapeOfPrim, firstPrimOfShape);
    return true;
}

} /* namespace ora */
