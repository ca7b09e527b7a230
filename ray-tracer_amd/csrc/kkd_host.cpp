/*
 * kkd_host.cpp -- scene flattening and table construction on the host (runs once per scene upload).
 *
 * What each part replaces in the reference (paths relative to /root/reference/src):
 *   shape map / global primitive ids      librender/skdtree.cpp:49-70
 *   emitter PDF                            librender/scene.cpp:376-380 (+ pmf.h DiscreteDistribution)
 *   triangle area CDF                      librender/trimesh.cpp:392-407, libcore/triangle.cpp:61-67
 *   BSDF configure()                       bsdfs/diffuse.cpp:93-104, roughconductor.cpp:228-250,
 *                                          roughplastic.cpp:255-307, dielectric.cpp:184-202, twosided.cpp:84-106,
 *                                          librender/bsdf.cpp:105-130 (ensureEnergyConservation)
 *   rough transmittance                    bsdfs/rtrans.h:183-280 (the data/microfacet/*.dat file it reads is absent
 *                                          from the reference tree; the table is integrated here from its definition,
 *                                          utils/rdielprec.cpp:40-112 + bsdfs/roughdielectric.cpp:424-511)
 *   EnvironmentMap::configure              emitters/envmap.cpp:260-329 (marginal / conditional CDF)
 *   perspective sensor                     sensors/perspective.cpp:126-163, librender/sensor.cpp:101-107
 *   reconstruction filter table            libcore/rfilter.cpp:37-55, rfilters/*.cpp
 */
#include "kkd_host.h"
#include "kkd_sobol_dirnums.h"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <zlib.h>

namespace kkd {

/* ---- half conversion (round to nearest even), as OpenEXR's half used by TMIPMap<Spectrum, SpectrumHalf> ---- */
uint16_t floatToHalf(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, e8 = (x >> 23) & 0xFFu;
    uint32_t mant = x & 0x7FFFFFu;
    if (e8 == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0u));
    const int32_t e = (int32_t)e8 - 112;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        mant |= 0x800000u;
        const int shift = 14 - e;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (mant >> 13);
    const uint32_t rem = mant & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
}
float halfToFloat(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu;
    uint32_t mant = h & 0x3FFu, x;
    if (e == 0) {
        if (!mant) x = sign;
        else {
            int sh = 0;
            while (!(mant & 0x400u)) { mant <<= 1; ++sh; }
            x = sign | ((uint32_t)(113 - sh) << 23) | ((mant & 0x3FFu) << 13);
        }
    } else if (e == 31) x = sign | 0x7F800000u | (mant << 13);
    else x = sign | ((e + 112u) << 23) | (mant << 13);
    float f; std::memcpy(&f, &x, 4); return f;
}

/* ---- reconstruction filter (libcore/rfilter.cpp:37-55) ---- */
static float filterEval(int type, float radius, const float *param, float x) {
    auto cubic = [](float x, float B, float C) {
        x = std::fabs(x);
        const float x2 = x * x, x3 = x2 * x;
        if (x < 1) return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * x3 + (-18 + 12 * B + 6 * C) * x2 + (6 - 2 * B));
        if (x < 2) return 1.0f / 6.0f * ((-B - 6 * C) * x3 + (6 * B + 30 * C) * x2 + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
        return 0.0f;
    };
    switch (type) {
        case KKD_RFILTER_GAUSSIAN: {   /* rfilters/gaussian.cpp:52-57 */
            const float alpha = -1.0f / (2.0f * param[0] * param[0]);
            return std::max(0.0f, std::exp(alpha * x * x) - std::exp(alpha * radius * radius));
        }
        case KKD_RFILTER_BOX: return std::fabs(x) <= radius ? 1.0f : 0.0f;            /* rfilters/box.cpp:44-46 */
        case KKD_RFILTER_TENT: return std::max(0.0f, 1.0f - std::fabs(x / radius));   /* rfilters/tent.cpp */
        case KKD_RFILTER_MITCHELL: return cubic(x, param[0], param[1]);               /* rfilters/mitchell.cpp */
        case KKD_RFILTER_CATMULLROM: return cubic(x, 0.0f, 0.5f);                     /* rfilters/catmullrom.cpp */
        case KKD_RFILTER_LANCZOS: {                                                   /* rfilters/lanczos.cpp */
            x = std::fabs(x);
            if (x < kEpsilon) return 1.0f;
            if (x > radius) return 0.0f;
            const float x1 = kPi * x, x2 = x1 / radius;
            return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
        }
    }
    return 0.0f;
}
void configureFilter(int type, float radius, const float *param, DFilter &f) {
    const int RES = 31;   /* MTS_FILTER_RESOLUTION */
    float sum = 0.0f;
    for (int i = 0; i < RES; ++i) { f.values[i] = filterEval(type, radius, param, (radius * i) / RES); sum += f.values[i]; }
    f.values[RES] = 0.0f;
    f.radius = radius; f.scaleFactor = RES / radius; f.border = (int)std::ceil(radius - 0.5f); f.pad = 0;
    sum *= 2 * radius / RES;
    const float normalization = 1.0f / sum;
    for (int i = 0; i < RES; ++i) f.values[i] *= normalization;
}

/* ---- rough transmittance by quadrature over the microfacet-normal sampling square -------------------
 * T(mu) = E_m[(1-F(wi.m)) G1(wo,m)] with m drawn from the visible-normal density
 *       = int (1-F) G1(wo,m) G1(wi,m) |wi.m| D(m) / mu dm.
 * Substituting the full-distribution warp (xi1 -> tan^2(theta_m), xi2 -> phi) turns D(m) cos(theta_m) dm
 * into dxi1 dxi2, so T = int_0^1 int_0^1 (1-F) G1o G1i |wi.m| / (mu cos(theta_m)) dxi1 dxi2. */
namespace {
struct MF {
    int type; double a;
    double D(double mx, double my, double mz) const {
        if (mz <= 0) return 0;
        const double c2 = mz * mz, ex = (mx * mx + my * my) / (a * a) / c2;
        double r;
        if (type == KKD_DISTR_BECKMANN) r = std::exp(-ex) / (M_PI * a * a * c2 * c2);
        else { const double root = (1 + ex) * c2; r = 1.0 / (M_PI * a * a * root * root); }
        if (r * mz < 1e-20) r = 0;
        return r;
    }
    /* bsdfs/microfacet.h:475-512 */
    double G1(double vx, double vy, double vz, double mx, double my, double mz) const {
        if ((vx * mx + vy * my + vz * mz) * vz <= 0) return 0;
        const double s2 = 1 - vz * vz;
        const double tanT = s2 <= 0 ? 0.0 : std::fabs(std::sqrt(s2) / vz);
        if (tanT == 0) return 1;
        if (type == KKD_DISTR_BECKMANN) {
            const double x = 1.0 / (a * tanT);
            if (x >= 1.6) return 1;
            return (3.535 * x + 2.181 * x * x) / (1.0 + 2.276 * x + 2.577 * x * x);
        }
        const double root = a * tanT;
        return 2.0 / (1.0 + std::sqrt(1.0 + root * root));
    }
};

void gaussLegendre01(int n, std::vector<double> &x, std::vector<double> &w) {
    x.resize(n); w.resize(n);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 1;
        for (int it = 0; it < 64; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 0; j < n; ++j) { const double p3 = p2; p2 = p1; p1 = ((2.0 * j + 1) * z * p2 - j * p3) / (j + 1); }
            pp = n * (z * p1 - p2) / (z * z - 1);
            const double dz = p1 / pp; z -= dz;
            if (std::fabs(dz) < 1e-15) break;
        }
        x[i] = 0.5 * (1 - z); x[n - 1 - i] = 0.5 * (1 + z);
        w[i] = w[n - 1 - i] = 1.0 / ((1 - z * z) * pp * pp);
    }
}

double transmittanceAt(const MF &d, double eta, double mu, const std::vector<double> &gx, const std::vector<double> &gw) {
    const double wix = std::sqrt(std::max(0.0, 1 - mu * mu)), wiz = mu;
    const int nPhi = 192;
    double total = 0;
    /* split xi1 at 0.9 so the GGX / Beckmann tails are resolved */
    const double seg[4] = { 0.0, 0.9, 0.999, 1.0 };
    for (int s = 0; s < 3; ++s) for (size_t i = 0; i < gx.size(); ++i) {
        const double xi = seg[s] + (seg[s + 1] - seg[s]) * gx[i], wgt = (seg[s + 1] - seg[s]) * gw[i];
        const double tan2 = d.type == KKD_DISTR_BECKMANN ? -d.a * d.a * std::log(1 - xi) : d.a * d.a * xi / (1 - xi);
        const double cosT = 1.0 / std::sqrt(1 + tan2), sinT = std::sqrt(std::max(0.0, 1 - cosT * cosT));
        double ring = 0;
        for (int ip = 0; ip < nPhi; ++ip) {
            const double phi = M_PI * (ip + 0.5) / nPhi;   /* half circle, mirrored */
            const double mx = sinT * std::cos(phi), my = sinT * std::sin(phi), mz = cosT;
            if (d.D(mx, my, mz) == 0) continue;
            const double c = wix * mx + wiz * mz;
            const double scale = c > 0 ? 1 / eta : eta, ct2 = 1 - (1 - c * c) * scale * scale;
            if (ct2 <= 0) continue;                         /* total internal reflection */
            const double ci = std::fabs(c), ct = std::sqrt(ct2);
            const double Rs = (ci - eta * ct) / (ci + eta * ct), Rp = (eta * ci - ct) / (eta * ci + ct);
            const double F = 0.5 * (Rs * Rs + Rp * Rp), cosTt = c > 0 ? -ct : ct;
            const double e = cosTt < 0 ? 1 / eta : eta, k = c * e + cosTt;   /* refract(), libcore/util.cpp:741-747 */
            const double wox = mx * k - wix * e, woy = my * k, woz = mz * k - wiz * e;
            if (wiz * woz >= 0) continue;
            ring += (1 - F) * d.G1(wox, woy, woz, mx, my, mz) * d.G1(wix, 0, wiz, mx, my, mz) * ci / (mu * mz);
        }
        total += wgt * ring / nPhi;
    }
    return total;
}

/* libcore/spline.cpp:23-60 */
double cubicInterp(double x, const double *v, int size) {
    if (!(x >= 0 && x <= 1)) return 0;
    double t = x * (size - 1);
    const int k = std::max(0, std::min((int)t, size - 2));
    const double f0 = v[k], f1 = v[k + 1];
    const double d0 = k > 0 ? 0.5 * (v[k + 1] - v[k - 1]) : v[k + 1] - v[k];
    const double d1 = k + 2 < size ? 0.5 * (v[k + 2] - v[k]) : v[k + 1] - v[k];
    t -= k;
    const double t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}
} /* namespace */

void roughTransmittanceTable(int distr, float alpha, float eta, float *trans100, float *diffTrans) {
    struct Entry { float t[100]; float d; };
    static std::map<std::tuple<int, float, float>, Entry> cache;
    static std::mutex mtx;
    std::lock_guard<std::mutex> g(mtx);
    const auto key = std::make_tuple(distr, alpha, eta);
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<double> gx, gw; gaussLegendre01(96, gx, gw);
        MF d{ distr, (double)alpha };
        double vals[100];
        Entry e;
        for (int i = 0; i < 100; ++i) {
            double t = i / 99.0; if (i == 0) t = 0.1 / 99.0;          /* utils/rdielprec.cpp:88-95 */
            vals[i] = transmittanceAt(d, (double)eta, t * t * t * t, gx, gw);
            e.t[i] = (float)vals[i];
        }
        /* diffuse transmittance 2 int_0^1 mu T(mu) dmu through the interpolant (rdielprec.cpp:58-62) */
        double acc = 0;
        for (size_t i = 0; i < gx.size(); ++i) acc += gw[i] * 2 * gx[i] * cubicInterp(std::pow(gx[i], 0.25), vals, 100);
        e.d = (float)acc;
        it = cache.emplace(key, e).first;
    }
    std::memcpy(trans100, it->second.t, sizeof(float) * 100);
    *diffTrans = it->second.d;
}

/* ---- scene flattening ---- */
static void conserve(float *c) {   /* librender/bsdf.cpp:105-130 for a constant texture */
    const float m = std::max(c[0], std::max(c[1], c[2]));
    if (m > 1.0f) { const float s = 0.99f * (1.0f / m); c[0] *= s; c[1] *= s; c[2] *= s; }
}
static float lum(const float *c) { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }

static void mat4mul(const double a[16], const double b[16], double r[16]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
        r[4 * i + j] = s;
    }
}
static bool mat4inv(const double m[16], double out[16]) {
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][j + 4] = i == j; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0) return false;
        for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = a[i][j + 4];
    return true;
}

#define KKD_FAIL(code, msg) do { err = (msg); if (errCode) *errCode = (code); return false; } while (0)

/* MIP pyramid of the environment map: what TMIPMap's constructor builds for envmap.cpp:166-182 -- every level half the size
 * of the one before (rounded up), resampled along x (repeat) and then y (clamp) as Bitmap::resample does (bitmap.cpp:2221-2320)
 * with the 2-lobed Lanczos filter (rfilters/lanczos.cpp), clamped to [0, inf), from the unquantised level before, stored as
 * half.  TMIPMap and Resampler live in headers the checkout does not hold (mipmap.h, rfilter.h): restated from the published
 * Mitsuba 0.5 sources, parity unpinned. */
namespace {
struct LineResampler {
    int sourceRes, targetRes, taps; bool repeat; std::vector<int> start; std::vector<float> weights;
    static float lanczos2(float x) {
        x = std::fabs(x);
        if (x < 1e-4f) return 1.0f; else if (x > 2.0f) return 0.0f;
        const float x1 = kPi * x, x2 = x1 / 2.0f;
        return (std::sin(x1) * std::sin(x2)) / (x1 * x2);
    }
    LineResampler(bool rep, int src, int trg) : sourceRes(src), targetRes(trg), repeat(rep) {
        float filterRadius = 2.0f, invScale = 1.0f;
        if (trg < src) { const float scale = (float)src / (float)trg; invScale = 1 / scale; filterRadius *= scale; }
        taps = (int)std::ceil(filterRadius * 2);
        start.resize(trg); weights.resize((size_t)taps * trg);
        for (int i = 0; i < trg; ++i) {
            const float center = (i + 0.5f) / trg * src;
            start[i] = (int)std::floor(center - filterRadius + 0.5f);
            float sum = 0;
            for (int j = 0; j < taps; ++j) { const float wgt = lanczos2((start[i] + j + 0.5f - center) * invScale); weights[(size_t)i * taps + j] = wgt; sum += wgt; }
            const float normalization = 1.0f / sum;
            for (int j = 0; j < taps; ++j) weights[(size_t)i * taps + j] *= normalization;
        }
    }
    void run(const float *src, size_t srcStride, float *trg, size_t trgStride) const {   /* strides in RGB triples */
        for (int i = 0; i < targetRes; ++i) {
            float r[3] = { 0.f, 0.f, 0.f };
            for (int j = 0; j < taps; ++j) {
                int pos = start[i] + j;
                if (pos < 0 || pos >= sourceRes) {
                    if (repeat) { pos %= sourceRes; if (pos < 0) pos += sourceRes; }
                    else pos = std::min(std::max(pos, 0), sourceRes - 1);
                }
                const float wgt = weights[(size_t)i * taps + j], *sp = src + (size_t)pos * srcStride * 3;
                /* (x + y * w) with separate roundings, as the oracle's vector arithmetic does */
                for (int c = 0; c < 3; ++c) { const float prod = sp[c] * wgt; r[c] = r[c] + prod; }
            }
            for (int c = 0; c < 3; ++c) trg[(size_t)i * trgStride * 3 + c] = std::max(r[c], 0.0f);
        }
    }
};
}
static void parallelLines(size_t n, const std::function<void(size_t)> &f) {
    const size_t nThreads = std::min<size_t>(std::max(1u, std::min(std::thread::hardware_concurrency(), 16u)), std::max<size_t>(1, n / 64));
    if (nThreads <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nThreads; ++t) pool.emplace_back([&, t] { for (size_t i = n * t / nThreads, e = n * (t + 1) / nThreads; i < e; ++i) f(i); });
    for (auto &th : pool) th.join();
}
static void buildEnvPyramid(const float *level0, int w, int h, HostScene &hs) {
    hs.envEwaLut.resize(64);
    for (int i = 0; i < 64; ++i) { const float r2 = (float)i / 63.0f; hs.envEwaLut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f); }
    hs.env.nLevels = 1; hs.env.levelOfs[0] = 0; hs.env.levelW[0] = (uint16_t)w; hs.env.levelH[0] = (uint16_t)h;
    std::vector<float> cur(level0, level0 + (size_t)w * h * 3), tmp, nxt;
    int cw = w, ch = h;
    while (cw > 1 || ch > 1) {
        const int nw = std::max(1, (cw + 1) / 2), nh = std::max(1, (ch + 1) / 2);
        if (nw != cw) {
            tmp.assign((size_t)nw * ch * 3, 0.f);
            const LineResampler r(true, cw, nw);
            parallelLines((size_t)ch, [&](size_t y) { r.run(&cur[y * cw * 3], 1, &tmp[y * nw * 3], 1); });
        } else tmp = cur;
        if (nh != ch) {
            nxt.assign((size_t)nw * nh * 3, 0.f);
            const LineResampler r(false, ch, nh);
            parallelLines((size_t)nw, [&](size_t x) { r.run(&tmp[x * 3], (size_t)nw, &nxt[x * 3], (size_t)nw); });
        } else nxt = tmp;
        const int l = hs.env.nLevels++;
        const size_t ofs = hs.envTexels.size() / 4;
        hs.env.levelOfs[l] = (uint32_t)ofs; hs.env.levelW[l] = (uint16_t)nw; hs.env.levelH[l] = (uint16_t)nh;
        hs.envTexels.resize((ofs + (size_t)nw * nh) * 4);
        for (size_t i = 0; i < (size_t)nw * nh; ++i) {
            for (int c = 0; c < 3; ++c) hs.envTexels[(ofs + i) * 4 + c] = floatToHalf(nxt[i * 3 + c]);
            hs.envTexels[(ofs + i) * 4 + 3] = 0;
        }
        cur.swap(nxt); cw = nw; ch = nh;
    }
}

bool flattenScene(const kkd_scene_desc *sd, HostScene &hs, std::string &err, int *errCode, bool buildBvh) {
    if (!sd) KKD_FAIL(KKD_ERR_INVALID, "scene description is NULL");
    if (sd->abi_version != KKD_ABI_VERSION) KKD_FAIL(KKD_ERR_INVALID, "scene description has a different ABI version");
    if (!sd->positions || !sd->indices || !sd->n_verts || !sd->n_tris) KKD_FAIL(KKD_ERR_INVALID, "scene has no geometry");
    if (!sd->n_shapes || !sd->shapes || !sd->n_materials || !sd->materials) KKD_FAIL(KKD_ERR_INVALID, "scene has no shapes or materials");
    if (sd->n_emitters && !sd->emitters) KKD_FAIL(KKD_ERR_INVALID, "emitters pointer is NULL");
    /* device code indexes the vertex and index tables with 32-bit element offsets (3 * i) */
    if (sd->n_verts > 0xFFFFFFFFull / 3 || sd->n_tris > 0xFFFFFFFFull / 3) KKD_FAIL(KKD_ERR_INVALID, "more than 2^32 / 3 vertices or triangles");
    hs.nVerts = sd->n_verts; hs.nTris = sd->n_tris;
    hs.hasNormals = sd->normals != nullptr; hs.hasTexcoords = sd->texcoords != nullptr;
    for (uint64_t i = 0; i < 3 * sd->n_tris; ++i)
        if (sd->indices[i] >= sd->n_verts) KKD_FAIL(KKD_ERR_INVALID, "vertex index out of range");

    /* shapes: must tile the triangle array in addShape order (skdtree.cpp:49-70) */
    hs.shapes.resize(sd->n_shapes); hs.shapeOfPrim.assign(sd->n_tris, 0);
    uint64_t expect = 0;
    for (uint32_t s = 0; s < sd->n_shapes; ++s) {
        const kkd_shape &k = sd->shapes[s];
        if (k.first_prim != expect) KKD_FAIL(KKD_ERR_INVALID, "shapes must tile the triangle array in order");
        expect += k.n_prims;
        if (expect > sd->n_tris) KKD_FAIL(KKD_ERR_INVALID, "shape exceeds the triangle array");
        if (k.material_id < 0 || (uint32_t)k.material_id >= sd->n_materials) KKD_FAIL(KKD_ERR_INVALID, "material id out of range");
        if (k.emitter_id >= (int32_t)sd->n_emitters) KKD_FAIL(KKD_ERR_INVALID, "emitter id out of range");
        if ((k.flags & KKD_SHAPE_HAS_NORMALS) && !sd->normals) KKD_FAIL(KKD_ERR_INVALID, "shape wants normals but none given");
        if ((k.flags & KKD_SHAPE_HAS_TEXCOORDS) && !sd->texcoords) KKD_FAIL(KKD_ERR_INVALID, "shape wants texcoords but none given");
        DShape &d = hs.shapes[s];
        d.firstPrim = k.first_prim; d.nPrims = k.n_prims; d.material = k.material_id; d.emitter = k.emitter_id < 0 ? -1 : k.emitter_id;
        d.flags = k.flags; d.pad[0] = d.pad[1] = d.pad[2] = 0;
        for (uint32_t p = 0; p < k.n_prims; ++p) hs.shapeOfPrim[k.first_prim + p] = s;
    }
    if (expect != sd->n_tris) KKD_FAIL(KKD_ERR_INVALID, "shapes do not cover all triangles");

    /* materials */
    hs.materials.resize(sd->n_materials); hs.rtTables.clear();
    for (uint32_t i = 0; i < sd->n_materials; ++i) {
        const kkd_material &k = sd->materials[i]; DMaterial &m = hs.materials[i];
        std::memset(&m, 0, sizeof(m));
        m.type = k.type; m.distr = k.distribution; m.sampleVisible = k.sample_visible != 0; m.nonlinear = k.nonlinear != 0;
        m.nested0 = k.nested[0]; m.nested1 = k.nested[1]; m.tableOffset = -1;
        m.alphaU = std::max(k.alpha_u, 1e-4f); m.alphaV = std::max(k.alpha_v, 1e-4f);   /* microfacet.h:70-71 */
        m.eta = k.eta;
        for (int c = 0; c < 3; ++c) {
            m.refl[c] = k.reflectance[c]; m.specRefl[c] = k.specular_reflectance[c]; m.specTrans[c] = k.specular_transmittance[c];
            m.condEta[c] = k.cond_eta[c]; m.condK[c] = k.cond_k[c];
        }
        if (m.distr < 0 || m.distr > KKD_DISTR_PHONG) KKD_FAIL(KKD_ERR_INVALID, "unknown microfacet distribution");
        if (m.distr == KKD_DISTR_PHONG) m.sampleVisible = 0;   /* microfacet.h:85-93 */
        switch (m.type) {
            case KKD_BSDF_DIFFUSE: {
                conserve(m.refl);
                const bool nz = std::max(m.refl[0], std::max(m.refl[1], m.refl[2])) > 0;
                m.typeFlags = nz ? (EDiffuseReflection | EFrontSide) : 0u;
                break;
            }
            case KKD_BSDF_ROUGHCONDUCTOR:
                conserve(m.specRefl);
                m.typeFlags = EGlossyReflection | EFrontSide;
                break;
            case KKD_BSDF_ROUGHPLASTIC: {
                if (m.distr == KKD_DISTR_PHONG) KKD_FAIL(KKD_ERR_UNSUPPORTED, "roughplastic with the phong distribution is not supported");
                if (!(m.eta > 0.f) || m.eta == 1.f) KKD_FAIL(KKD_ERR_INVALID, "roughplastic: bad relative index of refraction");
                conserve(m.specRefl); conserve(m.refl);
                m.alphaV = m.alphaU;
                const float dAvg = lum(m.refl), sAvg = lum(m.specRefl);
                m.specSamplingWeight = sAvg / (dAvg + sAvg);
                m.invEta2 = 1.0f / (m.eta * m.eta);
                m.tableOffset = (int32_t)hs.rtTables.size();
                hs.rtTables.resize(hs.rtTables.size() + 100);
                float dOut, dIn, inner[100];
                roughTransmittanceTable(m.distr, m.alphaU, m.eta, hs.rtTables.data() + m.tableOffset, &dOut);
                roughTransmittanceTable(m.distr, m.alphaU, 1.0f / m.eta, inner, &dIn);
                m.fdr = 1.0f - std::min(1.0f, std::max(0.0f, dIn));
                m.typeFlags = EGlossyReflection | EDiffuseReflection | EFrontSide;
                break;
            }
            case KKD_BSDF_DIELECTRIC:
                if (!(m.eta > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "dielectric: bad relative index of refraction");
                conserve(m.specRefl); conserve(m.specTrans);
                m.typeFlags = EDeltaReflection | EDeltaTransmission | EFrontSide | EBackSide;
                break;
            case KKD_BSDF_TWOSIDED:
                if (m.nested0 < 0 || (uint32_t)m.nested0 >= sd->n_materials || (m.nested1 >= 0 && (uint32_t)m.nested1 >= sd->n_materials))
                    KKD_FAIL(KKD_ERR_INVALID, "twosided: bad nested material id");
                if (m.nested1 < 0) m.nested1 = m.nested0;
                break;
            default: KKD_FAIL(KKD_ERR_INVALID, "unknown BSDF type");
        }
    }
    for (DMaterial &m : hs.materials) if (m.type == KKD_BSDF_TWOSIDED) {   /* twosided.cpp:84-106 */
        const DMaterial &a = hs.materials[m.nested0], &b = hs.materials[m.nested1];
        if (a.type == KKD_BSDF_TWOSIDED || b.type == KKD_BSDF_TWOSIDED) KKD_FAIL(KKD_ERR_UNSUPPORTED, "twosided cannot nest twosided");
        if ((a.typeFlags | b.typeFlags) & ETransmission) KKD_FAIL(KKD_ERR_INVALID, "twosided: only BRDFs can be nested");
        m.typeFlags = 0;
        if (a.typeFlags) m.typeFlags |= (a.typeFlags & ~EBackSide) | EFrontSide;
        if (b.typeFlags) m.typeFlags |= (b.typeFlags & ~EFrontSide) | EBackSide;
    }

    /* emitters, scene emitter PDF, area CDFs */
    hs.emitters.resize(sd->n_emitters); hs.emitterCdf.assign(1, 0.0f); hs.areaCdf.clear(); hs.envEmitter = -1;
    std::vector<int64_t> shapeCdf(sd->n_shapes, -1); std::vector<float> shapeInvArea(sd->n_shapes, 0.f);
    std::vector<uint32_t> shapeGuide(sd->n_shapes, 0u);
    hs.areaGuide.assign(1, 0u);
    for (uint32_t i = 0; i < sd->n_emitters; ++i) {
        const kkd_emitter &k = sd->emitters[i]; DEmitter &e = hs.emitters[i];
        std::memset(&e, 0, sizeof(e));
        e.type = k.type; e.shape = k.shape_id; e.weight = k.sampling_weight;
        for (int c = 0; c < 3; ++c) { e.radiance[c] = k.radiance[c]; e.position[c] = k.position[c]; }
        if (!(e.weight >= 0.f)) KKD_FAIL(KKD_ERR_INVALID, "emitter sampling weight must be >= 0");
        if (e.type == KKD_EMITTER_AREA) {
            if (e.shape < 0 || (uint32_t)e.shape >= sd->n_shapes) KKD_FAIL(KKD_ERR_INVALID, "area emitter: bad shape id");
            const DShape &s = hs.shapes[e.shape];
            if (s.emitter != (int32_t)i) KKD_FAIL(KKD_ERR_INVALID, "area emitter and its shape do not reference each other");
            if (shapeCdf[e.shape] < 0) {
                shapeCdf[e.shape] = (int64_t)hs.areaCdf.size();
                const size_t base = hs.areaCdf.size();
                hs.areaCdf.push_back(0.0f);
                for (uint32_t p = s.firstPrim; p < s.firstPrim + s.nPrims; ++p) {
                    const float *a = sd->positions + 3 * (size_t)sd->indices[3 * (size_t)p], *b = sd->positions + 3 * (size_t)sd->indices[3 * (size_t)p + 1],
                                *c = sd->positions + 3 * (size_t)sd->indices[3 * (size_t)p + 2];
                    const f3 s1(b[0] - a[0], b[1] - a[1], b[2] - a[2]), s2(c[0] - a[0], c[1] - a[1], c[2] - a[2]);
                    hs.areaCdf.push_back(hs.areaCdf.back() + 0.5f * length(cross(s1, s2)));
                }
                const float sum = hs.areaCdf.back();
                if (!(sum > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "area emitter on a shape with zero surface area");
                const float norm = 1.0f / sum;
                for (size_t j = base + 1; j < hs.areaCdf.size(); ++j) hs.areaCdf[j] *= norm;
                hs.areaCdf.back() = 1.0f;
                shapeInvArea[e.shape] = 1.0f / sum;
                if (s.nPrims >= kAreaGuideMinPrims) {   /* guide table of this CDF's search (kkd_types.h) */
                    shapeGuide[e.shape] = (uint32_t)hs.areaGuide.size() + 1u;
                    const float *cdf = hs.areaCdf.data() + base; const size_t n = (size_t)s.nPrims + 1;
                    for (uint32_t b = 0; b <= kAreaGuide; ++b)
                        hs.areaGuide.push_back(std::min<uint32_t>((uint32_t)(std::lower_bound(cdf, cdf + n, (float)b / (float)kAreaGuide) - cdf), (uint32_t)n - 1));
                }
            }
            e.cdfOffset = (uint32_t)shapeCdf[e.shape]; e.invArea = shapeInvArea[e.shape]; e.guideOffset = shapeGuide[e.shape];
            e.firstPrim = s.firstPrim; e.nPrims = s.nPrims; e.shapeFlags = s.flags;
        } else if (e.type == KKD_EMITTER_POINT) {
            e.shape = -1;
        } else if (e.type == KKD_EMITTER_ENVMAP) {
            if (hs.envEmitter >= 0) KKD_FAIL(KKD_ERR_INVALID, "more than one environment emitter");
            hs.envEmitter = (int)i; e.shape = -1;
        } else KKD_FAIL(KKD_ERR_INVALID, "unknown emitter type");
        hs.emitterCdf.push_back(hs.emitterCdf.back() + e.weight);
    }
    for (uint32_t s = 0; s < sd->n_shapes; ++s)
        if (hs.shapes[s].emitter >= 0 && hs.emitters[hs.shapes[s].emitter].type != KKD_EMITTER_AREA)
            KKD_FAIL(KKD_ERR_INVALID, "shape references a non-area emitter");
    hs.emitterNormalization = 0.f;
    if (sd->n_emitters) {
        const float sum = hs.emitterCdf.back();
        if (!(sum > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "all emitter sampling weights are zero");
        hs.emitterNormalization = 1.0f / sum;
        for (size_t j = 1; j < hs.emitterCdf.size(); ++j) hs.emitterCdf[j] *= hs.emitterNormalization;
        hs.emitterCdf.back() = 1.0f;
    }
    {   /* guide table of the emitter CDF search (kkd_types.h) */
        const size_t n = hs.emitterCdf.size();
        hs.emitterGuide.resize(kEmitterGuide + 1);
        for (uint32_t b = 0; b <= kEmitterGuide; ++b)
            hs.emitterGuide[b] = std::min<uint32_t>((uint32_t)(std::lower_bound(hs.emitterCdf.begin(), hs.emitterCdf.end(), (float)b / (float)kEmitterGuide) - hs.emitterCdf.begin()), (uint32_t)n - 1);
    }

    /* environment map (envmap.cpp:260-329) */
    std::memset(&hs.env, 0, sizeof(hs.env));
    if (hs.envEmitter >= 0) {
        const kkd_envmap &k = sd->envmap;
        if (!k.width || !k.height || !k.texels) KKD_FAIL(KKD_ERR_INVALID, "envmap emitter without texels");
        if (k.width > 65535 || k.height > 65535) KKD_FAIL(KKD_ERR_INVALID, "envmap larger than 65535 pixels per side");   /* envmap.cpp:161-163 */
        const int w = (int)k.width, h = (int)k.height;
        hs.env.w = w; hs.env.h = h; hs.env.scale = k.scale;
        hs.envTexels.resize((size_t)w * h * 4);
        hs.envCdfCols.assign((size_t)(w + 1) * h, 0.f); hs.envCdfRows.assign(h + 1, 0.f); hs.envRowWeights.assign(h, 0.f);
        size_t colPos = 0, rowPos = 0; float rowSum = 0.0f;
        hs.envCdfRows[rowPos++] = 0;
        for (int y = 0; y < h; ++y) {
            float colSum = 0;
            hs.envCdfCols[colPos++] = 0;
            for (int x = 0; x < w; ++x) {
                float rgb[3];
                for (int c = 0; c < 3; ++c) {
                    const float v = k.texels[((size_t)y * w + x) * 3 + c];
                    if (!(v >= 0.f) || !std::isfinite(v)) KKD_FAIL(KKD_ERR_INVALID, "envmap texel is negative or not finite");
                    const uint16_t hb = floatToHalf(v);
                    hs.envTexels[((size_t)y * w + x) * 4 + c] = hb;
                    rgb[c] = halfToFloat(hb);
                }
                hs.envTexels[((size_t)y * w + x) * 4 + 3] = 0;
                colSum += lum(rgb);
                hs.envCdfCols[colPos++] = colSum;
            }
            const float normalization = 1.0f / colSum;
            for (int x = 1; x < w; ++x) hs.envCdfCols[colPos - x - 1] *= normalization;
            hs.envCdfCols[colPos - 1] = 1.0f;
            const float weight = std::sin((y + 0.5f) * kPi / h);
            hs.envRowWeights[y] = weight;
            rowSum += colSum * weight;
            hs.envCdfRows[rowPos++] = rowSum;
        }
        if (!(rowSum > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "envmap is black");
        const float norm = 1.0f / rowSum;
        for (int y = 1; y < h; ++y) hs.envCdfRows[rowPos - y - 1] *= norm;
        hs.envCdfRows[rowPos - 1] = 1.0f;
        {   /* guide tables (kkd_types.h): lower_bound of the bucket boundaries, rows and every row's columns */
            auto guide = [](const float *cdf, size_t n, uint32_t G, uint32_t *out) {
                for (uint32_t b = 0; b <= G; ++b) out[b] = (uint32_t)(std::lower_bound(cdf, cdf + n, (float)b / (float)G) - cdf);
                out[G] = std::min<uint32_t>(out[G], (uint32_t)n - 1);   /* cdf[n - 1] = 1 >= every sample */
            };
            hs.envGuideRows.resize(kEnvGuideRows + 1);
            guide(hs.envCdfRows.data(), (size_t)h + 1, kEnvGuideRows, hs.envGuideRows.data());
            hs.envGuideCols.resize((size_t)h * (kEnvGuideCols + 1));
            for (int y = 0; y < h; ++y) guide(hs.envCdfCols.data() + (size_t)y * (w + 1), (size_t)w + 1, kEnvGuideCols, hs.envGuideCols.data() + (size_t)y * (kEnvGuideCols + 1));
            /* guide RECORDS (kkd_types.h, DEnvMap): per bucket the bracket and the CDF entries the search and the sample-reuse
             * step read, cdf[first - 1 ... last + 1]; brackets that do not fit (low-probability stretches of a row, rarely
             * sampled) keep the two-table path */
            auto records = [](const float *cdf, const uint32_t *g, uint32_t G, uint32_t W, uint32_t *out) {
                for (uint32_t b = 0; b < G; ++b, out += W) {
                    const uint32_t first = g[b], n = g[b + 1] - g[b] + 1u;
                    const bool fits = n + 3u <= W && first < (1u << 24);
                    out[0] = fits ? (first | n << 24) : 0xFF000000u;
                    for (uint32_t k = 0; k + 1u < W; ++k) {
                        float v = 2.f;                      /* never below a sample: the device counts entries below it */
                        if (fits && k <= n + 1u) v = first == 0u && k == 0u ? 0.f : cdf[first - 1u + k];
                        std::memcpy(out + 1 + k, &v, 4);
                    }
                }
            };
            hs.envRecRows.assign((size_t)kEnvGuideRows * kEnvRecRowWords, 0u);
            if (kEnvRecRowWords) records(hs.envCdfRows.data(), hs.envGuideRows.data(), kEnvGuideRows, kEnvRecRowWords, hs.envRecRows.data());
            hs.envRecCols.assign((size_t)h * kEnvGuideCols * kEnvRecColWords, 0u);
            if (kEnvRecColWords)
                for (int y = 0; y < h; ++y)
                    records(hs.envCdfCols.data() + (size_t)y * (w + 1), hs.envGuideCols.data() + (size_t)y * (kEnvGuideCols + 1), kEnvGuideCols,
                            kEnvRecColWords, hs.envRecCols.data() + (size_t)y * kEnvGuideCols * kEnvRecColWords);
        }
        buildEnvPyramid(k.texels, w, h, hs);
        hs.env.normalization = 1.0f / (rowSum * (2 * kPi / w) * (kPi / h));
        hs.env.pixelSizeX = 2 * kPi / w; hs.env.pixelSizeY = kPi / h;
        for (int i = 0; i < 9; ++i) hs.env.toWorld[i] = k.to_world[i];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) hs.env.toLocal[3 * i + j] = k.to_world[3 * j + i];
        for (int i = 0; i < 3; ++i) hs.env.center[i] = k.bsphere_center[i];
        hs.env.radius = k.bsphere_radius;
        if (!(hs.env.radius > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "envmap: bounding sphere radius must be positive");
    }

    /* sensor (perspective.cpp:126-163) */
    {
        const kkd_sensor &s = sd->sensor;
        if (!s.film_width || !s.film_height || !s.crop_width || !s.crop_height) KKD_FAIL(KKD_ERR_INVALID, "film has zero size");
        if (s.crop_offset_x + s.crop_width > s.film_width || s.crop_offset_y + s.crop_height > s.film_height) KKD_FAIL(KKD_ERR_INVALID, "crop window exceeds the film");
        if (!(s.xfov_deg > 0.f && s.xfov_deg < 180.f)) KKD_FAIL(KKD_ERR_INVALID, "field of view must be in (0, 180)");
        if (!(s.near_clip > 0.f && s.far_clip > s.near_clip)) KKD_FAIL(KKD_ERR_INVALID, "bad clipping planes");
        const double aspect = (double)s.film_width / (double)s.film_height;
        const double rsx = (double)s.crop_width / s.film_width, rsy = (double)s.crop_height / s.film_height;
        const double rox = (double)s.crop_offset_x / s.film_width, roy = (double)s.crop_offset_y / s.film_height;
        const double n = s.near_clip, f = s.far_clip, recip = 1.0 / (f - n), cot = 1.0 / std::tan((double)s.xfov_deg * 0.5 * M_PI / 180.0);
        const double P[16] = { cot, 0, 0, 0, 0, cot, 0, 0, 0, 0, f * recip, -n * f * recip, 0, 0, 1, 0 };   /* transform.cpp:99-123 */
        const double T1[16] = { 1, 0, 0, -1, 0, 1, 0, -1 / aspect, 0, 0, 1, 0, 0, 0, 0, 1 };
        const double S1[16] = { -0.5, 0, 0, 0, 0, -0.5 * aspect, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        const double T2[16] = { 1, 0, 0, -rox, 0, 1, 0, -roy, 0, 0, 1, 0, 0, 0, 0, 1 };
        const double S2[16] = { 1 / rsx, 0, 0, 0, 0, 1 / rsy, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        double a[16], b[16], c2s[16], s2c[16];
        mat4mul(T1, P, a); mat4mul(S1, a, b); mat4mul(T2, b, a); mat4mul(S2, a, c2s);
        if (!mat4inv(c2s, s2c)) KKD_FAIL(KKD_ERR_INVALID, "singular camera projection");
        for (int i = 0; i < 16; ++i) hs.sensor.s2c[i] = (float)s2c[i];
        for (int i = 0; i < 12; ++i) hs.sensor.toWorld[i] = s.to_world[i];
        {   /* m_dx, m_dy (perspective.cpp:159-163): sampleToCamera applied to points, in single precision like the ray itself */
            const float *m = hs.sensor.s2c;
            auto point = [&](float x, float y, float out[3]) {
                const float cx = m[0] * x + m[1] * y + m[3], cy = m[4] * x + m[5] * y + m[7], cz = m[8] * x + m[9] * y + m[11], cw = m[12] * x + m[13] * y + m[15];
                if (cw == 1.0f) { out[0] = cx; out[1] = cy; out[2] = cz; } else { out[0] = cx / cw; out[1] = cy / cw; out[2] = cz / cw; }
            };
            float p0[3], px[3], py[3];
            point(0.f, 0.f, p0); point(1.0f / (float)s.crop_width, 0.f, px); point(0.f, 1.0f / (float)s.crop_height, py);
            for (int c = 0; c < 3; ++c) { hs.sensor.dxCam[c] = px[c] - p0[c]; hs.sensor.dyCam[c] = py[c] - p0[c]; }
        }
        hs.sensor.nearClip = s.near_clip; hs.sensor.farClip = s.far_clip;
        hs.sensor.invResX = 1.0f / (float)s.crop_width; hs.sensor.invResY = 1.0f / (float)s.crop_height;
        hs.sensor.cropW = (int)s.crop_width; hs.sensor.cropH = (int)s.crop_height; hs.sensor.filmHasAlpha = s.film_has_alpha != 0; hs.sensor.hqEdges = s.film_high_quality_edges != 0;
        if (s.rfilter_type < 0 || s.rfilter_type > KKD_RFILTER_LANCZOS) KKD_FAIL(KKD_ERR_INVALID, "unknown reconstruction filter");
        if (!(s.rfilter_radius > 0.f)) KKD_FAIL(KKD_ERR_INVALID, "filter radius must be positive");
        configureFilter(s.rfilter_type, s.rfilter_radius, s.rfilter_param, hs.filter);
        if (hs.filter.border > 6) KKD_FAIL(KKD_ERR_UNSUPPORTED, "reconstruction filter radius above 6.5 pixels");
    }

    /* acceleration structure */
    if (sd->n_tris > (uint64_t)kLeafStartMask) KKD_FAIL(KKD_ERR_INVALID, "too many triangles for the 28-bit leaf slot field");
    if (buildBvh && !hs.bvh.build(sd->positions, sd->indices, sd->n_tris, hs.shapeOfPrim.data(), err)) { if (errCode) *errCode = KKD_ERR_INVALID; return false; }
    return true;
}

} /* namespace kkd */

/* ---- film output (host only) ------------------------------------------------------------------------------------ */
#include <cstdio>
namespace kkd {
/* ---- halton sampler tables (host) -------------------------------------------------------------------------------------------
 * primeTable of libcore/qmc.cpp:23 is the first 1024 primes (2 ... 8161): sieved here.  Faure permutations by the dynamic
 * programme of samplers/faure.cpp:101-131 (Faure 1992) over all bases up to the largest prime, then those of the primes are
 * kept (faure.cpp:36-63). */
void haltonTables(std::vector<uint32_t> &words) {
    std::vector<uint32_t> primes;
    {
        const uint32_t limit = 8200;
        std::vector<char> composite(limit + 1, 0);
        for (uint32_t i = 2; i <= limit && primes.size() < kHaltonDims; ++i) {
            if (composite[i]) continue;
            primes.push_back(i);
            for (uint32_t j = i * i; j <= limit; j += i) composite[j] = 1;
        }
    }
    const uint32_t maxBase = primes.back();
    std::vector<std::vector<uint16_t>> perm(maxBase + 1);
    perm[1] = { 0 }; perm[2] = { 0, 1 };
    for (uint32_t b = 3; b <= maxBase; ++b) {
        perm[b].resize(b);
        if (b & 1u) {
            const uint32_t c = (b - 1) / 2;
            for (uint32_t i = 0; i < b; ++i) {
                if (i == c) perm[b][i] = (uint16_t)c;
                else { const uint32_t f = perm[b - 1][i - (i > c ? 1u : 0u)]; perm[b][i] = (uint16_t)(f + (f >= c ? 1u : 0u)); }
            }
        } else {
            const uint32_t c = b / 2;
            for (uint32_t i = 0; i < b; ++i) perm[b][i] = (uint16_t)(i < c ? 2u * perm[c][i] : 2u * perm[c][i - c] + 1u);
        }
    }
    size_t total = 0; for (uint32_t pr : primes) total += pr;
    words.assign(2 * kHaltonDims + (total + 1) / 2, 0u);
    uint16_t *dst = reinterpret_cast<uint16_t *>(words.data() + 2 * kHaltonDims);
    size_t ofs = 0;
    for (uint32_t d = 0; d < kHaltonDims; ++d) {
        words[d] = primes[d]; words[kHaltonDims + d] = (uint32_t)ofs;
        std::memcpy(dst + ofs, perm[primes[d]].data(), primes[d] * sizeof(uint16_t));
        ofs += primes[d];
    }
}

namespace {
void extendedGcd(uint64_t a, uint64_t b, int64_t &x, int64_t &y) {          /* halton.cpp:217-227 */
    if (b == 0) { x = 1; y = 0; return; }
    const int64_t d = (int64_t)(a / b); int64_t x_, y_;
    extendedGcd(b, a % b, x_, y_);
    x = y_; y = x_ - d * y_;
}
uint64_t multiplicativeInverse(int64_t a, int64_t n) {                       /* halton.cpp:234-238, math::modulo */
    int64_t x, y; extendedGcd((uint64_t)a, (uint64_t)n, x, y);
    int64_t r = x % n; if (r < 0) r += n;
    return (uint64_t)r;
}
}
void haltonSetup(int resX, int resY, uint32_t nArrays, bool faure, const uint32_t *tables, HaltonSetup &h) {
    const int res[2] = { resX, resY }; uint64_t stride = 1;
    for (int i = 0; i < 2; ++i) {
        const int prime = i ? 3 : 2; int value = 1, exp = 0;
        while (value < std::min(res[i], 128)) { value *= prime; ++exp; }    /* MAX_RESOLUTION, halton.cpp:29, 252-255 */
        h.pow[i] = (uint32_t)value; h.exp[i] = (uint32_t)exp; stride *= (uint64_t)value;
    }
    h.stride = (uint32_t)stride;
    const uint64_t inv[2] = { multiplicativeInverse(h.pow[1], h.pow[0]), multiplicativeInverse(h.pow[0], h.pow[1]) };
    for (int i = 0; i < 2; ++i) h.mul[i] = (stride / h.pow[i]) * inv[i];
    h.arrayEnd = 5u + 2u * nArrays;                                          /* halton.cpp:276-278 */
    h.tables = tables; h.faure = faure ? 1u : 0u; h.hammersley = 0; h.count = 0; h.factor = 0.f;
}
/* ---- sobol sampler tables (samplers/sobolseq.{h,cpp}) ---- */
/* The 1024 x 52 generator matrices, top 32 bits of every 52-bit column, from Joe and Kuo's direction numbers: column k (1-based)
 * of a dimension is m_k << (52 - k) with m_1 ... m_s given and m_k = 2 a_1 m_{k-1} ^ 4 a_2 m_{k-2} ^ ... ^ 2^s m_{k-s} ^ m_{k-s}
 * beyond (the recurrence of the dimension's primitive polynomial x^s + a_1 x^(s-1) + ... + 1); dimension 0 is the identity
 * (van der Corput).  Checked against the CRC-32 of the reference's table (kSobolMatrixCrc). */
const std::vector<uint32_t> &sobolMatrices() {
    static const std::vector<uint32_t> table = [] {
        std::vector<uint32_t> t((size_t)kSobolDims * kSobolSize, 0u);
        for (uint32_t k = 0; k < 32; ++k) t[k] = 1u << (31 - k);
        const uint32_t *p = kJoeKuo;
        for (uint32_t d = 1; d < kSobolDims; ++d) {
            const uint32_t s = p[0], a = p[1];
            uint64_t m[kSobolSize];
            for (uint32_t k = 0; k < s; ++k) m[k] = p[2 + k];
            p += 2 + s;
            for (uint32_t k = s; k < kSobolSize; ++k) {
                uint64_t x = m[k - s] ^ (m[k - s] << s);
                for (uint32_t i = 1; i < s; ++i) if ((a >> (s - 1 - i)) & 1u) x ^= m[k - i] << i;
                m[k] = x;
            }
            for (uint32_t k = 0; k < kSobolSize; ++k) t[(size_t)d * kSobolSize + k] = (uint32_t)((m[k] << (kSobolSize - 1 - k)) >> 20);
        }
        return t;
    }();
    return table;
}

uint32_t crc32Words(const uint32_t *w, size_t n) {      /* CRC-32 (IEEE 802.3, as zlib) of the words' little-endian bytes */
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i)
        for (int b = 0; b < 4; ++b) {
            c ^= (w[i] >> (8 * b)) & 0xffu;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        }
    return ~c;
}

/* Tables of one render: the matrices, then for resolution 2^m the two 52-entry 64-bit tables sobolLookUp reads (kkd_math.h):
 * [0] what index bit 2m + c contributes to T = (top m bits of coordinate 0) << m | (top m bits of coordinate 1), [1] the
 * low index bits (below 2m) that make exactly bit j of T: the columns of the inverse, over GF(2), of the 2m x 2m matrix whose
 * column i is the contribution of index bit i.  It is invertible because the first two dimensions form a (0, 2)-sequence:
 * every 2^m x 2^m cell holds exactly one of any 2^(2m) consecutive points. */
bool sobolTables(uint32_t m, std::vector<uint32_t> &words) {
    const std::vector<uint32_t> &mat = sobolMatrices();
    words.assign(mat.begin(), mat.end());
    words.resize(mat.size() + 4 * kSobolSize, 0u);
    if (m == 0 || 2 * m > 40) return m == 0;
    auto contribution = [&](uint32_t bit) -> uint64_t {
        if (bit >= kSobolSize) return 0;
        return ((uint64_t)(mat[bit] >> (32 - m)) << m) | (uint64_t)(mat[kSobolSize + bit] >> (32 - m));
    };
    uint64_t *high = reinterpret_cast<uint64_t *>(words.data() + mat.size()), *inv = high + kSobolSize;
    for (uint32_t c = 0; c < kSobolSize; ++c) high[c] = contribution(2 * m + c);
    /* Gauss-Jordan on [A | I]: row i holds (column i of A, the combination of index bits that produces it) */
    const uint32_t n = 2 * m;
    std::vector<uint64_t> val(n), comb(n);
    for (uint32_t i = 0; i < n; ++i) { val[i] = contribution(i); comb[i] = 1ull << i; }
    for (uint32_t j = 0; j < n; ++j) {          /* make val[j] == 1 << j */
        uint32_t piv = j;
        while (piv < n && !((val[piv] >> j) & 1ull)) ++piv;
        if (piv == n) return false;
        std::swap(val[piv], val[j]); std::swap(comb[piv], comb[j]);
        for (uint32_t i = 0; i < n; ++i) if (i != j && ((val[i] >> j) & 1ull)) { val[i] ^= val[j]; comb[i] ^= comb[j]; }
    }
    for (uint32_t j = 0; j < n; ++j) inv[j] = comb[j];
    return true;
}

/* SobolSampler::setFilmResolution(res, blocked = true) + generate() (sobol.cpp:146-157, 173-179) */
void sobolSetup(int resX, int resY, uint32_t nArrays, const uint32_t *tables, SobolSetup &s) {
    uint32_t v = (uint32_t)std::max(1, std::max(resX, resY));                 /* math::roundToPowerOfTwo */
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    s.m = 0; while ((1u << s.m) < v) ++s.m;
    s.resolution = (float)v; s.arrayEnd = 5u + 2u * nArrays; s.pad = 0; s.tables = tables;
}

void hammersleySetup(int resX, int resY, uint32_t sampleCount, bool faure, const uint32_t *tables, HaltonSetup &h) {
    const int res[2] = { resX, resY };
    for (int i = 0; i < 2; ++i) {
        uint32_t v = (uint32_t)res[i];                                        /* math::roundToPowerOfTwo */
        v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
        h.pow[i] = std::min(128u, std::max(1u, v)); h.exp[i] = 0;            /* MAX_RESOLUTION, hammersley.cpp:29 */
        while ((1u << h.exp[i]) < h.pow[i]) ++h.exp[i];
    }
    h.stride = h.pow[1]; h.mul[0] = h.mul[1] = 0; h.arrayEnd = 5u;
    h.count = sampleCount;
    h.factor = (float)1.0f / (float)((size_t)sampleCount * (size_t)h.pow[0] * (size_t)h.pow[1]);   /* hammersley.cpp:191-192 */
    h.tables = tables; h.faure = faure ? 1u : 0u; h.hammersley = 1u;
}
} /* namespace kkd */

extern "C" {

/* Bitmap::writePFM, src/libcore/bitmap.cpp:3804-3843 */
int kkd_write_pfm(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format) {
    if (!path || (!pixels && (uint64_t)width * height != 0)) return KKD_ERR_INVALID;
    int ch;
    switch (pixel_format) {
        case KKD_PIXEL_LUMINANCE: ch = 1; break;
        case KKD_PIXEL_RGB: ch = 3; break;
        case KKD_PIXEL_RGBA: ch = 4; break;
        default: return KKD_ERR_INVALID;   /* "pixel format must be ERGB, ERGBA, ELuminance" */
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) return KKD_ERR_INVALID;
    const uint16_t one = 1; const bool little = *reinterpret_cast<const uint8_t *>(&one) == 1;
    std::fprintf(f, "P%c\n%u %u\n%s\n", ch == 1 ? 'f' : 'F', width, height, little ? "-1" : "1");
    bool ok = true;
    const size_t scan = (size_t)width * ch;
    std::vector<float> tmp(ch == 4 ? (size_t)width * 3 : 0);
    for (uint32_t y = 0; y < height && ok; ++y) {
        const float *src = pixels + scan * (height - 1 - y);
        if (ch != 4) ok = std::fwrite(src, sizeof(float), scan, f) == scan;
        else {      /* the alpha channel is stripped while saving */
            for (uint32_t x = 0; x < width; ++x) { tmp[3 * x] = src[4 * x]; tmp[3 * x + 1] = src[4 * x + 1]; tmp[3 * x + 2] = src[4 * x + 2]; }
            ok = std::fwrite(tmp.data(), sizeof(float), tmp.size(), f) == tmp.size();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KKD_OK : KKD_ERR_INVALID;
}

/* Bitmap::writeRGBE (src/libcore/bitmap.cpp:3679-3736) with RGBE_FromFloat (:3492-3508) and RGBE_WriteBytes_RLE (:3524-3572) */
namespace {
inline void rgbeFromFloat(const float *data, uint8_t rgbe[4]) {
    float mx = std::max(std::max(data[0], data[1]), data[2]);
    if (mx < 1e-32f) { rgbe[0] = rgbe[1] = rgbe[2] = rgbe[3] = 0; return; }
    int e;
    mx = std::frexp(mx, &e) * 256.0f / mx;
    rgbe[0] = (uint8_t)(data[0] * mx); rgbe[1] = (uint8_t)(data[1] * mx); rgbe[2] = (uint8_t)(data[2] * mx);
    rgbe[3] = (uint8_t)(e + 128);
}
inline void rgbeWriteRle(std::vector<uint8_t> &out, const uint8_t *data, int numbytes) {
    int cur = 0;
    while (cur < numbytes) {
        int begRun = cur, runCount = 0, oldRunCount = 0;
        while (runCount < 4 && begRun < numbytes) {                       /* the next run of at least 4, if there is one */
            begRun += runCount; oldRunCount = runCount; runCount = 1;
            while (begRun + runCount < numbytes && runCount < 127 && data[begRun] == data[begRun + runCount]) runCount++;
        }
        if (oldRunCount > 1 && oldRunCount == begRun - cur) {             /* a short run right before it */
            out.push_back((uint8_t)(128 + oldRunCount)); out.push_back(data[cur]);
            cur = begRun;
        }
        while (cur < begRun) {                                            /* literal bytes up to the run */
            const int n = std::min(begRun - cur, 128);
            out.push_back((uint8_t)n); out.insert(out.end(), data + cur, data + cur + n);
            cur += n;
        }
        if (runCount >= 4) { out.push_back((uint8_t)(128 + runCount)); out.push_back(data[begRun]); cur += runCount; }
    }
}
}
int kkd_write_rgbe(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format) {
    if (!path || (!pixels && (uint64_t)width * height != 0)) return KKD_ERR_INVALID;
    if (pixel_format != KKD_PIXEL_RGB && pixel_format != KKD_PIXEL_RGBA) return KKD_ERR_INVALID;   /* "pixel format must be ERGB or ERGBA" */
    const int ch = pixel_format == KKD_PIXEL_RGB ? 3 : 4;
    FILE *f = std::fopen(path, "wb");
    if (!f) return KKD_ERR_INVALID;
    std::vector<uint8_t> out;
    const std::string head = "#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(height) + " +X " + std::to_string(width) + "\n";
    out.insert(out.end(), head.begin(), head.end());
    const float *data = pixels;
    if (width < 8 || width > 0x7fff) {                                    /* run-length encoding is not allowed: flat pixels */
        for (size_t i = 0; i < (size_t)width * height; ++i, data += ch) { uint8_t q[4]; rgbeFromFloat(data, q); out.insert(out.end(), q, q + 4); }
    } else {
        std::vector<uint8_t> buffer((size_t)4 * width);
        for (uint32_t y = 0; y < height; ++y) {
            const uint8_t mark[4] = { 2, 2, (uint8_t)(width >> 8), (uint8_t)(width & 0xFF) };
            out.insert(out.end(), mark, mark + 4);
            for (uint32_t x = 0; x < width; ++x, data += ch) {
                uint8_t q[4]; rgbeFromFloat(data, q);
                buffer[x] = q[0]; buffer[width + x] = q[1]; buffer[2 * (size_t)width + x] = q[2]; buffer[3 * (size_t)width + x] = q[3];
            }
            for (int i = 0; i < 4; ++i) rgbeWriteRle(out, &buffer[(size_t)i * width], (int)width);
        }
    }
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KKD_OK : KKD_ERR_INVALID;
}

/* Bitmap::writeOpenEXR (src/libcore/bitmap.cpp:3171-3318) without the OpenEXR library: the published layout of a single-part
 * scanline file -- magic, version 2, attributes in the order of Imf::Header's map (by name), line offset table, chunks of 1 (no
 * compression) or 16 (ZIP) scanlines, each scanline channel by channel in alphabetical order; ZIP = byte de-interleave,
 * delta predictor, deflate, kept only when smaller. */
namespace {
struct ExrBuf {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void i32(int32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)((uint32_t)v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    void f32(float v) { uint32_t u; std::memcpy(&u, &v, 4); i32((int32_t)u); }
    void str0(const char *s) { while (*s) b.push_back((uint8_t)*s++); b.push_back(0); }
    void attr(const char *name, const char *type, const ExrBuf &value) { str0(name); str0(type); i32((int32_t)value.b.size()); b.insert(b.end(), value.b.begin(), value.b.end()); }
};
}
int kkd_write_exr(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format,
                  int32_t component_format, int32_t compression, const char *generated_by) {
    if (!path || !pixels || !width || !height || width > 0x3fffffffu || height > 0x3fffffffu) return KKD_ERR_INVALID;
    if (component_format != KKD_COMPONENT_FLOAT16 && component_format != KKD_COMPONENT_FLOAT32) return KKD_ERR_INVALID;
    if (compression != KKD_EXR_NO_COMPRESSION && compression != KKD_EXR_ZIP_COMPRESSION) return KKD_ERR_UNSUPPORTED;
    /* channel names in file (= alphabetical) order with their position in the interleaved source pixel */
    struct Ch { const char *name; int src; };
    std::vector<Ch> chans; int nSrc;
    switch (pixel_format) {
        case KKD_PIXEL_LUMINANCE: chans = { { "Y", 0 } }; nSrc = 1; break;
        case KKD_PIXEL_LUMINANCE_ALPHA: chans = { { "A", 1 }, { "Y", 0 } }; nSrc = 2; break;
        case KKD_PIXEL_RGB: case KKD_PIXEL_XYZ: chans = { { "B", 2 }, { "G", 1 }, { "R", 0 } }; nSrc = 3; break;
        case KKD_PIXEL_RGBA: case KKD_PIXEL_XYZA: chans = { { "A", 3 }, { "B", 2 }, { "G", 1 }, { "R", 0 } }; nSrc = 4; break;
        default: return KKD_ERR_INVALID;
    }
    const bool half = component_format == KKD_COMPONENT_FLOAT16;
    const size_t comp = half ? 2 : 4, lineBytes = (size_t)width * chans.size() * comp;
    ExrBuf f;
    f.u8(0x76); f.u8(0x2f); f.u8(0x31); f.u8(0x01);            /* magic 20000630 */
    f.i32(2);                                                  /* version 2, no flags: single-part scanline */
    { ExrBuf v; for (const Ch &c : chans) { v.str0(c.name); v.i32(half ? 1 : 2); v.u8(0); v.u8(0); v.u8(0); v.u8(0); v.i32(1); v.i32(1); } v.u8(0); f.attr("channels", "chlist", v); }
    if (pixel_format >= KKD_PIXEL_RGB) {
        ExrBuf v;
        const bool xyz = pixel_format == KKD_PIXEL_XYZ || pixel_format == KKD_PIXEL_XYZA;
        const float rec709[8] = { 0.64f, 0.33f, 0.30f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f };        /* Imf::Chromaticities() */
        const float cie[8] = { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f / 3.0f, 1.0f / 3.0f };         /* bitmap.cpp:3224-3228 */
        for (int i = 0; i < 8; ++i) v.f32(xyz ? cie[i] : rec709[i]);
        f.attr("chromaticities", "chromaticities", v);
    }
    { ExrBuf v; v.u8((uint8_t)compression); f.attr("compression", "compression", v); }
    { ExrBuf v; v.i32(0); v.i32(0); v.i32((int32_t)width - 1); v.i32((int32_t)height - 1); f.attr("dataWindow", "box2i", v); f.attr("displayWindow", "box2i", v); }
    { ExrBuf v; const char *g = generated_by ? generated_by : "kkdirect"; while (*g) v.u8((uint8_t)*g++); f.attr("generatedBy", "string", v); }
    { ExrBuf v; v.u8(0); f.attr("lineOrder", "lineOrder", v); }
    { ExrBuf v; v.f32(1.0f); f.attr("pixelAspectRatio", "float", v); }
    { ExrBuf v; v.f32(0.0f); v.f32(0.0f); f.attr("screenWindowCenter", "v2f", v); }
    { ExrBuf v; v.f32(1.0f); f.attr("screenWindowWidth", "float", v); }
    f.u8(0);                                                   /* end of the header */
    const uint32_t linesPerChunk = compression == KKD_EXR_ZIP_COMPRESSION ? 16u : 1u, nChunks = (height + linesPerChunk - 1) / linesPerChunk;
    const size_t tablePos = f.b.size();
    for (uint32_t c = 0; c < nChunks; ++c) f.u64(0);
    std::vector<uint8_t> raw, tmp;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const uint32_t y0 = c * linesPerChunk, y1 = std::min(height, y0 + linesPerChunk);
        raw.clear(); raw.reserve((size_t)(y1 - y0) * lineBytes);
        for (uint32_t y = y0; y < y1; ++y)
            for (const Ch &ch : chans) {
                const float *src = pixels + (size_t)y * width * nSrc + ch.src;
                for (uint32_t x = 0; x < width; ++x, src += nSrc) {
                    if (half) { const uint16_t h = kkd::floatToHalf(*src); raw.push_back((uint8_t)h); raw.push_back((uint8_t)(h >> 8)); }
                    else { uint32_t u; std::memcpy(&u, src, 4); for (int i = 0; i < 4; ++i) raw.push_back((uint8_t)(u >> (8 * i))); }
                }
            }
        const uint8_t *data = raw.data(); size_t n = raw.size();
        std::vector<uint8_t> packed;
        if (compression == KKD_EXR_ZIP_COMPRESSION) {
            tmp.resize(n);
            {   /* even bytes to the first half, odd bytes to the second */
                size_t t1 = 0, t2 = (n + 1) / 2;
                for (size_t i = 0; i < n; ++i) { if ((i & 1) == 0) tmp[t1++] = raw[i]; else tmp[t2++] = raw[i]; }
            }
            {   /* delta predictor */
                int p = tmp[0];
                for (size_t i = 1; i < n; ++i) { const int d = (int)tmp[i] - p + (128 + 256); p = tmp[i]; tmp[i] = (uint8_t)d; }
            }
            uLongf outLen = compressBound((uLong)n);
            packed.resize(outLen);
            if (compress(packed.data(), &outLen, tmp.data(), (uLong)n) != Z_OK) return KKD_ERR_INVALID;
            if (outLen < n) { packed.resize(outLen); data = packed.data(); n = outLen; }      /* else: the chunk is stored as it is */
        }
        const uint64_t pos = f.b.size();
        for (int i = 0; i < 8; ++i) f.b[tablePos + 8 * (size_t)c + i] = (uint8_t)(pos >> (8 * i));
        f.i32((int32_t)y0); f.i32((int32_t)n);
        f.b.insert(f.b.end(), data, data + n);
    }
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return KKD_ERR_INVALID;
    bool ok = std::fwrite(f.b.data(), 1, f.b.size(), fp) == f.b.size();
    ok = (std::fclose(fp) == 0) && ok;
    return ok ? KKD_OK : KKD_ERR_INVALID;
}

/* MFilm::develop, numpy branch (src/films/mfilm.cpp:336-352): cnpy::npy_save of a float array, shape (H, W[, C]) */
int kkd_write_npy(const char *path, const float *pixels, uint32_t width, uint32_t height, uint32_t channels) {
    if (!path || channels == 0 || (!pixels && (uint64_t)width * height != 0)) return KKD_ERR_INVALID;
    FILE *f = std::fopen(path, "wb");
    if (!f) return KKD_ERR_INVALID;
    char dict[160];
    if (channels == 1) std::snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (%u, %u), }", height, width);
    else std::snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (%u, %u, %u), }", height, width, channels);
    std::string hdr(dict);
    /* magic (6) + version (2) + header length (2) + dict, padded with spaces to a multiple of 16, ending in a newline */
    while ((10 + hdr.size() + 1) % 16 != 0) hdr.push_back(' ');
    hdr.push_back('\n');
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    const unsigned char len[2] = { (unsigned char)(hdr.size() & 0xff), (unsigned char)(hdr.size() >> 8) };
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(len, 1, 2, f) == 2 && std::fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
    const size_t n = (size_t)width * height * channels;
    ok = ok && std::fwrite(pixels, sizeof(float), n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KKD_OK : KKD_ERR_INVALID;
}

/* MFilm::develop, matlab / mathematica branch (src/films/mfilm.cpp:269-334).  `os << float` under setprecision(d) with the
 * default floatfield is printf's %.{d}g of the value (precision 0 is read as 1 by both). */
int kkd_write_mfilm_text(const char *path, const float *pixels, uint32_t width, uint32_t height, uint32_t channels,
                         int32_t file_format, int32_t digits, const char *variable) {
    if (!path || channels == 0 || width == 0 || height == 0 || !pixels || digits < 0 || digits > 64
        || (file_format != KKD_MFILM_MATLAB && file_format != KKD_MFILM_MATHEMATICA)) return KKD_ERR_INVALID;
    FILE *f = std::fopen(path, "w");
    if (!f) return KKD_ERR_INVALID;
    const std::string var = variable ? variable : "data";
    const int prec = digits ? digits : 4;
    const bool matlab = file_format == KKD_MFILM_MATLAB;
    std::string out; out.reserve((size_t)1 << 20);
    bool ok = true;
    char num[96];
    for (uint32_t ch = 0; ch < channels; ++ch) {
        if (matlab) {
            if (ch == 0) out += var + " = [";
            else out += "\n" + var + "(:, :, " + std::to_string(ch + 1) + ") = [";
        } else if (ch == 0) out += var + (channels == 1 ? " = {{" : " = Transpose[{{{");
        const float *ptr = pixels + ch;
        for (uint32_t y = 0; y < height; ++y) {
            for (uint32_t x = 0; x < width; ++x, ptr += channels) {
                std::snprintf(num, sizeof(num), "%.*g", prec, (double)*ptr);
                if (matlab) out += num;
                else {
                    std::string str(num);                          /* boost::replace_first(str, "e", "*^") */
                    const size_t e = str.find('e');
                    if (e != std::string::npos) str.replace(e, 1, "*^");
                    out += str;
                }
                if (x + 1 < width) out += ", ";
                else if (matlab) out += (y + 1 < height) ? ";\n\t" : "];\n";
                else if (y + 1 < height) out += "},\n\t{";
                else if (ch + 1 == channels) out += (channels == 1) ? "}};\n" : "}}}, {3,1,2}];\n";
                else out += "}},\n\n\t{{";
            }
            if (out.size() >= ((size_t)1 << 20)) { ok = ok && std::fwrite(out.data(), 1, out.size(), f) == out.size(); out.clear(); }
        }
    }
    ok = ok && std::fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KKD_OK : KKD_ERR_INVALID;
}

} /* extern "C" */

/* ---- BSDF parameter extraction for the Mitsuba-side adapter (host only) -------------------------------------------
 * The adapter cannot reach into a BSDF plugin, but it can serialise it: InstanceManager::serialize(stream, bsdf)
 * (src/libcore/serialization.cpp:73-87) writes [uint32 id][class name, NUL-terminated][body] for a new object, the id
 * alone for one already written, 0 for NULL.  Bodies (little endian, Float = float32, Spectrum = 3 floats):
 *   BSDF base                 bool ensureEnergyConservation                       src/librender/bsdf.cpp:43-46
 *   SmoothDiffuse             base, Texture reflectance                           src/bsdfs/diffuse.cpp:161-165
 *   RoughConductor            base, uint type, bool sampleVisible, Texture alphaU, alphaV, specularReflectance,
 *                             Spectrum eta, Spectrum k                            src/bsdfs/roughconductor.cpp:216-226
 *   RoughPlastic              base, uint type, bool sampleVisible, Texture specularReflectance, diffuseReflectance,
 *                             alpha, float eta, bool nonlinear                    src/bsdfs/roughplastic.cpp:243-253
 *   SmoothDielectric          base, float eta, Texture specularReflectance, specularTransmittance
 *                                                                                 src/bsdfs/dielectric.cpp:176-182
 *   TwoSidedBRDF              base, BSDF nested[0], BSDF nested[1]                src/bsdfs/twosided.cpp:77-82
 *   ConstantSpectrumTexture   Spectrum value;  ConstantFloatTexture  float value  src/libhw/basicshader.cpp:29-49
 * Any other class (bitmap textures, other BSDFs) is rejected: the device path takes constant parameters only. */
namespace {
struct BsdfStream {
    const uint8_t *p, *end; std::string err;
    std::map<uint32_t, int> bsdfOfId;                 /* serialisation id -> index in out[] */
    std::map<uint32_t, std::array<float, 3>> texOfId; /* constant textures already seen     */
    kkd_material *out; uint32_t cap, n = 0;
    bool take(void *dst, size_t k) { if ((size_t)(end - p) < k) { err = "BSDF stream: truncated"; return false; } std::memcpy(dst, p, k); p += k; return true; }
    bool u32(uint32_t &v) { return take(&v, 4); }
    bool f32(float &v) { return take(&v, 4); }
    bool b8(bool &v) { uint8_t c; if (!take(&c, 1)) return false; v = c != 0; return true; }
    bool str(std::string &s) { s.clear(); while (p < end && *p) s.push_back((char)*p++); if (p >= end) { err = "BSDF stream: unterminated class name"; return false; } ++p; return true; }
    bool texture(float rgb[3]) {
        uint32_t id; if (!u32(id)) return false;
        if (id == 0) { err = "BSDF stream: missing texture"; return false; }
        auto it = texOfId.find(id);
        if (it != texOfId.end()) { for (int k = 0; k < 3; ++k) rgb[k] = it->second[k]; return true; }
        std::string cls; if (!str(cls)) return false;
        std::array<float, 3> v;
        if (cls == "ConstantSpectrumTexture") { if (!take(v.data(), 12)) return false; }
        else if (cls == "ConstantFloatTexture") { float f; if (!f32(f)) return false; v = { f, f, f }; }
        else { err = "BSDF stream: texture class \"" + cls + "\" is not constant"; return false; }
        texOfId[id] = v; for (int k = 0; k < 3; ++k) rgb[k] = v[k];
        return true;
    }
    /* returns the material index or -1 */
    int bsdf(bool allowNull) {
        uint32_t id; if (!u32(id)) return -1;
        if (id == 0) { if (!allowNull) err = "BSDF stream: missing BSDF"; return -1; }
        auto it = bsdfOfId.find(id);
        if (it != bsdfOfId.end()) return it->second;
        std::string cls; if (!str(cls)) return -1;
        if (n >= cap) { err = "BSDF stream: more materials than the output array holds"; return -1; }
        const int me = (int)n++;
        bsdfOfId[id] = me;
        kkd_material m; std::memset(&m, 0, sizeof(m));
        m.sample_visible = 1; m.nested[0] = m.nested[1] = -1;
        for (int k = 0; k < 3; ++k) { m.specular_reflectance[k] = 1.f; m.specular_transmittance[k] = 1.f; }
        bool ec;
        if (!b8(ec)) return -1;                                   /* BSDF::serialize */
        uint32_t type; bool sv; float a[3];
        if (cls == "SmoothDiffuse") {
            m.type = KKD_BSDF_DIFFUSE;
            if (!texture(m.reflectance)) return -1;
        } else if (cls == "RoughConductor") {
            m.type = KKD_BSDF_ROUGHCONDUCTOR;
            if (!u32(type) || !b8(sv)) return -1;
            m.distribution = (int32_t)type; m.sample_visible = sv;
            if (!texture(a)) return -1;
            m.alpha_u = a[0];
            if (!texture(a)) return -1;
            m.alpha_v = a[0];
            if (!texture(m.specular_reflectance) || !take(m.cond_eta, 12) || !take(m.cond_k, 12)) return -1;
        } else if (cls == "RoughPlastic") {
            m.type = KKD_BSDF_ROUGHPLASTIC;
            if (!u32(type) || !b8(sv)) return -1;
            m.distribution = (int32_t)type; m.sample_visible = sv;
            if (!texture(m.specular_reflectance) || !texture(m.reflectance) || !texture(a)) return -1;
            m.alpha_u = m.alpha_v = a[0];
            bool nl; if (!f32(m.eta) || !b8(nl)) return -1;
            m.nonlinear = nl;
        } else if (cls == "SmoothDielectric") {
            m.type = KKD_BSDF_DIELECTRIC;
            if (!f32(m.eta) || !texture(m.specular_reflectance) || !texture(m.specular_transmittance)) return -1;
        } else if (cls == "TwoSidedBRDF") {
            m.type = KKD_BSDF_TWOSIDED;
            out[me] = m;                                          /* reserve the slot before the children take theirs */
            const int n0 = bsdf(false); if (n0 < 0) return -1;
            const int n1 = bsdf(true); if (n1 < 0 && !err.empty()) return -1;
            m.nested[0] = n0; m.nested[1] = n1;
        } else { err = "BSDF stream: BSDF class \"" + cls + "\" is not on the kk direct path"; return -1; }
        if (m.distribution < 0 || m.distribution > KKD_DISTR_PHONG) { err = "BSDF stream: unknown microfacet distribution"; return -1; }
        out[me] = m;
        return me;
    }
};
} /* namespace */

extern "C" int kkd_parse_bsdf_stream(const void *bytes, uint64_t n_bytes, kkd_material *out, uint32_t capacity, uint32_t *count) {
    if (!bytes || !out || !count) { kkd::g_globalError = "kkd_parse_bsdf_stream: null argument"; return KKD_ERR_INVALID; }
    BsdfStream s{ (const uint8_t *)bytes, (const uint8_t *)bytes + n_bytes, std::string(), {}, {}, out, capacity };
    *count = 0;
    /* the stream may hold several top-level BSDFs in a row (one InstanceManager for the whole scene): parse to the end */
    while (s.p < s.end) {
        if (s.bsdf(false) < 0) { kkd::g_globalError = s.err.empty() ? "BSDF stream: malformed" : s.err; return KKD_ERR_INVALID; }
    }
    *count = s.n;
    return KKD_OK;
}
