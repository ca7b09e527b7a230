/*
 * ORACLE (test infrastructure only): CPU restatement of the BSDF models on the kk direct path.
 *
 * Follows (relative to /root/reference/src):
 *   bsdfs/microfacet.h:67-93,191-234,240-275,287-395,421-464,475-520,539-562,570-701
 *   bsdfs/diffuse.cpp:93-150
 *   bsdfs/roughconductor.cpp:228-415
 *   bsdfs/roughplastic.cpp:255-501   (+ bsdfs/rtrans.h:183-280,288-384 table reduction)
 *   bsdfs/dielectric.cpp:184-333
 *   bsdfs/twosided.cpp:84-183
 *   librender/bsdf.cpp:105-130       ensureEnergyConservation
 *   libcore/spline.cpp:23-60         evalCubicInterp1D
 *   utils/rdielprec.cpp:40-112 + bsdfs/roughdielectric.cpp:424-511: definition of the rough
 *     transmittance that data/microfacet/<distr>.dat tabulates.  That data file is ABSENT from the
 *     reference tree, so the 100-entry slice rtrans.h would produce by setEta()/setAlpha() is
 *     regenerated here by direct numerical integration at the exact (eta, alpha): parity against
 *     the shipped table is UNPINNED (tests/test_rtrans.cpp:154-190 allows 1e-3 abs. error).
 */
#pragma once
#include "ora_math.h"
#include <vector>
#include <map>
#include <tuple>
#include <mutex>

namespace ora {

/* BSDF::EBSDFType bits used on this path (bsdf.h absent; names follow the reference) */
enum {
    EDiffuseReflection = 0x0004, EGlossyReflection = 0x0010, EDeltaReflection = 0x0040,
    EDeltaTransmission = 0x0080, EFrontSide = 0x1000, EBackSide = 0x2000,
    EDelta = EDeltaReflection | EDeltaTransmission,
    ETransmission = EDeltaTransmission,
    ESmooth = EDiffuseReflection | 0x0008 /* EDiffuseTransmission */ | EGlossyReflection | 0x0020 /* EGlossyTransmission */
};

/* ---- bsdfs/microfacet.h ------------------------------------------------------------------- */
template <typename T> struct MicrofacetT {
    int type; T alphaU, alphaV; bool sampleVisibleFlag; T expU, expV;
    MicrofacetT(int type_, T au, T av, bool sv) : type(type_), alphaU(au), alphaV(av), sampleVisibleFlag(sv), expU(0), expV(0) {
        alphaU = std::max(alphaU, (T)1e-4f); alphaV = std::max(alphaV, (T)1e-4f);
        if (type == 2) { sampleVisibleFlag = false; computePhongExponent(); }
    }
    void computePhongExponent() {
        expU = std::max((T)2 / (alphaU * alphaU) - (T)2, (T)0);
        expV = std::max((T)2 / (alphaV * alphaV) - (T)2, (T)0);
    }
    bool isIsotropic() const { return alphaU == alphaV; }
    T projectRoughness(T vx, T vy, T vz) const {
        T invSinTheta2 = 1 / (1 - vz * vz);
        if (isIsotropic() || invSinTheta2 <= 0) return alphaU;
        T cosPhi2 = vx * vx * invSinTheta2, sinPhi2 = vy * vy * invSinTheta2;
        return std::sqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }
    T interpolatePhongExponent(T vx, T vy, T vz) const {
        T invSinTheta2 = 1 / (1 - vz * vz);
        if (isIsotropic() || invSinTheta2 <= 0) return expU;
        T cosPhi2 = vx * vx * invSinTheta2, sinPhi2 = vy * vy * invSinTheta2;
        return expU * cosPhi2 + expV * sinPhi2;
    }
    /* microfacet.h:191-234 */
    T eval(T mx, T my, T mz) const {
        if (mz <= 0) return 0;
        T cosTheta2 = mz * mz;
        T beckmannExponent = ((mx * mx) / (alphaU * alphaU) + (my * my) / (alphaV * alphaV)) / cosTheta2;
        T result;
        const T pi = (T)3.14159265358979323846;
        if (type == 0) {
            result = std::exp(-beckmannExponent) / (pi * alphaU * alphaV * cosTheta2 * cosTheta2);
        } else if (type == 1) {
            T root = ((T)1 + beckmannExponent) * cosTheta2;
            result = (T)1 / (pi * alphaU * alphaV * root * root);
        } else {
            T exponent = interpolatePhongExponent(mx, my, mz);
            result = std::sqrt((expU + 2) * (expV + 2)) * (T)0.15915494309189533577 * std::pow(mz, exponent);
        }
        if (result * mz < (T)1e-20f) result = 0;
        return result;
    }
    /* microfacet.h:475-512 */
    T smithG1(T vx, T vy, T vz, T mx, T my, T mz) const {
        if ((vx * mx + vy * my + vz * mz) * vz <= 0) return 0;
        T temp = 1 - vz * vz;
        T tanTheta = temp <= 0 ? (T)0 : std::abs(std::sqrt(temp) / vz);
        if (tanTheta == 0) return 1;
        T alpha = projectRoughness(vx, vy, vz);
        if (type == 0 || type == 2) {
            T a = (T)1 / (alpha * tanTheta);
            if (a >= (T)1.6f) return 1;
            T aSqr = a * a;
            return ((T)3.535f * a + (T)2.181f * aSqr) / ((T)1 + (T)2.276f * a + (T)2.577f * aSqr);
        } else {
            T root = alpha * tanTheta;
            return (T)2 / ((T)1 + std::sqrt((T)1 + root * root));
        }
    }
};

struct Microfacet : MicrofacetT<Float> {
    Microfacet(int t, Float au, Float av, bool sv) : MicrofacetT<Float>(t, au, av, sv) {}
    Float evalV(V3 m) const { return eval(m.x, m.y, m.z); }
    Float G1(V3 v, V3 m) const { return smithG1(v.x, v.y, v.z, m.x, m.y, m.z); }
    Float G(V3 wi, V3 wo, V3 m) const { return G1(wi, m) * G1(wo, m); }
    Float pdfAll(V3 m) const { return evalV(m) * m.z; }
    Float pdfVisible(V3 wi, V3 m) const { return G1(wi, m) * absDot(wi, m) * evalV(m) / std::fabs(wi.z); }
    Float pdf(V3 wi, V3 m) const { return sampleVisibleFlag ? pdfVisible(wi, m) : pdfAll(m); }

    void sampleFirstQuadrant(Float u1, Float &phi, Float &exponent) const {
        phi = std::atan(std::sqrt((expU + 2.0f) / (expV + 2.0f)) * std::tan(kPi * u1 * 0.5f));
        Float c = std::cos(phi), s = std::sin(phi);
        exponent = expU * c * c + expV * s * s;
    }
    /* microfacet.h:287-395 */
    V3 sampleAll(V2 sample, Float &pdf) const {
        Float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
        if (type == 0 || type == 1) {
            if (isIsotropic()) {
                Float a = (2.0f * kPi) * sample.y;
                sinPhiM = std::sin(a); cosPhiM = std::cos(a);
                alphaSqr = alphaU * alphaU;
            } else {
                Float phiM = std::atan(alphaV / alphaU * std::tan(kPi + 2 * kPi * sample.y)) + kPi * std::floor(2 * sample.y + 0.5f);
                sinPhiM = std::sin(phiM); cosPhiM = std::cos(phiM);
                Float cosSc = cosPhiM / alphaU, sinSc = sinPhiM / alphaV;
                alphaSqr = 1.0f / (cosSc * cosSc + sinSc * sinSc);
            }
            if (type == 0) {
                Float tanThetaMSqr = alphaSqr * -std::log(1.0f - sample.x);
                cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
                pdf = (1.0f - sample.x) / (kPi * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
            } else {
                Float tanThetaMSqr = alphaSqr * sample.x / (1.0f - sample.x);
                cosThetaM = 1.0f / std::sqrt(1.0f + tanThetaMSqr);
                Float temp = 1 + tanThetaMSqr / alphaSqr;
                pdf = kInvPi / (alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
            }
        } else {
            Float phiM, exponent;
            if (isIsotropic()) { phiM = (2.0f * kPi) * sample.y; exponent = expU; }
            else if (sample.y < 0.25f) { sampleFirstQuadrant(4 * sample.y, phiM, exponent); }
            else if (sample.y < 0.5f) { sampleFirstQuadrant(4 * (0.5f - sample.y), phiM, exponent); phiM = kPi - phiM; }
            else if (sample.y < 0.75f) { sampleFirstQuadrant(4 * (sample.y - 0.5f), phiM, exponent); phiM += kPi; }
            else { sampleFirstQuadrant(4 * (1 - sample.y), phiM, exponent); phiM = 2 * kPi - phiM; }
            sinPhiM = std::sin(phiM); cosPhiM = std::cos(phiM);
            cosThetaM = std::pow(sample.x, 1.0f / (exponent + 2.0f));
            pdf = std::sqrt((expU + 2.0f) * (expV + 2.0f)) * kInvTwoPi * std::pow(cosThetaM, exponent + 1.0f);
        }
        if (pdf < 1e-20f) pdf = 0;
        Float sinThetaM = std::sqrt(std::max(0.f, 1 - cosThetaM * cosThetaM));
        return V3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }
    /* microfacet.h:570-694 */
    V2 sampleVisible11(Float thetaI, V2 sample) const {
        const Float SQRT_PI_INV = 1 / std::sqrt(kPi);
        V2 slope;
        if (type == 0) {
            if (thetaI < 1e-4f) {
                Float r = std::sqrt(-std::log(1.0f - sample.x));
                Float a = 2 * kPi * sample.y;
                return V2{ r * std::cos(a), r * std::sin(a) };
            }
            Float tanThetaI = std::tan(thetaI), cotThetaI = 1 / tanThetaI;
            Float a = -1, c = erf_as(cotThetaI);
            Float sample_x = std::max(sample.x, 1e-6f);
            Float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            Float b = c - (1 + c) * std::pow(1 - sample_x, fit);
            Float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * std::exp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                Float invErf = erfinv(b);
                Float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * std::exp(-invErf * invErf)) - sample_x;
                Float derivative = normalization * (1 - invErf * tanThetaI);
                if (std::fabs(value) < 1e-5f) break;
                if (value > 0) c = b; else a = b;
                b -= value / derivative;
            }
            slope.x = erfinv(b);
            slope.y = erfinv(2.0f * std::max(sample.y, 1e-6f) - 1.0f);
        } else {
            if (thetaI < 1e-4f) {
                Float r = safe_sqrt(sample.x / (1 - sample.x));
                Float a = 2 * kPi * sample.y;
                return V2{ r * std::cos(a), r * std::sin(a) };
            }
            Float tanThetaI = std::tan(thetaI);
            Float a = 1 / tanThetaI;
            Float G1 = 2.0f / (1.0f + safe_sqrt(1.0f + 1.0f / (a * a)));
            Float A = 2.0f * sample.x / G1 - 1.0f;
            if (std::fabs(A) == 1) A -= signum(A) * kEpsilon;
            Float tmp = 1.0f / (A * A - 1.0f);
            Float B = tanThetaI;
            Float D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
            Float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
            slope.x = (A < 0.0f || slope_x_2 > 1.0f / tanThetaI) ? slope_x_1 : slope_x_2;
            Float S;
            if (sample.y > 0.5f) { S = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
            else { S = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
            Float z = (sample.y * (sample.y * (sample.y * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f) /
                      (sample.y * (sample.y * (sample.y * (sample.y * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
            slope.y = S * z * std::sqrt(1.0f + slope.x * slope.x);
        }
        return slope;
    }
    /* microfacet.h:421-459 */
    V3 sampleVisible(V3 _wi, V2 sample) const {
        V3 wi = normalize(V3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        Float theta = 0, phi = 0;
        if (wi.z < 0.99999f) { theta = std::acos(wi.z); phi = std::atan2(wi.y, wi.x); }
        Float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
        V2 slope = sampleVisible11(theta, sample);
        slope = V2{ cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y };
        slope.x *= alphaU; slope.y *= alphaV;
        Float normalization = 1.0f / std::sqrt(slope.x * slope.x + slope.y * slope.y + 1.0f);
        return V3(-slope.x * normalization, -slope.y * normalization, normalization);
    }
    /* microfacet.h:240-249 */
    V3 sample(V3 wi, V2 s, Float &pdf) const {
        if (sampleVisibleFlag) { V3 m = sampleVisible(wi, s); pdf = pdfVisible(wi, m); return m; }
        return sampleAll(s, pdf);
    }
};

/* ---- libcore/spline.cpp:23-60 ---------------------------------------------------------------- */
template <typename T> inline T evalCubicInterp1D(T x, const T *values, size_t size, T min, T max) {
    if (!(x >= min && x <= max)) return 0;
    T t = ((x - min) * (size - 1)) / (max - min);
    size_t k = std::max((size_t)0, std::min((size_t)t, size - 2));
    T f0 = values[k], f1 = values[k + 1], d0, d1;
    if (k > 0) d0 = (T)0.5 * (values[k + 1] - values[k - 1]); else d0 = values[k + 1] - values[k];
    if (k + 2 < size) d1 = (T)0.5 * (values[k + 2] - values[k]); else d1 = values[k + 1] - values[k];
    t = t - (T)k;
    T t2 = t * t, t3 = t2 * t;
    return (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
}

/* ---- rough transmittance (rtrans.h after setEta+setAlpha) regenerated by quadrature ------------ */
struct RoughTransTable {
    Float trans[100];   /* T(cosTheta), knots at cosTheta^(1/4) = i/99 (rdielprec.cpp:88-95) */
    Float diffTrans;    /* 2 int_0^1 mu T(mu) dmu  (rdielprec.cpp:58-62,103-106)              */
    /* rtrans.h:183-191,229 */
    Float eval(Float cosTheta) const {
        Float warped = std::pow(std::fabs(cosTheta), 0.25f);
        Float r = evalCubicInterp1D<Float>(warped, trans, 100, 0.0f, 1.0f);
        return std::min(1.0f, std::max(0.0f, r));
    }
    Float evalDiffuse() const { return std::min(1.0f, std::max(0.0f, diffTrans)); }
};

inline void gaussLegendre(int n, std::vector<double> &x, std::vector<double> &w) {
    x.resize(n); w.resize(n);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(3.14159265358979323846 * (i + 0.75) / (n + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 0; j < n; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j + 1) * z * p2 - j * p3) / (j + 1); }
            pp = n * (z * p1 - p2) / (z * z - 1);
            double z1 = z; z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-15) break;
        }
        x[i] = -z; x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2 / ((1 - z * z) * pp * pp);
    }
}

/* Transmittance through a rough dielectric boundary for direction cosTheta (wi = (sin,0,cos)),
 * = E[ BSDF::sample weight ] of roughdielectric restricted to ETransmission in EImportance mode
 * (rdielprec.cpp:40-56; roughdielectric.cpp:453-510 with sampleVisible=true):
 *   T = int (1-F(wi.m)) G1(wo,m) * [G1(wi,m) |wi.m| D(m) / cos(theta_i)] dm,  wo = refract(wi,m),
 * zero where total internal reflection occurs or wo ends on the wrong side. */
inline double roughTransmittanceAt(int distr, double alpha, double eta, double cosThetaI) {
    static std::vector<double> gx, gw;
    static std::once_flag once;
    std::call_once(once, [] { gaussLegendre(384, gx, gw); });
    MicrofacetT<double> d(distr, alpha, alpha, true);
    const double pi = 3.14159265358979323846;
    const double wix = std::sqrt(std::max(0.0, 1 - cosThetaI * cosThetaI)), wiz = cosThetaI;
    const int nPhi = 256;
    double sum = 0;
    for (size_t ir = 0; ir < gx.size(); ++ir) {
        /* radial variable r = tan(theta_m)/alpha;  Beckmann: r in [0,7], GGX: r = t/(1-t), t in [0,1) */
        double r, jac;
        if (distr == 0) { r = 3.5 * (gx[ir] + 1); jac = 3.5 * gw[ir]; }
        else { double t = 0.5 * (gx[ir] + 1); r = t / (1 - t); jac = 0.5 * gw[ir] / ((1 - t) * (1 - t)); }
        const double tanT = alpha * r, cosT = 1 / std::sqrt(1 + tanT * tanT), sinT = tanT * cosT;
        /* D(m) dω = D sinθ dθ dφ ; dθ = α cos²θ dr */
        const double Dm_base = sinT * alpha * cosT * cosT * jac;
        double ring = 0;
        for (int ip = 0; ip < nPhi; ++ip) {
            const double phi = 2 * pi * (ip + 0.5) / nPhi;
            const double mx = sinT * std::cos(phi), my = sinT * std::sin(phi), mz = cosT;
            const double D = d.eval(mx, my, mz);
            if (D == 0) continue;
            const double wim = wix * mx + wiz * mz;
            /* fresnelDielectricExt in double (libcore/util.cpp:625-655) */
            double cosTt, F;
            {
                double scale = (wim > 0) ? 1 / eta : eta, c2 = 1 - (1 - wim * wim) * (scale * scale);
                if (c2 <= 0) { cosTt = 0; F = 1; }
                else {
                    double ci = std::fabs(wim), ct = std::sqrt(c2);
                    double Rs = (ci - eta * ct) / (ci + eta * ct), Rp = (eta * ci - ct) / (eta * ci + ct);
                    cosTt = (wim > 0) ? -ct : ct; F = 0.5 * (Rs * Rs + Rp * Rp);
                }
            }
            if (cosTt == 0) continue;
            /* refract(wi, m, eta, cosThetaT): libcore/util.cpp:741-747 */
            double e = eta; if (cosTt < 0) e = 1 / e;
            const double k = wim * e + cosTt;
            const double wox = mx * k - wix * e, woy = my * k, woz = mz * k - wiz * e;
            if (wiz * woz >= 0) continue;
            const double G1i = d.smithG1(wix, 0.0, wiz, mx, my, mz);
            const double G1o = d.smithG1(wox, woy, woz, mx, my, mz);
            ring += (1 - F) * G1o * G1i * std::fabs(wim) * D / wiz;
        }
        sum += ring * (2 * pi / nPhi) * Dm_base;
    }
    return sum;
}

inline RoughTransTable makeRoughTransTable(int distr, Float alpha, Float eta) {
    RoughTransTable tab;
    double vals[100];
    const double step = 1.0 / 99.0;
    for (int i = 0; i < 100; ++i) {
        double t = i * step; if (i == 0) t = step / 10;
        double cosTheta = t * t * t * t;
        vals[i] = roughTransmittanceAt(distr, (double)alpha, (double)eta, cosTheta);
        tab.trans[i] = (Float)vals[i];
    }
    /* diffuse transmittance: 2 int mu * interp(mu^(1/4)) dmu */
    std::vector<double> gx, gw; gaussLegendre(200, gx, gw);
    double acc = 0;
    for (size_t i = 0; i < gx.size(); ++i) {
        double mu = 0.5 * (gx[i] + 1);
        acc += 0.5 * gw[i] * 2 * mu * evalCubicInterp1D<double>(std::pow(mu, 0.25), vals, 100, 0.0, 1.0);
    }
    tab.diffTrans = (Float)acc;
    return tab;
}

/* ---- material table ------------------------------------------------------------------------- */
struct Material {
    int type = 0, distr = 0; bool sampleVisible = true, nonlinear = false;
    int nested[2] = { -1, -1 };
    Float alphaU = 0.1f, alphaV = 0.1f, eta = 1.5f;
    V3 reflectance, specRefl, specTrans, condEta, condK;
    /* roughplastic derived state (roughplastic.cpp:255-307) */
    RoughTransTable extT, intT; Float specSamplingWeight = 0, invEta2 = 0;
    uint32_t typeFlags = 0;   /* BSDF::getType(): OR of the component flags */
    int componentCount = 0;
};

inline V3 ensureEnergyConservation(V3 c) {    /* librender/bsdf.cpp:105-130, constant texture */
    Float m = maxComp(c);
    if (m > 1.0f) { Float scale = 0.99f * (1.0f / m); return c * scale; }
    return c;
}

struct BSDFSample { V3 wo; Float pdf; uint32_t sampledType; V3 weight; };

class BSDFSet {
public:
    std::vector<Material> mats;

    void configure() {
        for (size_t i = 0; i < mats.size(); ++i) configureOne((int)i);
        /* twosided after the nested ones (twosided.cpp:84-106) */
        for (auto &m : mats) if (m.type == 4) {
            if (m.nested[1] < 0) m.nested[1] = m.nested[0];
            const Material &a = mats[m.nested[0]], &b = mats[m.nested[1]];
            m.typeFlags = 0;
            if (a.typeFlags) m.typeFlags |= (a.typeFlags & ~EBackSide) | EFrontSide;
            if (b.typeFlags) m.typeFlags |= (b.typeFlags & ~EFrontSide) | EBackSide;
            m.componentCount = a.componentCount + b.componentCount;
        }
    }
    uint32_t getType(int id) const { return mats[id].typeFlags; }

    V3 eval(int id, V3 wi, V3 wo) const;      /* ESolidAngle measure, value includes cos(theta_o) */
    Float pdf(int id, V3 wi, V3 wo) const;    /* ESolidAngle measure */
    V3 sample(int id, V3 wi, V2 s, V3 &wo, Float &pdf, uint32_t &sampledType) const;

private:
    void configureOne(int i) {
        Material &m = mats[i];
        switch (m.type) {
            case 0: /* diffuse.cpp:93-104 */
                m.reflectance = ensureEnergyConservation(m.reflectance);
                m.typeFlags = maxComp(m.reflectance) > 0 ? (EDiffuseReflection | EFrontSide) : 0;
                m.componentCount = maxComp(m.reflectance) > 0 ? 1 : 0;
                break;
            case 1: /* roughconductor.cpp:228-250 */
                m.specRefl = ensureEnergyConservation(m.specRefl);
                m.typeFlags = EGlossyReflection | EFrontSide; m.componentCount = 1;
                break;
            case 2: { /* roughplastic.cpp:255-307 */
                m.specRefl = ensureEnergyConservation(m.specRefl);
                m.reflectance = ensureEnergyConservation(m.reflectance);
                Float dAvg = luminance(m.reflectance), sAvg = luminance(m.specRefl);
                m.specSamplingWeight = sAvg / (dAvg + sAvg);
                m.invEta2 = 1.0f / (m.eta * m.eta);
                Float a = std::max(m.alphaU, 1e-4f);
                m.extT = cachedTable(m.distr, a, m.eta);
                m.intT = cachedTable(m.distr, a, 1.0f / m.eta);
                m.typeFlags = EGlossyReflection | EDiffuseReflection | EFrontSide; m.componentCount = 2;
                break;
            }
            case 3: /* dielectric.cpp:184-202 */
                m.specRefl = ensureEnergyConservation(m.specRefl);
                m.specTrans = ensureEnergyConservation(m.specTrans);
                m.typeFlags = EDeltaReflection | EDeltaTransmission | EFrontSide | EBackSide; m.componentCount = 2;
                break;
            default: break;
        }
    }
    static RoughTransTable cachedTable(int distr, Float alpha, Float eta) {
        static std::map<std::tuple<int, Float, Float>, RoughTransTable> cache;
        static std::mutex mtx;
        std::lock_guard<std::mutex> g(mtx);
        auto key = std::make_tuple(distr, alpha, eta);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        RoughTransTable t = makeRoughTransTable(distr, alpha, eta);
        cache[key] = t;
        return t;
    }
    static V3 reflectLocal(V3 wi, V3 m) { return 2 * dot(wi, m) * m - wi; }

    /* roughplastic.cpp:407-417,459-467 */
    static Float plasticProbSpecular(const Material &m, Float cosThetaI) {
        Float p = 1 - m.extT.eval(cosThetaI);
        return (p * m.specSamplingWeight) / (p * m.specSamplingWeight + (1 - p) * (1 - m.specSamplingWeight));
    }
};

inline V3 BSDFSet::eval(int id, V3 wi, V3 wo) const {
    const Material &m = mats[id];
    switch (m.type) {
        case 0: { /* diffuse.cpp:110-118 */
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0 || wo.z <= 0) return V3(0.f);
            return m.reflectance * (kInvPi * wo.z);
        }
        case 1: { /* roughconductor.cpp:257-294 */
            if (wi.z < 0 || wo.z < 0) return V3(0.f);
            V3 H = normalize(wo + wi);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible);
            const Float D = distr.evalV(H);
            if (D == 0) return V3(0.f);
            const V3 F = fresnelConductorExact(dot(wi, H), m.condEta, m.condK) * m.specRefl;
            const Float G = distr.G(wi, wo, H);
            Float model = D * G / (4.0f * wi.z);
            return F * model;
        }
        case 2: { /* roughplastic.cpp:326-382 */
            if (wi.z <= 0 || wo.z <= 0) return V3(0.f);
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible);
            V3 result(0.f);
            {
                const V3 H = normalize(wo + wi);
                const Float D = distr.evalV(H);
                const Float F = fresnelDielectricExt(dot(wi, H), m.eta);
                const Float G = distr.G(wi, wo, H);
                Float value = F * D * G / (4.0f * wi.z);
                result += m.specRefl * value;
            }
            {
                V3 diff = m.reflectance;
                Float T12 = m.extT.eval(wi.z), T21 = m.extT.eval(wo.z);
                Float Fdr = 1 - m.intT.evalDiffuse();
                if (m.nonlinear) diff = diff / (V3(1.0f) - diff * Fdr); else diff = diff / (1 - Fdr);
                result += diff * (kInvPi * wo.z * T12 * T21 * m.invEta2);
            }
            return result;
        }
        case 3: /* dielectric.cpp:228-253: nothing for the solid-angle measure */
            return V3(0.f);
        case 4: { /* twosided.cpp:108-120 */
            if (wi.z > 0) return eval(m.nested[0], wi, wo);
            wi.z *= -1; wo.z *= -1;
            return eval(m.nested[1], wi, wo);
        }
    }
    return V3(0.f);
}

inline Float BSDFSet::pdf(int id, V3 wi, V3 wo) const {
    const Material &m = mats[id];
    switch (m.type) {
        case 0: /* diffuse.cpp:120-127 */
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0 || wo.z <= 0) return 0.0f;
            return squareToCosineHemispherePdf(wo);
        case 1: { /* roughconductor.cpp:296-321 */
            if (wi.z < 0 || wo.z < 0) return 0.0f;
            V3 H = normalize(wo + wi);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible);
            if (distr.sampleVisibleFlag) return distr.evalV(H) * distr.G1(wi, H) / (4.0f * wi.z);
            return distr.pdf(wi, H) / (4 * absDot(wo, H));
        }
        case 2: { /* roughplastic.cpp:384-437 */
            if (wi.z <= 0 || wo.z <= 0) return 0.0f;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible);
            const V3 H = normalize(wo + wi);
            Float probSpecular = plasticProbSpecular(m, wi.z), probDiffuse = 1 - probSpecular;
            const Float dwh_dwo = 1.0f / (4.0f * dot(wo, H));
            const Float prob = distr.pdf(wi, H);
            Float result = prob * dwh_dwo * probSpecular;
            result += probDiffuse * squareToCosineHemispherePdf(wo);
            return result;
        }
        case 3: return 0.0f;
        case 4: /* twosided.cpp:122-134 */
            if (wi.z > 0) return pdf(m.nested[0], wi, wo);
            wi.z *= -1; wo.z *= -1;
            return pdf(m.nested[1], wi, wo);
    }
    return 0.0f;
}

inline V3 BSDFSet::sample(int id, V3 wi, V2 s, V3 &wo, Float &pdfOut, uint32_t &sampledType) const {
    const Material &m = mats[id];
    sampledType = 0; pdfOut = 0;
    switch (m.type) {
        case 0: { /* diffuse.cpp:140-150 */
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0) return V3(0.f);
            wo = squareToCosineHemisphere(s);
            sampledType = EDiffuseReflection;
            pdfOut = squareToCosineHemispherePdf(wo);
            return m.reflectance;
        }
        case 1: { /* roughconductor.cpp:369-415 */
            if (wi.z < 0) return V3(0.f);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible);
            V3 mm = distr.sample(wi, s, pdfOut);
            if (pdfOut == 0) return V3(0.f);
            wo = reflectLocal(wi, mm);
            sampledType = EGlossyReflection;
            if (wo.z <= 0) return V3(0.f);
            V3 F = fresnelConductorExact(dot(wi, mm), m.condEta, m.condK) * m.specRefl;
            Float weight;
            if (distr.sampleVisibleFlag) weight = distr.G1(wo, mm);
            else weight = distr.evalV(mm) * distr.G(wi, wo, mm) * dot(wi, mm) / (pdfOut * wi.z);
            pdfOut /= 4.0f * dot(wo, mm);
            return F * weight;
        }
        case 2: { /* roughplastic.cpp:439-501 */
            if (wi.z <= 0) return V3(0.f);
            bool choseSpecular = true;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible);
            Float probSpecular = plasticProbSpecular(m, wi.z);
            if (s.y < probSpecular) s.y /= probSpecular;
            else { s.y = (s.y - probSpecular) / (1 - probSpecular); choseSpecular = false; }
            if (choseSpecular) {
                Float dummy;
                V3 mm = distr.sampleVisibleFlag ? distr.sampleVisible(wi, s) : distr.sampleAll(s, dummy);
                wo = reflectLocal(wi, mm);
                sampledType = EGlossyReflection;
                if (wo.z <= 0) return V3(0.f);
            } else {
                sampledType = EDiffuseReflection;
                wo = squareToCosineHemisphere(s);
            }
            pdfOut = pdf(id, wi, wo);
            if (pdfOut == 0) return V3(0.f);
            return eval(id, wi, wo) / pdfOut;
        }
        case 3: { /* dielectric.cpp:277-308, ERadiance mode */
            Float cosThetaT;
            Float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (s.x <= F) {
                sampledType = EDeltaReflection;
                wo = V3(-wi.x, -wi.y, wi.z);
                pdfOut = F;
                return m.specRefl;
            } else {
                sampledType = EDeltaTransmission;
                Float invEta = 1 / m.eta;
                Float scale = -(cosThetaT < 0 ? invEta : m.eta);
                wo = V3(scale * wi.x, scale * wi.y, cosThetaT);
                pdfOut = 1 - F;
                Float factor = cosThetaT < 0 ? invEta : m.eta;
                return m.specTrans * (factor * factor);
            }
        }
        case 4: { /* twosided.cpp:161-183 */
            bool flipped = false;
            if (wi.z < 0) { wi.z *= -1; flipped = true; }
            V3 result = sample(m.nested[flipped ? 1 : 0], wi, s, wo, pdfOut, sampledType);
            if (flipped && !isZero(result) && pdfOut != 0) wo.z *= -1;
            return result;
        }
    }
    return V3(0.f);
}

} /* namespace ora */
