/*
 * kkd_bsdf.h -- device-side BSDF models of the kk direct path: eval / pdf / sample in the local shading
 * frame, solid-angle measure, eval() includes the cosine foreshortening term (Mitsuba convention).
 *
 * Semantics follow the reference (paths relative to /root/reference/src):
 *   bsdfs/microfacet.h:191-234 eval, :287-395 sampleAll, :421-464 sampleVisible/pdfVisible,
 *                      :475-512 smithG1, :570-694 sampleVisible11
 *   bsdfs/diffuse.cpp:110-150, bsdfs/roughconductor.cpp:257-415, bsdfs/roughplastic.cpp:326-501,
 *   bsdfs/dielectric.cpp:217-333, bsdfs/twosided.cpp:108-183, bsdfs/rtrans.h:183-191 (table lookup),
 *   libcore/spline.cpp:23-60 (Catmull-Rom interpolation)
 * Materials are parameter blocks (DMaterial) switched on `type`; no virtual calls, no heap.
 */
#pragma once
#include "kkd_types.h"

namespace kkd {

struct Microfacet {
    int type; float alphaU, alphaV; bool sampleVisible; float expU, expV;
    __device__ Microfacet(int t, float au, float av, bool sv) : type(t), alphaU(au), alphaV(av), sampleVisible(sv), expU(0.f), expV(0.f) {
        if (type == KKD_DISTR_PHONG) {
            sampleVisible = false;
            expU = fmaxf(fdiv(2.0f, alphaU * alphaU) - 2.0f, 0.0f);
            expV = fmaxf(fdiv(2.0f, alphaV * alphaV) - 2.0f, 0.0f);
        }
    }
    __device__ bool isotropic() const { return alphaU == alphaV; }
    __device__ float projectRoughness(f3 v) const {
        if (isotropic()) return alphaU;
        float invSinTheta2 = frcp(1.f - v.z * v.z);
        if (invSinTheta2 <= 0.f) return alphaU;
        float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return fsqrt(cosPhi2 * alphaU * alphaU + sinPhi2 * alphaV * alphaV);
    }
    __device__ float interpolatePhongExponent(f3 v) const {
        if (isotropic()) return expU;
        float invSinTheta2 = frcp(1.f - v.z * v.z);
        if (invSinTheta2 <= 0.f) return expU;
        float cosPhi2 = v.x * v.x * invSinTheta2, sinPhi2 = v.y * v.y * invSinTheta2;
        return expU * cosPhi2 + expV * sinPhi2;
    }
    __device__ float eval(f3 m) const {
        if (m.z <= 0.f) return 0.0f;
        float cosTheta2 = m.z * m.z;
        float beckmannExponent = (isotropic() ? fdiv(m.x * m.x + m.y * m.y, alphaU * alphaU)
                                               : fdiv(m.x * m.x, alphaU * alphaU) + fdiv(m.y * m.y, alphaV * alphaV)) * frcp(cosTheta2);
        float result;
        if (type == KKD_DISTR_BECKMANN) {
            result = fdiv(fexp(-beckmannExponent), kPi * alphaU * alphaV * cosTheta2 * cosTheta2);
        } else if (type == KKD_DISTR_GGX) {
            float root = (1.0f + beckmannExponent) * cosTheta2;
            result = frcp(kPi * alphaU * alphaV * root * root);
        } else {
            float exponent = interpolatePhongExponent(m);
            result = fsqrt((expU + 2.f) * (expV + 2.f)) * kInvTwoPi * powf(m.z, exponent);
        }
        if (result * m.z < 1e-20f) result = 0.f;
        return result;
    }
    __device__ float smithG1(f3 v, f3 m) const {
        if (dot(v, m) * v.z <= 0.f) return 0.0f;
        float temp = 1.f - v.z * v.z;
        float tanTheta = temp <= 0.f ? 0.f : fabsf(fdiv(fsqrt(temp), v.z));
        if (tanTheta == 0.0f) return 1.0f;
        float alpha = projectRoughness(v);
        if (type == KKD_DISTR_GGX) {
            float root = alpha * tanTheta;
            return fdiv(2.0f, 1.0f + fsqrt(1.0f + root * root));
        }
        float a = frcp(alpha * tanTheta);
        if (a >= 1.6f) return 1.0f;
        float aSqr = a * a;
        return fdiv(3.535f * a + 2.181f * aSqr, 1.0f + 2.276f * a + 2.577f * aSqr);
    }
    __device__ float G(f3 wi, f3 wo, f3 m) const { return smithG1(wi, m) * smithG1(wo, m); }
    __device__ float pdfAll(f3 m) const { return eval(m) * m.z; }
    __device__ float pdfVisible(f3 wi, f3 m) const { return fdiv(smithG1(wi, m) * fabsf(dot(wi, m)) * eval(m), fabsf(wi.z)); }
    __device__ float pdf(f3 wi, f3 m) const { return sampleVisible ? pdfVisible(wi, m) : pdfAll(m); }

    __device__ void sampleFirstQuadrant(float u1, float &phi, float &exponent) const {
        phi = atanf(fsqrt(fdiv(expU + 2.0f, expV + 2.0f)) * tanf(kPi * u1 * 0.5f));
        float c = cosf(phi), s = sinf(phi);
        exponent = expU * c * c + expV * s * s;
    }
    __device__ f3 sampleAll(f2 sample, float &pdf) const {
        float cosThetaM = 0.0f, sinPhiM, cosPhiM, alphaSqr;
        if (type != KKD_DISTR_PHONG) {
            if (isotropic()) {
                sincosf((2.0f * kPi) * sample.y, &sinPhiM, &cosPhiM);
                alphaSqr = alphaU * alphaU;
            } else {
                float phiM = atanf(fdiv(alphaV, alphaU) * tanf(kPi + 2.f * kPi * sample.y)) + kPi * floorf(2.f * sample.y + 0.5f);
                sincosf(phiM, &sinPhiM, &cosPhiM);
                float cosSc = fdiv(cosPhiM, alphaU), sinSc = fdiv(sinPhiM, alphaV);
                alphaSqr = frcp(cosSc * cosSc + sinSc * sinSc);
            }
            if (type == KKD_DISTR_BECKMANN) {
                float tanThetaMSqr = alphaSqr * -logf(1.0f - sample.x);
                cosThetaM = frsq(1.0f + tanThetaMSqr);
                pdf = fdiv(1.0f - sample.x, kPi * alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM);
            } else {
                float tanThetaMSqr = fdiv(alphaSqr * sample.x, 1.0f - sample.x);
                cosThetaM = frsq(1.0f + tanThetaMSqr);
                float temp = 1.f + fdiv(tanThetaMSqr, alphaSqr);
                pdf = fdiv(kInvPi, alphaU * alphaV * cosThetaM * cosThetaM * cosThetaM * temp * temp);
            }
        } else {
            float phiM, exponent;
            if (isotropic()) { phiM = (2.0f * kPi) * sample.y; exponent = expU; }
            else if (sample.y < 0.25f) { sampleFirstQuadrant(4.f * sample.y, phiM, exponent); }
            else if (sample.y < 0.5f) { sampleFirstQuadrant(4.f * (0.5f - sample.y), phiM, exponent); phiM = kPi - phiM; }
            else if (sample.y < 0.75f) { sampleFirstQuadrant(4.f * (sample.y - 0.5f), phiM, exponent); phiM += kPi; }
            else { sampleFirstQuadrant(4.f * (1.f - sample.y), phiM, exponent); phiM = 2.f * kPi - phiM; }
            sincosf(phiM, &sinPhiM, &cosPhiM);
            cosThetaM = powf(sample.x, frcp(exponent + 2.0f));
            pdf = fsqrt((expU + 2.0f) * (expV + 2.0f)) * kInvTwoPi * powf(cosThetaM, exponent + 1.0f);
        }
        if (pdf < 1e-20f) pdf = 0.f;
        float sinThetaM = fsqrt(fmaxf(0.f, 1.f - cosThetaM * cosThetaM));
        return f3(sinThetaM * cosPhiM, sinThetaM * sinPhiM, cosThetaM);
    }
    /* tanThetaI = tan(thetaI) is handed in: the caller has cos(thetaI), and sqrt(1 - c^2) / c is 3 instructions where tanf is ~80 */
    __device__ f2 sampleVisible11(float thetaI, float tanThetaI, f2 sample) const {
        const float SQRT_PI_INV = 0.56418958354775628f;
        f2 slope;
        if (type == KKD_DISTR_BECKMANN) {
            if (thetaI < 1e-4f) {
                float r = fsqrt(-logf(1.0f - sample.x)), s, c;
                sincosf(2.f * kPi * sample.y, &s, &c);
                return f2{ r * c, r * s };
            }
            float cotThetaI = frcp(tanThetaI);
            float a = -1.f, c = erfApprox(cotThetaI);
            float sample_x = fmaxf(sample.x, 1e-6f);
            float fit = 1.f + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
            float b = c - (1.f + c) * fpow(1.f - sample_x, fit);
            float normalization = frcp(1.f + c + SQRT_PI_INV * tanThetaI * fexp(-cotThetaI * cotThetaI));
            int it = 0;
            while (++it < 10) {
                if (!(b >= a && b <= c)) b = 0.5f * (a + c);
                float invErf = erfinvApprox(b);
                float value = normalization * (1.f + b + SQRT_PI_INV * tanThetaI * fexp(-invErf * invErf)) - sample_x;
                float derivative = normalization * (1.f - invErf * tanThetaI);
                if (fabsf(value) < 1e-5f) break;
                if (value > 0.f) c = b; else a = b;
                b -= fdiv(value, derivative);
            }
            slope.x = erfinvApprox(b);
            slope.y = erfinvApprox(2.0f * fmaxf(sample.y, 1e-6f) - 1.0f);
        } else {
            if (thetaI < 1e-4f) {
                float r = safeSqrt(fdiv(sample.x, 1.f - sample.x)), s, c;
                sincosf(2.f * kPi * sample.y, &s, &c);
                return f2{ r * c, r * s };
            }
            float a = frcp(tanThetaI);
            float G1 = fdiv(2.0f, 1.0f + safeSqrt(1.0f + frcp(a * a)));
            float A = fdiv(2.0f * sample.x, G1) - 1.0f;
            if (fabsf(A) == 1.f) A -= signum(A) * kEpsilon;
            float tmp = frcp(A * A - 1.0f);
            float B = tanThetaI;
            float D = safeSqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
            float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
            slope.x = (A < 0.0f || slope_x_2 > a) ? slope_x_1 : slope_x_2;
            float S;
            if (sample.y > 0.5f) { S = 1.0f; sample.y = 2.0f * (sample.y - 0.5f); }
            else { S = -1.0f; sample.y = 2.0f * (0.5f - sample.y); }
            float z = fdiv(sample.y * (sample.y * (sample.y * (-0.365728915865723f) + 0.790235037209296f) - 0.424965825137544f) + 0.000152998850436920f,
                           sample.y * (sample.y * (sample.y * (sample.y * 0.169507819808272f - 0.397203533833404f) - 0.232500544458471f) + 1.0f) - 0.539825872510702f);
            slope.y = S * z * fsqrt(1.0f + slope.x * slope.x);
        }
        return slope;
    }
    __device__ f3 sampleVisibleNormal(f3 _wi, f2 sample) const {
        f3 wi = normalize(f3(alphaU * _wi.x, alphaV * _wi.y, _wi.z));
        /* microfacet.h:421-459 takes theta = acos(wi.z), phi = atan2(wi.y, wi.x) and then sin / cos / tan of them: the same
         * numbers come straight from the components (cos phi = x / |xy|, tan theta = |xy| / z); only Beckmann's fit needs theta */
        float theta = 0.f, tanTheta = 0.f, sinPhi = 0.f, cosPhi = 1.f;
        if (wi.z < 0.99999f) {
            const float s2 = wi.x * wi.x + wi.y * wi.y, r = frsq(s2);
            cosPhi = wi.x * r; sinPhi = wi.y * r;
            tanTheta = fdiv(s2 * r, wi.z);
            theta = type == KKD_DISTR_BECKMANN ? acosf(wi.z) : 1.0f;   /* GGX only tests theta < 1e-4 */
        }
        f2 slope = sampleVisible11(theta, tanTheta, sample);
        slope = f2{ cosPhi * slope.x - sinPhi * slope.y, sinPhi * slope.x + cosPhi * slope.y };
        slope.x *= alphaU; slope.y *= alphaV;
        float normalization = frsq(slope.x * slope.x + slope.y * slope.y + 1.0f);
        return f3(-slope.x * normalization, -slope.y * normalization, normalization);
    }
    __device__ f3 sample(f3 wi, f2 s, float &pdfOut) const {
        if (sampleVisible) { f3 m = sampleVisibleNormal(wi, s); pdfOut = pdfVisible(wi, m); return m; }
        return sampleAll(s, pdfOut);
    }
};

__device__ __forceinline__ f3 ld3(const float *p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ f3 reflectLocal(f3 wi, f3 m) { return 2.f * dot(wi, m) * m - wi; }

/* rtrans.h:183-191: Catmull-Rom lookup in cosTheta^(1/4), clamped to [0,1] */
__device__ __forceinline__ float roughTransEval(const float *tab, float cosTheta) {
    float x = fsqrt(fsqrt(fabsf(cosTheta)));   /* cosTheta^(1/4), rtrans.h:187 */
    if (!(x >= 0.f && x <= 1.f)) return 0.f;
    float t = x * 99.f;
    int k = max(0, min((int)t, 98));
    float f0 = tab[k], f1 = tab[k + 1];
    float d0 = k > 0 ? 0.5f * (f1 - tab[k - 1]) : f1 - f0;
    float d1 = k + 2 < 100 ? 0.5f * (tab[k + 2] - f0) : f1 - f0;
    t -= (float)k;
    float t2 = t * t, t3 = t2 * t;
    float r = (2.f * t3 - 3.f * t2 + 1.f) * f0 + (-2.f * t3 + 3.f * t2) * f1 + (t3 - 2.f * t2 + t) * d0 + (t3 - t2) * d1;
    return fminf(1.0f, fmaxf(0.0f, r));
}
__device__ __forceinline__ float plasticProbSpecular(const DMaterial &m, const float *tab, float cosThetaI) {
    float p = 1.f - roughTransEval(tab, cosThetaI);
    return fdiv(p * m.specSamplingWeight, p * m.specSamplingWeight + (1.f - p) * (1.f - m.specSamplingWeight));
}

/* eval / pdf of a non-twosided material (wi.z sign already resolved by the caller) */
__device__ __forceinline__ f3 bsdfEvalBase(const DMaterial &m, const float *rtTables, f3 wi, f3 wo) {
    switch (m.type) {
        case KKD_BSDF_DIFFUSE:
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0.f || wo.z <= 0.f) return f3(0.f);
            return ld3(m.refl) * (kInvPi * wo.z);
        case KKD_BSDF_ROUGHCONDUCTOR: {
            if (wi.z < 0.f || wo.z < 0.f) return f3(0.f);
            f3 H = normalize(wo + wi);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible != 0);
            float D = distr.eval(H);
            if (D == 0.f) return f3(0.f);
            f3 F = fresnelConductorExact(dot(wi, H), ld3(m.condEta), ld3(m.condK)) * ld3(m.specRefl);
            float model = fdiv(D * distr.G(wi, wo, H), 4.0f * wi.z);
            return F * model;
        }
        case KKD_BSDF_ROUGHPLASTIC: {
            if (wi.z <= 0.f || wo.z <= 0.f) return f3(0.f);
            const float *tab = rtTables + m.tableOffset;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible != 0);
            f3 H = normalize(wo + wi);
            float value = fdiv(fresnelDielectricExt(dot(wi, H), m.eta) * distr.eval(H) * distr.G(wi, wo, H), 4.0f * wi.z);
            f3 result = ld3(m.specRefl) * value;
            f3 diff = ld3(m.refl);
            float T12 = roughTransEval(tab, wi.z), T21 = roughTransEval(tab, wo.z);
            if (m.nonlinear) diff = diff / (f3(1.0f) - diff * m.fdr); else diff = diff * frcp(1.f - m.fdr);
            result += diff * (kInvPi * wo.z * T12 * T21 * m.invEta2);
            return result;
        }
        default: return f3(0.f);   /* dielectric: nothing in the solid-angle measure (dielectric.cpp:228-253) */
    }
}
__device__ __forceinline__ float bsdfPdfBase(const DMaterial &m, const float *rtTables, f3 wi, f3 wo) {
    switch (m.type) {
        case KKD_BSDF_DIFFUSE:
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0.f || wo.z <= 0.f) return 0.f;
            return kInvPi * wo.z;
        case KKD_BSDF_ROUGHCONDUCTOR: {
            if (wi.z < 0.f || wo.z < 0.f) return 0.f;
            f3 H = normalize(wo + wi);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible != 0);
            if (distr.sampleVisible) return fdiv(distr.eval(H) * distr.smithG1(wi, H), 4.0f * wi.z);
            return fdiv(distr.pdf(wi, H), 4.f * fabsf(dot(wo, H)));
        }
        case KKD_BSDF_ROUGHPLASTIC: {
            if (wi.z <= 0.f || wo.z <= 0.f) return 0.f;
            const float *tab = rtTables + m.tableOffset;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible != 0);
            f3 H = normalize(wo + wi);
            float probSpecular = plasticProbSpecular(m, tab, wi.z);
            float result = distr.pdf(wi, H) * frcp(4.0f * dot(wo, H)) * probSpecular;
            result += (1.f - probSpecular) * (kInvPi * wo.z);
            return result;
        }
        default: return 0.f;
    }
}
/* bsdfEvalBase and bsdfPdfBase of ONE pair of directions (the emitter sample's MIS weight needs both, kk/direct.cpp:146-158):
 * half vector, D, G1(wi) and the plastic's T12 are computed once; every expression is the one of the two functions above, so
 * value and density are theirs bit for bit */
__device__ __forceinline__ f3 bsdfEvalPdfBase(const DMaterial &m, const float *rtTables, f3 wi, f3 wo, float &pdf) {
    pdf = 0.f;
    switch (m.type) {
        case KKD_BSDF_DIFFUSE:
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0.f || wo.z <= 0.f) return f3(0.f);
            pdf = kInvPi * wo.z;
            return ld3(m.refl) * (kInvPi * wo.z);
        case KKD_BSDF_ROUGHCONDUCTOR: {
            if (wi.z < 0.f || wo.z < 0.f) return f3(0.f);
            f3 H = normalize(wo + wi);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible != 0);
            const float D = distr.eval(H), G1i = distr.smithG1(wi, H);
            if (distr.sampleVisible) pdf = fdiv(D * G1i, 4.0f * wi.z);
            else pdf = fdiv(D * H.z, 4.f * fabsf(dot(wo, H)));
            if (D == 0.f) return f3(0.f);
            f3 F = fresnelConductorExact(dot(wi, H), ld3(m.condEta), ld3(m.condK)) * ld3(m.specRefl);
            float model = fdiv(D * (G1i * distr.smithG1(wo, H)), 4.0f * wi.z);
            return F * model;
        }
        case KKD_BSDF_ROUGHPLASTIC: {
            if (wi.z <= 0.f || wo.z <= 0.f) return f3(0.f);
            const float *tab = rtTables + m.tableOffset;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible != 0);
            f3 H = normalize(wo + wi);
            const float D = distr.eval(H), G1i = distr.smithG1(wi, H);
            const float T12 = roughTransEval(tab, wi.z), T21 = roughTransEval(tab, wo.z);
            {   /* bsdfPdfBase */
                const float p = 1.f - T12;
                const float probSpecular = fdiv(p * m.specSamplingWeight, p * m.specSamplingWeight + (1.f - p) * (1.f - m.specSamplingWeight));
                const float dpdf = distr.sampleVisible ? fdiv(G1i * fabsf(dot(wi, H)) * D, fabsf(wi.z)) : D * H.z;
                pdf = dpdf * frcp(4.0f * dot(wo, H)) * probSpecular;
                pdf += (1.f - probSpecular) * (kInvPi * wo.z);
            }
            float value = fdiv(fresnelDielectricExt(dot(wi, H), m.eta) * D * (G1i * distr.smithG1(wo, H)), 4.0f * wi.z);
            f3 result = ld3(m.specRefl) * value;
            f3 diff = ld3(m.refl);
            if (m.nonlinear) diff = diff / (f3(1.0f) - diff * m.fdr); else diff = diff * frcp(1.f - m.fdr);
            result += diff * (kInvPi * wo.z * T12 * T21 * m.invEta2);
            return result;
        }
        default: return f3(0.f);
    }
}
/* returns the importance weight f*cos/pdf; sampledType gets the BSDF::EBSDFType bit of the lobe */
__device__ __forceinline__ f3 bsdfSampleBase(const DMaterial &m, const float *rtTables, f3 wi, f2 s, f3 &wo, float &pdfOut, uint32_t &sampledType) {
    sampledType = 0; pdfOut = 0.f; wo = f3(0.f);
    switch (m.type) {
        case KKD_BSDF_DIFFUSE:
            if (!(m.typeFlags & EDiffuseReflection) || wi.z <= 0.f) return f3(0.f);
            wo = squareToCosineHemisphere(s);
            sampledType = EDiffuseReflection;
            pdfOut = kInvPi * wo.z;
            return ld3(m.refl);
        case KKD_BSDF_ROUGHCONDUCTOR: {
            if (wi.z < 0.f) return f3(0.f);
            Microfacet distr(m.distr, m.alphaU, m.alphaV, m.sampleVisible != 0);
            f3 mm = distr.sample(wi, s, pdfOut);
            if (pdfOut == 0.f) return f3(0.f);
            wo = reflectLocal(wi, mm);
            sampledType = EGlossyReflection;
            if (wo.z <= 0.f) return f3(0.f);
            f3 F = fresnelConductorExact(dot(wi, mm), ld3(m.condEta), ld3(m.condK)) * ld3(m.specRefl);
            float weight;
            if (distr.sampleVisible) weight = distr.smithG1(wo, mm);
            else weight = fdiv(distr.eval(mm) * distr.G(wi, wo, mm) * dot(wi, mm), pdfOut * wi.z);
            pdfOut = fdiv(pdfOut, 4.0f * dot(wo, mm));
            return F * weight;
        }
        case KKD_BSDF_ROUGHPLASTIC: {
            if (wi.z <= 0.f) return f3(0.f);
            const float *tab = rtTables + m.tableOffset;
            Microfacet distr(m.distr, m.alphaU, m.alphaU, m.sampleVisible != 0);
            float probSpecular = plasticProbSpecular(m, tab, wi.z);
            bool choseSpecular = true;
            if (s.y < probSpecular) s.y = fdiv(s.y, probSpecular);
            else { s.y = fdiv(s.y - probSpecular, 1.f - probSpecular); choseSpecular = false; }
            if (choseSpecular) {
                float dummy;
                f3 mm = distr.sampleVisible ? distr.sampleVisibleNormal(wi, s) : distr.sampleAll(s, dummy);
                wo = reflectLocal(wi, mm);
                sampledType = EGlossyReflection;
                if (wo.z <= 0.f) return f3(0.f);
            } else {
                sampledType = EDiffuseReflection;
                wo = squareToCosineHemisphere(s);
            }
            const f3 value = bsdfEvalPdfBase(m, rtTables, wi, wo, pdfOut);      /* roughplastic.cpp: pdf(), then eval() / pdf -- one pass here */
            if (pdfOut == 0.f) return f3(0.f);
            return value / pdfOut;
        }
        case KKD_BSDF_DIELECTRIC: {   /* dielectric.cpp:277-308, ERadiance transport */
            float cosThetaT;
            float F = fresnelDielectricExt(wi.z, cosThetaT, m.eta);
            if (s.x <= F) {
                sampledType = EDeltaReflection;
                wo = f3(-wi.x, -wi.y, wi.z);
                pdfOut = F;
                return ld3(m.specRefl);
            }
            sampledType = EDeltaTransmission;
            float factor = cosThetaT < 0.f ? frcp(m.eta) : m.eta;
            wo = f3(-factor * wi.x, -factor * wi.y, cosThetaT);
            pdfOut = 1.f - F;
            return ld3(m.specTrans) * (factor * factor);
        }
        default: return f3(0.f);
    }
}

/* ---- dispatch through the twosided adapter (twosided.cpp:108-183) ---- */
__device__ __forceinline__ f3 evalBsdf(const DScene &sc, int id, f3 wi, f3 wo) {
    const DMaterial *m = sc.materials + id;
    if (m->type == KKD_BSDF_TWOSIDED) {
        if (wi.z > 0.f) m = sc.materials + m->nested0;
        else { wi.z = -wi.z; wo.z = -wo.z; m = sc.materials + m->nested1; }
    }
    return bsdfEvalBase(*m, sc.rtTables, wi, wo);
}
__device__ __forceinline__ float pdfBsdf(const DScene &sc, int id, f3 wi, f3 wo) {
    const DMaterial *m = sc.materials + id;
    if (m->type == KKD_BSDF_TWOSIDED) {
        if (wi.z > 0.f) m = sc.materials + m->nested0;
        else { wi.z = -wi.z; wo.z = -wo.z; m = sc.materials + m->nested1; }
    }
    return bsdfPdfBase(*m, sc.rtTables, wi, wo);
}
__device__ __forceinline__ f3 evalPdfBsdf(const DScene &sc, int id, f3 wi, f3 wo, float &pdf) {
    const DMaterial *m = sc.materials + id;
    if (m->type == KKD_BSDF_TWOSIDED) {
        if (wi.z > 0.f) m = sc.materials + m->nested0;
        else { wi.z = -wi.z; wo.z = -wo.z; m = sc.materials + m->nested1; }
    }
    return bsdfEvalPdfBase(*m, sc.rtTables, wi, wo, pdf);
}
__device__ __forceinline__ f3 sampleBsdf(const DScene &sc, int id, f3 wi, f2 s, f3 &wo, float &pdfOut, uint32_t &sampledType) {
    const DMaterial *m = sc.materials + id;
    bool flipped = false;
    if (m->type == KKD_BSDF_TWOSIDED) {
        if (wi.z < 0.f) { wi.z = -wi.z; flipped = true; m = sc.materials + m->nested1; }
        else m = sc.materials + m->nested0;
    }
    f3 result = bsdfSampleBase(*m, sc.rtTables, wi, s, wo, pdfOut, sampledType);
    if (flipped && !isZero(result) && pdfOut != 0.f) wo.z = -wo.z;
    return result;
}

} /* namespace kkd */
