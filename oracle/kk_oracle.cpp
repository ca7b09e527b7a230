/*
 * ORACLE (test infrastructure only) -- C entry points for ctypes.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libkkdirect.so) never links or calls it.
 * Build: make -C oracle   ->  oracle/build/libkkoracle.so
 */
#include "ora_scene.h"
#include <string>
#include <chrono>

using namespace ora;

struct ora_ctx { Scene scene; std::string err; };

extern "C" {

ora_ctx *ora_create(const kkd_scene_desc *sd, int brute_force) {
    ora_ctx *c = new ora_ctx();
    c->scene.useBruteForce = brute_force != 0;
    if (!c->scene.load(sd, c->err)) { fprintf(stderr, "ora_create: %s\n", c->err.c_str()); delete c; return nullptr; }
    return c;
}
void ora_destroy(ora_ctx *c) { delete c; }

/* mode: 0 = kd-tree (Havran), 1 = brute force over all triangles.  counts (nullable): nodes, tris. */
int ora_trace(ora_ctx *c, const float *rays, uint64_t n, int any_hit, int mode, kkd_hit *out, uint64_t *counts) {
    uint64_t st[2] = { 0, 0 };
    for (uint64_t i = 0; i < n; ++i) {
        const float *r = rays + 8 * i;
        Ray ray(V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]);
        Hit h; h.t = std::numeric_limits<Float>::infinity(); h.prim = kNoPrim; h.u = h.v = 0;
        bool found;
        if (mode == 1) found = c->scene.kd.bruteForce(ray, any_hit != 0, h);
        else if (any_hit) { found = c->scene.kd.rayIntersectAny(ray, st); if (found) { h.prim = 0; h.t = 0; } }
        else found = c->scene.kd.rayIntersectClosest(ray, h, st);
        if (!found) { h.t = std::numeric_limits<Float>::infinity(); h.prim = kNoPrim; h.u = h.v = 0; }
        out[i].t = h.t; out[i].u = h.u; out[i].v = h.v; out[i].prim = h.prim;
    }
    if (counts) { counts[0] = st[0]; counts[1] = st[1]; }
    return 0;
}

/* Full hit record for one ray (test_dgeom vectors): out = p(3) uv(2) geoN(3) shN(3) dpdu(3) dpdv(3) wi(3) t(1) = 21 floats */
int ora_intersect_record(ora_ctx *c, const float *r, float *out) {
    Ray ray(V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]);
    Hit h;
    if (!c->scene.rayIntersect(ray, h, nullptr)) return 0;
    Intersection its; c->scene.fillIntersection(ray, h, its);
    float v[21] = { its.p.x, its.p.y, its.p.z, its.uv.x, its.uv.y, its.geoFrame.n.x, its.geoFrame.n.y, its.geoFrame.n.z,
                    its.shFrame.n.x, its.shFrame.n.y, its.shFrame.n.z, its.dpdu.x, its.dpdu.y, its.dpdu.z,
                    its.dpdv.x, its.dpdv.y, its.dpdv.z, its.wi.x, its.wi.y, its.wi.z, its.t };
    std::memcpy(out, v, sizeof(v));
    return 1;
}

int ora_render(ora_ctx *c, const kkd_render_params *p, uint64_t seed, int threads, float *film, kkd_stats *stats) {
    if (p->integrator == KKD_INTEGRATOR_PATH) { if (!p->spp || p->max_depth == 0 || p->rr_depth <= 0) return KKD_ERR_INVALID; }
    else if (!p->spp || p->bsdf_samples + p->light_samples == 0) return KKD_ERR_UNSUPPORTED;
    if (p->integrator == KKD_INTEGRATOR_KK && (!p->bsdf_samples || !p->light_samples)) return KKD_ERR_UNSUPPORTED;
    RenderParams rp{ p->spp, p->bsdf_samples, p->light_samples, p->hide_emitters != 0, p->block_size, p->tile_index, p->tile_count, p->sample_first, seed, p->integrator, p->strict_normals != 0, p->max_depth, p->rr_depth, p->sampler, p->sampler_dimension, p->sample_count_total };
    rp.prepare(c->scene.sensor.cropW, c->scene.sensor.cropH);
    Stats st;
    auto t0 = std::chrono::steady_clock::now();
    c->scene.render(rp, film, &st, threads);
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = st.samples; stats->rays_closest = st.raysClosest; stats->rays_shadow = st.raysShadow;
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
    return 0;
}

/* Radiance of single camera samples, no film: pix = n x (x, y, sampleIndex); out = n x (r,g,b,alpha,sx,sy). */
int ora_li_samples(ora_ctx *c, const kkd_render_params *p, uint64_t seed, const uint32_t *pix, uint64_t n, float *out) {
    RenderParams rp{ p->spp, p->bsdf_samples, p->light_samples, p->hide_emitters != 0, p->block_size, 0, 1, 0, seed, p->integrator, p->strict_normals != 0, p->max_depth, p->rr_depth, p->sampler, p->sampler_dimension, p->sample_count_total };
    rp.prepare(c->scene.sensor.cropW, c->scene.sensor.cropH);
    const int W = c->scene.sensor.cropW;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t X = pix[3 * i], Y = pix[3 * i + 1], s = pix[3 * i + 2];
        const int edge = c->scene.hqEdges ? c->scene.filter.borderSize : 0;   /* same pixel key as Scene::render */
        SampleStream rng(seed, (uint32_t)(((int)Y + edge) * (W + 2 * edge) + ((int)X + edge)), s);
        rp.configure(rng, (int)X, (int)Y);
        V2 jit = rng.next2D();
        V2 sp{ (Float)X + jit.x, (Float)Y + jit.y };
        Ray ray = c->scene.sensor.sampleRay(sp, 1.0f / std::sqrt((Float)rp.total()));
        Float alpha = 1.0f;
        V3 L = c->scene.Li(ray, rng, rp, alpha, nullptr);
        float *o = out + 6 * i; o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = alpha; o[4] = sp.x; o[5] = sp.y;
    }
    return 0;
}

/* camera rays: pix = n x (sx, sy) sample positions -> rays n x 8 */
int ora_camera_rays(ora_ctx *c, const float *samplePos, uint64_t n, float *rays) {
    for (uint64_t i = 0; i < n; ++i) {
        Ray r = c->scene.sensor.sampleRay(V2{ samplePos[2 * i], samplePos[2 * i + 1] });
        float *o = rays + 8 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.mint; o[4] = r.d.x; o[5] = r.d.y; o[6] = r.d.z; o[7] = r.maxt;
    }
    return 0;
}

/* ---- unit probes ----------------------------------------------------------------------------- */
void ora_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = { ctr[0], ctr[1], ctr[2], ctr[3] };
    Philox::gen(c, key[0], key[1]);
    std::memcpy(out, c, 16);
}
void ora_sample_stream(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float *out) {
    SampleStream s(seed, pixel, sample);
    for (int i = 0; i < n; ++i) out[i] = s.next1D();
}
/* ldsampler values of samples 0 ... count-1 of one pixel: kind 0 = 2D request `index`, 1 = 1D request `index`, 2 = array `index` */
void ora_ldsampler(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t arrayN, uint32_t setSize, uint32_t count, float *out) {
    for (uint32_t j = 0; j < count; ++j) {
        SampleStream s(seed, pixel, j);
        s.useLdSampler(setSize, 0xFFFFFFFFu);
        if (kind == 0) { s.dim2D = index; V2 v = s.next2D(); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) { s.dim1D = index; out[j] = s.next1D(); }
        else for (uint32_t e = 0; e < arrayN; ++e) { V2 v = s.next2DArray(index, arrayN, e); out[2 * ((size_t)j * arrayN + e)] = v.x; out[2 * ((size_t)j * arrayN + e) + 1] = v.y; }
    }
}
/* the same for the stratified sampler; flatDim = dimension of the independent stream that supplies the offsets inside the cells */
void ora_stratified(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t arrayN, uint32_t res, uint32_t flatDim, uint32_t count, float *out) {
    for (uint32_t j = 0; j < count; ++j) {
        SampleStream s(seed, pixel, j);
        s.useStratified(res, 0xFFFFFFFFu);
        s.dim = flatDim;
        if (kind == 0) { s.dim2D = index; V2 v = s.next2D(); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) { s.dim1D = index; out[j] = s.next1D(); }
        else for (uint32_t e = 0; e < arrayN; ++e) { V2 v = s.next2DArray(index, arrayN, e); out[2 * ((size_t)j * arrayN + e)] = v.x; out[2 * ((size_t)j * arrayN + e) + 1] = v.y; }
    }
}
/* halton sampler on a resX x resY film, samples 0 ... count-1 of pixel (x, y): kind 0 = next2D at running dimension dim, 1 = next1D, 2 = array dim */
void ora_halton(uint32_t resX, uint32_t resY, int32_t x, int32_t y, int faure, int kind, uint32_t dim, uint32_t arrayN, uint32_t nArrays, uint32_t count, float *out) {
    HaltonCfg cfg;
    if (nArrays == 0xFFFFFFFFu) cfg.setFilmResolutionHammersley((int)resX, (int)resY, faure != 0, arrayN);   /* hammersley with sampleCount = arrayN */
    else cfg.setFilmResolution((int)resX, (int)resY, faure != 0, nArrays);
    for (uint32_t j = 0; j < count; ++j) {
        SampleStream s(0, 0, j);
        s.useHalton(&cfg, x, y); s.hdim = dim;
        if (kind == 0) { V2 v = s.next2D(); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = s.next1D();
        else for (uint32_t e = 0; e < arrayN; ++e) { V2 v = s.next2DArray(dim, arrayN, e); out[2 * ((size_t)j * arrayN + e)] = v.x; out[2 * ((size_t)j * arrayN + e) + 1] = v.y; }
    }
}
/* sobol sampler, arguments as ora_halton; and its building blocks: sampleSingle's 32 bits, look_up's index */
void ora_sobol(uint32_t resX, uint32_t resY, int32_t x, int32_t y, int kind, uint32_t dim, uint32_t arrayN, uint32_t nArrays, uint32_t count, float *out) {
    SobolCfg cfg; cfg.setFilmResolution((int)resX, (int)resY, nArrays);
    for (uint32_t j = 0; j < count; ++j) {
        SampleStream s(0, 0, j);
        s.useSobol(&cfg, x, y); s.hdim = dim;
        if (kind == 0) { V2 v = s.next2D(); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = s.next1D();
        else for (uint32_t e = 0; e < arrayN; ++e) { V2 v = s.next2DArray(dim, arrayN, e); out[2 * ((size_t)j * arrayN + e)] = v.x; out[2 * ((size_t)j * arrayN + e) + 1] = v.y; }
    }
}
uint32_t ora_sobol_bits(uint64_t index, uint32_t dim) { return SobolCfg::bits(index, dim); }
uint64_t ora_sobol_look_up(uint32_t m, uint32_t frame, uint32_t px, uint32_t py) { return SobolCfg::lookUp(m, frame, px, py); }
void ora_clip_aabb(const float *tri9, const float *box6, float *out6) {
    AABB b(V3(box6[0], box6[1], box6[2]), V3(box6[3], box6[4], box6[5]));
    AABB r = triClippedAABB(V3(tri9[0], tri9[1], tri9[2]), V3(tri9[3], tri9[4], tri9[5]), V3(tri9[6], tri9[7], tri9[8]), b);
    out6[0] = r.min.x; out6[1] = r.min.y; out6[2] = r.min.z; out6[3] = r.max.x; out6[4] = r.max.y; out6[5] = r.max.z;
}
/* BSDF probes on the scene's material table; wi, wo in the local shading frame */
void ora_bsdf_eval(ora_ctx *c, int mat, const float *wi, const float *wo, float *f3, float *pdf) {
    V3 f = c->scene.bsdfs.eval(mat, V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]));
    f3[0] = f.x; f3[1] = f.y; f3[2] = f.z;
    *pdf = c->scene.bsdfs.pdf(mat, V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]));
}
void ora_bsdf_sample(ora_ctx *c, int mat, const float *wi, const float *u2, uint64_t n, float *out) {
    /* out: n x (wo3, weight3, pdf, type) */
    for (uint64_t i = 0; i < n; ++i) {
        V3 wo(0.f); Float pdf = 0; uint32_t ty = 0;
        V3 w = c->scene.bsdfs.sample(mat, V3(wi[0], wi[1], wi[2]), V2{ u2[2 * i], u2[2 * i + 1] }, wo, pdf, ty);
        float *o = out + 8 * i;
        o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = w.x; o[4] = w.y; o[5] = w.z; o[6] = pdf; o[7] = (float)ty;
    }
}
uint32_t ora_bsdf_type(ora_ctx *c, int mat) { return c->scene.bsdfs.getType(mat); }
/* microfacet probes (tests/test_microfacet.cpp) */
void ora_microfacet_sample(int type, float au, float av, int visible, const float *wi, const float *u2, uint64_t n, float *out) {
    Microfacet d(type, au, av, visible != 0);
    for (uint64_t i = 0; i < n; ++i) {
        Float pdf; V3 m = d.sample(V3(wi[0], wi[1], wi[2]), V2{ u2[2 * i], u2[2 * i + 1] }, pdf);
        float *o = out + 4 * i; o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = pdf;
    }
}
void ora_microfacet_eval(int type, float au, float av, int visible, const float *wi, const float *m3, uint64_t n, float *out) {
    /* out: n x (D, pdf(wi,m), G1(wi,m)) */
    Microfacet d(type, au, av, visible != 0);
    V3 w(wi[0], wi[1], wi[2]);
    for (uint64_t i = 0; i < n; ++i) {
        V3 m(m3[3 * i], m3[3 * i + 1], m3[3 * i + 2]);
        out[3 * i] = d.evalV(m); out[3 * i + 1] = d.pdf(w, m); out[3 * i + 2] = d.G1(w, m);
    }
}
void ora_rough_transmittance(int distr, float alpha, float eta, float *trans100, float *diffTrans) {
    RoughTransTable t = makeRoughTransTable(distr, alpha, eta);
    std::memcpy(trans100, t.trans, sizeof(t.trans)); *diffTrans = t.diffTrans;
}
double ora_rough_transmittance_at(int distr, double alpha, double eta, double cosTheta) {
    return roughTransmittanceAt(distr, alpha, eta, cosTheta);
}
void ora_rfilter_table(int type, float radius, const float *param, float *values32, float *scaleFactor, int *border) {
    RFilter f; f.configure(type, radius, param);
    std::memcpy(values32, f.values, sizeof(f.values)); *scaleFactor = f.scaleFactor; *border = f.borderSize;
}
/* emitter probes: sampleEmitterDirect without visibility is not separable in the reference, so the probe
   reports the full call: out = n x (value3, d3, dist, pdf, emitterIndex) from ref point/normal */
void ora_sample_emitter_direct(ora_ctx *c, const float *ref, const float *refN, const float *u2, uint64_t n, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        DirectRec d; d.ref = V3(ref[0], ref[1], ref[2]); d.refN = V3(refN[0], refN[1], refN[2]); d.emitter = -1;
        V3 v = c->scene.sampleEmitterDirect(d, V2{ u2[2 * i], u2[2 * i + 1] }, nullptr);
        float *o = out + 9 * i;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = d.d.x; o[4] = d.d.y; o[5] = d.d.z; o[6] = d.dist; o[7] = isZero(v) ? 0.f : d.pdf; o[8] = (float)d.emitter;
    }
}
void ora_env_probe(ora_ctx *c, const float *dirs, uint64_t n, float *out) {
    /* out: n x (radiance3, pdfDirection) for world directions */
    for (uint64_t i = 0; i < n; ++i) {
        V3 d(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        V3 v = c->scene.env.evalEnvironment(d);
        out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z;
        out[4 * i + 3] = c->scene.env.internalPdfDirection(c->scene.env.rot(c->scene.env.toLocal, d));
    }
}
void ora_kd_info(ora_ctx *c, uint64_t *nNodes, uint64_t *nIndices, int *maxDepth) {
    *nNodes = c->scene.kd.nodes.size(); *nIndices = c->scene.kd.indices.size(); *maxDepth = c->scene.kd.maxDepthSeen;
}

} /* extern "C" */
