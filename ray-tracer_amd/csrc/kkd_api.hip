/*
 * kkd_api.hip -- the C ABI of libkkdirect.so (include/kkdirect.h): context, scene upload, render loop,
 * fixed-ray-set tracing.  Host orchestration only; all per-sample work is in kkd_kernels.hip.
 *
 * Replaces, for the kk direct path, SamplingIntegrator::render + BlockedRenderProcess + BlockRenderer
 * (reference src/librender/integrator.cpp:95-129, src/librender/renderproc.cpp:26-184): instead of a
 * work queue of 32x32 RectangularWorkUnits pulled by one thread per core, the tiles assigned to this
 * context are walked in batches of camera samples, each batch being five kernel launches on one stream.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "kkd_host.h"
#include "kkd_kernels.h"
#include "kkd_gpu_build.h"

using namespace kkd;

namespace {


struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) hipFree(p); p = nullptr; bytes = 0; }
};

} /* namespace */

static constexpr int kMaxLanes = 4;
static constexpr size_t kMemHeadroom = 768ull << 20;   /* device memory a render leaves free for the runtime's own allocations */

/* ---- scratch of the entry point that allocates nothing (kkd_trace_device_noalloc) ----------------------------------
 * The sharded ray / traversal counters of a trace launch (kCounterShards lines of 128 bytes = 128 KiB), which the other
 * entry points keep in a hipMalloc'ed buffer of the context.  Every GPU the library runs on holds its own copy of this
 * array until the process ends.  In host code the array's NAME is only a stand-in for the symbol, not a device address:
 * the address on a device comes from hipGetSymbolAddress, once per device with that device current (scratchForDevice,
 * called by kkd_create), and only that pointer is handed to hipMemsetAsync, the copies and the kernels. */
static __device__ __align__(256) unsigned char g_noallocScratch[kCounterShards * sizeof(DeviceCounters)];
static constexpr int kMaxDevices = 64;
static void *g_noallocScratchPtr[kMaxDevices];      /* per device: address of that device's copy, nullptr until looked up */
static std::mutex g_noallocMutex;

/* device must be current */
static hipError_t scratchForDevice(int device, void **out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_noallocMutex);
    if (!g_noallocScratchPtr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noallocScratch));
        if (e != hipSuccess) return e;
        g_noallocScratchPtr[device] = p;
    }
    *out = g_noallocScratchPtr[device];
    return hipSuccess;
}

/* developer switches from the environment are performance knobs only, read once and clamped to what the launch code can take */
static int envInt(const char *text, int dflt, int lo, int hi) {
    if (!text || !*text) return dflt;
    return std::max(lo, std::min(hi, atoi(text)));
}

struct kkd_ctx {
    int device = 0;
    kkd_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t laneStreams[kMaxLanes - 1] = {};    /* further batch lanes of a render (created on first use) */
    std::string err;
    std::atomic<int> cancel{0};
    bool haveScene = false;
    BVH4::Info accelInfo{};
    DScene ds{};
    /* scene buffers */
    DevBuf nodes, wideNodes, tris, projTris, positions, normals, texcoords, indices, shapes, materials, emitters, emitterCdf, emitterGuide, areaGuide, areaCdf, rtTables,
           envTexels, envCdfRows, envCdfCols, envRowWeights, envGuideRows, envGuideCols, envRecRows, envRecCols, envEwaLut;
    /* work buffers */
    DevBuf contRecs, contStacks, contCount, pathAlive, pBsdfRays, pBrec, pRefN, pShadowRays, pLrec, pSampleIdx;
    /* first-round step budgets of the round-based traversal; adapted from the share of rays the last batch parked */
    int capClosest = 26, capAny = 20;
    DevBuf hits, accum, refN, bflags, aflags, origins, bsdfRays, bsdfHits, brec0, shadowRays, occluded, lrec, film, counters, apiRays, apiHits;
    std::vector<hipEvent_t> events;
    std::vector<hipEvent_t> doneEvents;   /* one per batch in flight: recorded after the batch's splat (progress reports) */
    kkd_progress_fn progressFn = nullptr; void *progressUser = nullptr;
    kkd_flush_fn flushFn = nullptr; void *flushUser = nullptr; std::atomic<int> flushReq{0};
    float *flushHost = nullptr; size_t flushHostBytes = 0;     /* pinned copy of the film handed to flushFn */
    /* set up by kkd_create so that kkd_trace_device_noalloc never allocates: this device's scratch array, two timing
     * events, a pinned host copy of the counter shards */
    /* tile sharding: this context's tiles (Morton round-robin), cached for the last (grid, index, count) */
    DevBuf haltonTables;       /* halton sampler: primes, offsets, Faure permutations (uploaded by the first render that uses it) */
    DevBuf sobolTables; uint32_t sobolM = 0xFFFFFFFFu;   /* sobol sampler: generator matrices + the pixel look-up of resolution 2^sobolM */
    DevBuf tileList; std::vector<uint32_t> tileListHost; uint32_t tileKey[4] = { 0, 0, 0, 0 };
    void *noallocScratch = nullptr;
    hipEvent_t traceEv[2] = { nullptr, nullptr };
    DeviceCounters *hostShards = nullptr;
};

#define KKD_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return KKD_ERR_DEVICE; } } while (0)

static int fail(kkd_ctx *ctx, int code, const std::string &msg) { ctx->err = msg; return code; }

template <typename T> static int upload(kkd_ctx *ctx, DevBuf &buf, const T *src, size_t count) {
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = buf.ensure(bytes);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE; }
    if (count) KKD_HIP(ctx, hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return KKD_OK;
}

extern "C" {

int kkd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

kkd_ctx *kkd_create(const kkd_config *cfg) {
    if (!cfg) { g_globalError = "kkd_create: config is NULL"; return nullptr; }
    int n = kkd_device_count();
    if (n <= 0) { g_globalError = "kkd_create: no HIP device available (libkkdirect has no CPU fallback)"; return nullptr; }
    if (cfg->device < 0 || cfg->device >= n) { g_globalError = "kkd_create: device ordinal out of range"; return nullptr; }
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { g_globalError = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
    kkd_ctx *ctx = new kkd_ctx();
    ctx->device = cfg->device; ctx->cfg = *cfg;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_globalError = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete ctx; return nullptr; }
    e = scratchForDevice(cfg->device, &ctx->noallocScratch);
    if (e == hipSuccess) e = hipEventCreate(&ctx->traceEv[0]);
    if (e == hipSuccess) e = hipEventCreate(&ctx->traceEv[1]);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->hostShards, kCounterShards * sizeof(DeviceCounters), hipHostMallocDefault);
    if (e != hipSuccess) { g_globalError = std::string("kkd_create: ") + hipGetErrorString(e); kkd_destroy(ctx); return nullptr; }
    g_globalError = "";
    return ctx;
}

int kkd_probe_rough_transmittance(int distribution, float alpha, float eta, float *trans100, float *diffuse_transmittance) {
    if (!trans100 || !diffuse_transmittance || (distribution != KKD_DISTR_BECKMANN && distribution != KKD_DISTR_GGX) || !(alpha > 0.f) || !(eta > 0.f))
        return KKD_ERR_INVALID;
    roughTransmittanceTable(distribution, alpha, eta, trans100, diffuse_transmittance);
    return KKD_OK;
}

int kkd_probe_ldsampler(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t array_n, uint32_t set_size, uint32_t count, float *out) {
    if (!out || !set_size || (set_size & (set_size - 1u)) || kind < 0 || kind > 2 || (kind == 2 && !array_n)) return KKD_ERR_INVALID;
    for (uint32_t j = 0; j < count; ++j) {
        if (kind == 0) { const f2 v = ldSample2D(seed, pixel, 2u * index, j, set_size); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = ldSample1D(seed, pixel, 2u * index + 1u, j, set_size);
        else for (uint32_t e = 0; e < array_n; ++e) {
            const f2 v = ldSample2D(seed, pixel, 0x10000u + index, j * array_n + e, set_size * array_n);
            out[2 * ((size_t)j * array_n + e)] = v.x; out[2 * ((size_t)j * array_n + e) + 1] = v.y;
        }
    }
    return KKD_OK;
}

int kkd_probe_stratified(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t array_n, uint32_t resolution, uint32_t flat_dim, uint32_t count, float *out) {
    if (!out || !resolution || resolution > 4096u || (flat_dim & 1u) || kind < 0 || kind > 2 || (kind == 2 && !array_n)) return KKD_ERR_INVALID;
    const uint32_t set = resolution * resolution;
    for (uint32_t j = 0; j < count; ++j) {
        const f2 jit = sample2D(seed, pixel, j, flat_dim);
        if (kind == 0) { const f2 v = stratSample2D(seed, pixel, 2u * index, j, resolution, jit); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = stratSample1D(seed, pixel, 2u * index + 1u, j, resolution, jit.x);
        else for (uint32_t e = 0; e < array_n; ++e) {
            const f2 v = latinHypercube2D(seed, pixel, 0x10000u + index, j * array_n + e, set * array_n, sample2D(seed, pixel, j, flat_dim + 2u * e));
            out[2 * ((size_t)j * array_n + e)] = v.x; out[2 * ((size_t)j * array_n + e) + 1] = v.y;
        }
    }
    return KKD_OK;
}

int kkd_probe_halton(uint32_t res_x, uint32_t res_y, int32_t x, int32_t y, int faure, int kind, uint32_t dim, uint32_t array_n, uint32_t n_arrays, uint32_t count, float *out) {
    if (!out || !res_x || !res_y || kind < 0 || kind > 2 || (kind == 2 && (!array_n || dim >= n_arrays)) || (kind != 2 && dim + (kind == 0 ? 1u : 0u) >= kHaltonDims)) return KKD_ERR_INVALID;
    static const std::vector<uint32_t> tables = [] { std::vector<uint32_t> t; haltonTables(t); return t; }();
    HaltonSetup h;
    if (n_arrays == 0xFFFFFFFFu) { if (kind == 2 || !array_n) return KKD_ERR_INVALID; hammersleySetup((int)res_x, (int)res_y, array_n, faure != 0, tables.data(), h); }
    else haltonSetup((int)res_x, (int)res_y, n_arrays, faure != 0, tables.data(), h);
    for (uint32_t j = 0; j < count; ++j) {
        if (kind == 0) { const f2 v = haltonSample2D(h, x, y, j, haltonSkipArrays(h, dim)); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = haltonSample1D(h, x, y, j, haltonSkipArrays(h, dim));
        else for (uint32_t e = 0; e < array_n; ++e) {
            const f2 v = haltonArray2D(h, x, y, dim, (uint64_t)j * array_n + e);
            out[2 * ((size_t)j * array_n + e)] = v.x; out[2 * ((size_t)j * array_n + e) + 1] = v.y;
        }
    }
    return KKD_OK;
}

int kkd_probe_sobol(uint32_t res_x, uint32_t res_y, int32_t x, int32_t y, int kind, uint32_t dim, uint32_t array_n, uint32_t n_arrays, uint32_t count, float *out) {
    if (!out || !res_x || !res_y || kind < 0 || kind > 2 || (kind == 2 && (!array_n || dim >= n_arrays)) || (kind != 2 && dim + (kind == 0 ? 1u : 0u) >= kSobolDimsMax)) return KKD_ERR_INVALID;
    SobolSetup s; sobolSetup((int)res_x, (int)res_y, n_arrays, nullptr, s);
    std::vector<uint32_t> words;
    if (!sobolTables(s.m, words)) return KKD_ERR_UNSUPPORTED;
    s.tables = words.data();
    for (uint32_t j = 0; j < count; ++j) {
        if (kind == 0) { const f2 v = sobolSample2D(s, x, y, j, sobolSkipArrays(s, dim)); out[2 * j] = v.x; out[2 * j + 1] = v.y; }
        else if (kind == 1) out[j] = sobolSample1D(s, x, y, j, sobolSkipArrays(s, dim));
        else for (uint32_t e = 0; e < array_n; ++e) {
            const f2 v = sobolArray2D(s, x, y, dim, (uint64_t)j * array_n + e);
            out[2 * ((size_t)j * array_n + e)] = v.x; out[2 * ((size_t)j * array_n + e) + 1] = v.y;
        }
    }
    return KKD_OK;
}

int kkd_probe_sobol_raw(uint64_t index, uint32_t dim, uint32_t m, uint32_t frame, uint32_t px, uint32_t py, uint32_t *sample_bits, uint64_t *look_up_index, uint32_t *crc32) {
    if (dim >= kSobolDimsMax || m > 20 || (m && (px >> m || py >> m))) return KKD_ERR_INVALID;
    const std::vector<uint32_t> &mat = sobolMatrices();
    if (crc32) *crc32 = crc32Words(mat.data(), mat.size());
    if (sample_bits) { uint32_t r = 0; for (const uint32_t *c = mat.data() + (size_t)dim * kSobolColumns; index; index >>= 1, ++c) if (index & 1u) r ^= *c; *sample_bits = r; }
    if (look_up_index) {
        if (!m) return KKD_ERR_INVALID;
        std::vector<uint32_t> words;
        if (!sobolTables(m, words)) return KKD_ERR_UNSUPPORTED;
        SobolSetup s; s.m = m; s.resolution = (float)(1u << m); s.arrayEnd = 5; s.pad = 0; s.tables = words.data();
        *look_up_index = sobolLookUp(s, frame, px, py);
    }
    return KKD_OK;
}

int kkd_probe_faure_permutation(uint32_t base, uint16_t *out) {
    if (!out || base < 2) return KKD_ERR_INVALID;
    static const std::vector<uint32_t> tables = [] { std::vector<uint32_t> t; haltonTables(t); return t; }();
    for (uint32_t d = 0; d < kHaltonDims; ++d) if (tables[d] == base) {
        std::memcpy(out, reinterpret_cast<const uint16_t *>(tables.data() + 2 * kHaltonDims) + tables[kHaltonDims + d], base * sizeof(uint16_t));
        return KKD_OK;
    }
    return KKD_ERR_INVALID;     /* only the permutations of the prime bases are kept (faure.cpp:57-63) */
}

int kkd_probe_rfilter(int rfilter_type, float radius, const float *param2, float *values32, float *scale_factor, int *border) {
    if (!param2 || !values32 || !scale_factor || !border || rfilter_type < 0 || rfilter_type > KKD_RFILTER_LANCZOS || !(radius > 0.f)) return KKD_ERR_INVALID;
    DFilter f;
    configureFilter(rfilter_type, radius, param2, f);
    std::memcpy(values32, f.values, sizeof(f.values)); *scale_factor = f.scaleFactor; *border = f.border;
    return KKD_OK;
}

const char *kkd_last_error(kkd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_globalError.c_str(); }

void kkd_request_cancel(kkd_ctx *ctx) { if (ctx) ctx->cancel.store(1); }

int kkd_set_progress_callback(kkd_ctx *ctx, kkd_progress_fn fn, void *user) {
    if (!ctx) return KKD_ERR_INVALID;
    ctx->progressFn = fn; ctx->progressUser = user;
    return KKD_OK;
}

int kkd_set_flush_callback(kkd_ctx *ctx, kkd_flush_fn fn, void *user) {
    if (!ctx) return KKD_ERR_INVALID;
    ctx->flushFn = fn; ctx->flushUser = user;
    return KKD_OK;
}

void kkd_request_flush(kkd_ctx *ctx) { if (ctx) ctx->flushReq.store(1); }

void kkd_destroy(kkd_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (hipStream_t s : ctx->laneStreams) if (s) hipStreamSynchronize(s);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    DevBuf *all[] = { &ctx->nodes, &ctx->wideNodes, &ctx->tris, &ctx->projTris, &ctx->positions, &ctx->normals, &ctx->texcoords, &ctx->indices, &ctx->shapes, &ctx->materials,
                      &ctx->emitters, &ctx->emitterCdf, &ctx->areaCdf, &ctx->rtTables, &ctx->envTexels, &ctx->envCdfRows, &ctx->envCdfCols,
                      &ctx->envRowWeights, &ctx->contRecs, &ctx->contStacks, &ctx->contCount, &ctx->hits, &ctx->accum, &ctx->refN, &ctx->bflags, &ctx->aflags, &ctx->origins, &ctx->bsdfRays, &ctx->bsdfHits, &ctx->brec0, &ctx->shadowRays,
                      &ctx->occluded, &ctx->lrec, &ctx->film, &ctx->counters, &ctx->apiRays, &ctx->apiHits,
                      &ctx->envGuideRows, &ctx->envGuideCols, &ctx->envRecRows, &ctx->envRecCols, &ctx->emitterGuide, &ctx->areaGuide, &ctx->pathAlive, &ctx->pBsdfRays, &ctx->pBrec, &ctx->pRefN, &ctx->pShadowRays, &ctx->pLrec, &ctx->pSampleIdx, &ctx->tileList, &ctx->envEwaLut, &ctx->haltonTables, &ctx->sobolTables };
    for (DevBuf *b : all) b->release();
    for (hipEvent_t ev : ctx->events) hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->doneEvents) hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->traceEv) if (ev) hipEventDestroy(ev);
    if (ctx->hostShards) hipHostFree(ctx->hostShards);
    if (ctx->flushHost) hipHostFree(ctx->flushHost);
    for (hipStream_t s : ctx->laneStreams) if (s) hipStreamDestroy(s);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

/* host half of kkd_scene_upload: validate + flatten + build (once per scene, whatever the number of devices) */
static int buildHostScene(kkd_ctx *ctx, const kkd_scene_desc *sd, HostScene &hs) {
    int code = KKD_ERR_INVALID;
    const bool gpuBuild = (ctx->cfg.flags & KKD_FLAG_GPU_BUILD) != 0;
    if (!flattenScene(sd, hs, ctx->err, &code, !gpuBuild)) return code;
    if (gpuBuild) return KKD_OK;          /* the tree is built by uploadHostScene, on the device */
    /* the kernels index their stack without a bound check: the builder's worst-case occupancy must fit */
    if (hs.bvh.info.maxStack > kMaxTraversalStack) return fail(ctx, KKD_ERR_UNSUPPORTED, "acceleration structure deeper than the traversal stack");
    if (hs.bvh.nodes.size() * sizeof(Node4) > kMaxAccelBytes || hs.bvh.tris.size() * sizeof(LeafTri) > kMaxAccelBytes)
        return fail(ctx, KKD_ERR_UNSUPPORTED, "acceleration structure larger than 4 GiB per array (more than 89 M triangles)");
    for (LeafTri &t : hs.bvh.tris) { const int32_t e = hs.shapes[t.shape].emitter; t.emitter = e >= 0 ? (uint32_t)e + 1u : 0u; }
    return KKD_OK;
}

/* device half: copy the flattened scene `hs` (this context's own, or another context's: multi-device renders build once) */
static int uploadHostScene(kkd_ctx *ctx, const HostScene &hs, const kkd_scene_desc *sd) {
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
#define UP(buf, ptr, n) if ((rc = upload(ctx, ctx->buf, ptr, n)) != KKD_OK) return rc;
    const bool gpuBuild = (ctx->cfg.flags & KKD_FLAG_GPU_BUILD) != 0;
    const bool projTest = (ctx->cfg.flags & KKD_FLAG_WATERTIGHT_TEST) == 0;   /* the reference's leaf test unless the caller asks for the watertight one */
    std::vector<ProjTri> pt;   /* read by an asynchronous copy: lives until the stream has been synchronised below */
    if (!gpuBuild) {
        UP(nodes, hs.bvh.nodes.data(), hs.bvh.nodes.size())
        UP(wideNodes, hs.bvh.wide.data(), hs.bvh.wide.size())
        UP(tris, hs.bvh.tris.data(), hs.bvh.tris.size())
        if (projTest) {
            makeProjTris(hs.bvh.tris, pt);
            UP(projTris, pt.data(), pt.size())
        }
    }
    UP(positions, sd->positions, 3 * (size_t)sd->n_verts)
    UP(normals, sd->normals, sd->normals ? 3 * (size_t)sd->n_verts : 0)
    UP(texcoords, sd->texcoords, sd->texcoords ? 2 * (size_t)sd->n_verts : 0)
    UP(indices, sd->indices, 3 * (size_t)sd->n_tris)
    UP(shapes, hs.shapes.data(), hs.shapes.size())
    UP(materials, hs.materials.data(), hs.materials.size())
    UP(emitters, hs.emitters.data(), hs.emitters.size())
    UP(emitterCdf, hs.emitterCdf.data(), hs.emitterCdf.size())
    UP(emitterGuide, hs.emitterGuide.data(), hs.emitterGuide.size())
    UP(areaGuide, hs.areaGuide.data(), hs.areaGuide.size())
    UP(areaCdf, hs.areaCdf.data(), hs.areaCdf.size())
    UP(rtTables, hs.rtTables.data(), hs.rtTables.size())
    UP(envTexels, hs.envTexels.data(), hs.envTexels.size())
    UP(envCdfRows, hs.envCdfRows.data(), hs.envCdfRows.size())
    UP(envCdfCols, hs.envCdfCols.data(), hs.envCdfCols.size())
    UP(envRowWeights, hs.envRowWeights.data(), hs.envRowWeights.size())
    UP(envGuideRows, hs.envGuideRows.data(), hs.envGuideRows.size())
    UP(envGuideCols, hs.envGuideCols.data(), hs.envGuideCols.size())
    UP(envRecRows, hs.envRecRows.data(), hs.envRecRows.size())
    UP(envRecCols, hs.envRecCols.data(), hs.envRecCols.size())
    UP(envEwaLut, hs.envEwaLut.data(), hs.envEwaLut.size())
#undef UP
    KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BVH4::Info accel = hs.bvh.info;
    if (gpuBuild) {
        /* KKD_FLAG_GPU_BUILD: the tree is built from the vertex and index arrays that are on the device already */
        const uint32_t nTris = (uint32_t)sd->n_tris;
        if ((uint64_t)nTris * sizeof(Node4) > kMaxAccelBytes) return fail(ctx, KKD_ERR_UNSUPPORTED, "acceleration structure larger than 4 GiB per array");
        hipError_t e = ctx->nodes.ensure((size_t)nTris * sizeof(Node4));
        if (e == hipSuccess) e = ctx->wideNodes.ensure((size_t)nTris * sizeof(WideNode));
        if (e == hipSuccess) e = ctx->tris.ensure((size_t)nTris * sizeof(LeafTri));
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
        GpuBuildResult res{}; std::string berr;
        e = gpuBuildBvh(ctx->stream, (const float *)ctx->positions.p, (const uint32_t *)ctx->indices.p, nTris, (const DShape *)ctx->shapes.p, (uint32_t)hs.shapes.size(),
                        (Node4 *)ctx->nodes.p, (WideNode *)ctx->wideNodes.p, (LeafTri *)ctx->tris.p, nTris, &res, berr);
        if (e != hipSuccess) return fail(ctx, e == hipErrorInvalidValue ? KKD_ERR_INVALID : (e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE), "device build: " + berr);
        if (res.maxStack > kMaxTraversalStack) return fail(ctx, KKD_ERR_UNSUPPORTED, "acceleration structure deeper than the traversal stack");
        accel = BVH4::Info(); accel.nNodes = res.nNodes; accel.nLeafTris = nTris; accel.maxDepth = res.maxDepth; accel.maxStack = res.maxStack;
        if (projTest) {
            /* the TriAccel table follows the leaf order: one more kernel over the leaf records (same arithmetic as makeProjTris) */
            e = ctx->projTris.ensure((size_t)nTris * sizeof(ProjTri));
            if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
            gpuMakeProjTris(ctx->stream, (const LeafTri *)ctx->tris.p, nTris, (ProjTri *)ctx->projTris.p);
            KKD_HIP(ctx, hipGetLastError());
            KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    DScene &d = ctx->ds;
    std::memset(&d, 0, sizeof(d));
    d.nodes = (const Node4 *)ctx->nodes.p; d.tris = (const LeafTri *)ctx->tris.p; d.wideNodes = (const WideNode *)ctx->wideNodes.p;
    d.projTris = projTest ? (const ProjTri *)ctx->projTris.p : nullptr;
    d.positions = (const float *)ctx->positions.p; d.normals = (const float *)ctx->normals.p; d.texcoords = (const float *)ctx->texcoords.p;
    d.indices = (const uint32_t *)ctx->indices.p;
    d.shapes = (const DShape *)ctx->shapes.p; d.materials = (const DMaterial *)ctx->materials.p; d.emitters = (const DEmitter *)ctx->emitters.p;
    d.emitterGuide = (const uint32_t *)ctx->emitterGuide.p; d.areaGuide = (const uint32_t *)ctx->areaGuide.p;
    d.emitterCdf = (const float *)ctx->emitterCdf.p; d.areaCdf = (const float *)ctx->areaCdf.p; d.rtTables = (const float *)ctx->rtTables.p;
    d.nShapes = (uint32_t)hs.shapes.size(); d.nMaterials = (uint32_t)hs.materials.size(); d.nEmitters = (uint32_t)hs.emitters.size();
    d.envEmitter = hs.envEmitter; d.emitterNormalization = hs.emitterNormalization;
    d.env = hs.env;
    d.env.texels = (const uint32_t *)ctx->envTexels.p; d.env.cdfRows = (const float *)ctx->envCdfRows.p;
    d.env.cdfCols = (const float *)ctx->envCdfCols.p; d.env.rowWeights = (const float *)ctx->envRowWeights.p;
    d.env.guideRows = (const uint32_t *)ctx->envGuideRows.p; d.env.guideCols = (const uint32_t *)ctx->envGuideCols.p;
    d.env.recRows = (const uint32_t *)ctx->envRecRows.p; d.env.recCols = (const uint32_t *)ctx->envRecCols.p;
    d.env.ewaLut = (const float *)ctx->envEwaLut.p;
    d.sensor = hs.sensor; d.filter = hs.filter;
    ctx->accelInfo = accel;
    ctx->tileKey[3] = 0;            /* a new film / tile grid: the cached tile list is stale */
    ctx->haveScene = true;
    ctx->err.clear();
    return KKD_OK;
}

int kkd_scene_upload(kkd_ctx *ctx, const kkd_scene_desc *sd) {
    if (!ctx) return KKD_ERR_INVALID;
    ctx->haveScene = false;
    HostScene hs;                    /* host copies are only needed for the upload */
    int rc = buildHostScene(ctx, sd, hs);
    if (rc == KKD_OK) rc = uploadHostScene(ctx, hs, sd);
    return rc;
}

int kkd_accel_info(kkd_ctx *ctx, uint32_t *node_bytes, uint32_t *tri_bytes, uint64_t *n_nodes, uint64_t *n_leaf_tris) {
    if (!ctx) return KKD_ERR_INVALID;
    if (!ctx->haveScene) return fail(ctx, KKD_ERR_INVALID, "no scene uploaded");
    if (node_bytes) *node_bytes = (uint32_t)sizeof(Node4);
    if (tri_bytes) *tri_bytes = (uint32_t)sizeof(LeafTri);
    if (n_nodes) *n_nodes = ctx->accelInfo.nNodes;
    if (n_leaf_tris) *n_leaf_tris = ctx->accelInfo.nLeafTris;
    return KKD_OK;
}

static int developChannels(int fmt) {
    switch (fmt) {
        case KKD_PIXEL_LUMINANCE: return 1;
        case KKD_PIXEL_LUMINANCE_ALPHA: return 2;
        case KKD_PIXEL_RGB: case KKD_PIXEL_XYZ: return 3;
        case KKD_PIXEL_RGBA: case KKD_PIXEL_XYZA: return 4;
        default: return 0;
    }
}

int kkd_develop_device(kkd_ctx *ctx, const void *dev_film, uint32_t width, uint32_t height, int32_t pixel_format, void *dev_out) {
    if (!ctx) return KKD_ERR_INVALID;
    if (!developChannels(pixel_format)) return fail(ctx, KKD_ERR_INVALID, "kkd_develop: unknown pixel format");
    if ((!dev_film || !dev_out) && (uint64_t)width * height != 0) return fail(ctx, KKD_ERR_INVALID, "kkd_develop: null buffer");
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    launchDevelop(ctx->stream, (const float *)dev_film, (uint64_t)width * height, pixel_format, (float *)dev_out);
    KKD_HIP(ctx, hipGetLastError());
    KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KKD_OK;
}

int kkd_develop(kkd_ctx *ctx, const float *film, uint32_t width, uint32_t height, int32_t pixel_format, float *out) {
    if (!ctx) return KKD_ERR_INVALID;
    const int ch = developChannels(pixel_format);
    if (!ch) return fail(ctx, KKD_ERR_INVALID, "kkd_develop: unknown pixel format");
    const uint64_t n = (uint64_t)width * height;
    if (n == 0) return KKD_OK;
    if (!film || !out) return fail(ctx, KKD_ERR_INVALID, "kkd_develop: null buffer");
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = ctx->film.ensure(n * 5 * sizeof(float));
    if (e == hipSuccess) e = ctx->apiHits.ensure(n * ch * sizeof(float));
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE; }
    KKD_HIP(ctx, hipMemcpyAsync(ctx->film.p, film, n * 5 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    launchDevelop(ctx->stream, (const float *)ctx->film.p, n, pixel_format, (float *)ctx->apiHits.p);
    KKD_HIP(ctx, hipGetLastError());
    KKD_HIP(ctx, hipMemcpyAsync(out, ctx->apiHits.p, n * ch * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KKD_OK;
}

static DeviceCounters sumShards(const DeviceCounters *shards, size_t n) {
    DeviceCounters c; std::memset(&c, 0, sizeof(c));
    for (size_t k = 0; k < n; ++k) {
        const DeviceCounters &s = shards[k];
        c.samples += s.samples;
        for (int k = 0; k < 3; ++k) c.parked[k] += s.parked[k];
        for (int k = 0; k < 3; ++k) { c.rays[k] += s.rays[k]; c.nodes[k] += s.nodes[k]; c.tris[k] += s.tris[k]; }
    }
    return c;
}

static void fillCounterStats(kkd_stats *stats, const DeviceCounters &c, bool counted) {
    stats->samples = c.samples;
    stats->rays_closest = c.rays[KKD_RAYS_PRIMARY] + c.rays[KKD_RAYS_BSDF]; stats->rays_shadow = c.rays[KKD_RAYS_SHADOW];
    double bytes = 0.0;
    for (int k = 0; k < 3; ++k) {
        stats->class_rays[k] = c.rays[k]; stats->class_nodes[k] = c.nodes[k]; stats->class_tris[k] = c.tris[k];
        stats->nodes_visited += c.nodes[k]; stats->tris_tested += c.tris[k];
        /* B_ray of SURVEY 8d: ray read (none for primary rays, they are generated in registers) + result write
         * (16-byte hit record, 4-byte occlusion word for shadow rays) + visited nodes + tested triangles */
        const double io = k == KKD_RAYS_PRIMARY ? 16.0 : (k == KKD_RAYS_BSDF ? 48.0 : 36.0);
        bytes += (double)c.rays[k] * io + (double)c.nodes[k] * sizeof(Node4) + (double)c.tris[k] * sizeof(LeafTri);
    }
    stats->algorithmic_bytes = counted ? bytes : 0.0;
}

/* ---- fixed ray sets ---- */
/* noAlloc: the variant behind kkd_trace_device_noalloc -- counters in this device's static scratch array, the events and
 * the host copy of the shards created by kkd_create: no hipMalloc / hipFree / hipEventCreate / host allocation in here. */
static int traceImpl(kkd_ctx *ctx, const void *devRays, uint64_t n, int anyHit, void *devOut, bool count, kkd_stats *stats, bool noAlloc = false) {
    if (!ctx->haveScene) return fail(ctx, KKD_ERR_INVALID, "no scene uploaded");
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    if (!noAlloc) KKD_HIP(ctx, ctx->counters.ensure(kCounterShards * sizeof(DeviceCounters)));
    DeviceCounters *ctr = (DeviceCounters *)(noAlloc ? ctx->noallocScratch : ctx->counters.p);
    /* the array keeps the last call's contents: the counters must start at zero */
    KKD_HIP(ctx, hipMemsetAsync(ctr, 0, kCounterShards * sizeof(DeviceCounters), ctx->stream));
    const hipEvent_t e0 = ctx->traceEv[0], e1 = ctx->traceEv[1];
    KKD_HIP(ctx, hipEventRecord(e0, ctx->stream));
    const uint64_t chunk = 1ull << 30;   /* keep grid sizes within 32 bits */
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        launchTrace(ctx->stream, ctx->ds, (const DRay *)devRays + off, m, anyHit != 0, count, 1, (DHit *)devOut + off, ctr,
                    anyHit ? KKD_RAYS_SHADOW : KKD_RAYS_BSDF);
    }
    KKD_HIP(ctx, hipGetLastError());
    KKD_HIP(ctx, hipEventRecord(e1, ctx->stream));
    KKD_HIP(ctx, hipMemcpyAsync(ctx->hostShards, ctr, kCounterShards * sizeof(DeviceCounters), hipMemcpyDeviceToHost, ctx->stream));
    KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const DeviceCounters c = sumShards(ctx->hostShards, kCounterShards);
    float ms = 0.f;
    KKD_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        fillCounterStats(stats, c, count);
        /* the C-ABI trace entry points read 32-byte rays and write 16-byte hits for both ray kinds */
        if (count) stats->algorithmic_bytes += 12.0 * (double)c.rays[KKD_RAYS_SHADOW];
        stats->seconds = stats->seconds_trace = ms * 1e-3;
        const int stage = anyHit ? KKD_STAGE_SHADOW : KKD_STAGE_BSDF;
        stats->stage_seconds[stage] = ms * 1e-3; stats->stage_launches[stage] = (n + chunk - 1) / chunk;
    }
    return KKD_OK;
}

int kkd_trace_device(kkd_ctx *ctx, const void *dev_rays, uint64_t n, int any_hit, void *dev_out, kkd_stats *stats) {
    if (!ctx) return KKD_ERR_INVALID;
    if (n && (!dev_rays || !dev_out)) return fail(ctx, KKD_ERR_INVALID, "kkd_trace_device: NULL buffer");
    return traceImpl(ctx, dev_rays, n, any_hit, dev_out, false, stats);
}

/* The entry point that allocates nothing (include/kkdirect.h): kkd_trace_device with the counter shards in the static
 * device array above.  One launch, so n is bounded by the largest grid. */
int kkd_trace_device_noalloc(kkd_ctx *ctx, const void *dev_rays, uint64_t n, int any_hit, void *dev_out, kkd_stats *stats) {
    if (!ctx) return KKD_ERR_INVALID;
    if (n > KKD_NOALLOC_MAX_RAYS) return fail(ctx, KKD_ERR_UNSUPPORTED, "kkd_trace_device_noalloc: more than KKD_NOALLOC_MAX_RAYS (2^30) rays in one call");
    if (n && (!dev_rays || !dev_out)) return fail(ctx, KKD_ERR_INVALID, "kkd_trace_device_noalloc: NULL buffer");
    if (!ctx->noallocScratch) return fail(ctx, KKD_ERR_DEVICE, "kkd_trace_device_noalloc: no scratch array on this device");
    return traceImpl(ctx, dev_rays, n, any_hit, dev_out, false, stats, true);
}

static int traceHost(kkd_ctx *ctx, const float *rays, uint64_t n, int anyHit, kkd_hit *out, bool count, kkd_stats *stats) {
    if (!ctx) return KKD_ERR_INVALID;
    if (n && (!rays || !out)) return fail(ctx, KKD_ERR_INVALID, "kkd_trace: NULL buffer");
    if (!ctx->haveScene) return fail(ctx, KKD_ERR_INVALID, "no scene uploaded");
    if (n == 0) { if (stats) std::memset(stats, 0, sizeof(*stats)); return KKD_OK; }
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = ctx->apiRays.ensure(n * sizeof(DRay));
    if (e == hipSuccess) e = ctx->apiHits.ensure(n * sizeof(DHit));
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    KKD_HIP(ctx, hipMemcpyAsync(ctx->apiRays.p, rays, n * sizeof(DRay), hipMemcpyHostToDevice, ctx->stream));
    int rc = traceImpl(ctx, ctx->apiRays.p, n, anyHit, ctx->apiHits.p, count, stats);
    if (rc != KKD_OK) return rc;
    KKD_HIP(ctx, hipMemcpy(out, ctx->apiHits.p, n * sizeof(DHit), hipMemcpyDeviceToHost));
    return KKD_OK;
}

int kkd_trace(kkd_ctx *ctx, const float *rays, uint64_t n, int any_hit, kkd_hit *out) { return traceHost(ctx, rays, n, any_hit, out, false, nullptr); }
int kkd_trace_counted(kkd_ctx *ctx, const float *rays, uint64_t n, int any_hit, kkd_hit *out, kkd_stats *stats) { return traceHost(ctx, rays, n, any_hit, out, true, stats); }

/* ---- render ---- */
static uint32_t pow2ceil(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

/* Tile -> rank assignment of the tile-sharded mode: the tiles of the nbx x nby grid are walked along their
 * Morton curve and dealt out round-robin, tile k of that walk to rank k % count.  Every rank's tiles are spread evenly over
 * the image (lights and dense geometry cluster spatially), consecutive tiles of a rank stay close to each other, and N = 2
 * ... 8 never degenerate into stripes the way "row-major index % N" does when N divides the row length.  Replaces the
 * work queue of BlockedImageProcess (reference src/librender/imageproc.cpp:28-78), which balances dynamically. */
static uint32_t part1by1(uint32_t x) { x &= 0xffffu; x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; return (x | (x << 1)) & 0x55555555u; }
/* out: the tiles of rank `index`; rankOf (optional): the rank of every tile of the grid, appended to out (the film merge
 * uses it to send a pixel that several of a rank's tile regions cover only once) */
static void tilesOfRank(uint32_t nbx, uint32_t nby, uint32_t index, uint32_t count, std::vector<uint32_t> &out, bool appendRankOf = false) {
    std::vector<std::pair<uint32_t, uint32_t>> order; order.reserve((size_t)nbx * nby);
    for (uint32_t ty = 0; ty < nby; ++ty) for (uint32_t tx = 0; tx < nbx; ++tx) order.emplace_back(part1by1(tx) | (part1by1(ty) << 1), ty * nbx + tx);
    std::sort(order.begin(), order.end());
    out.clear();
    for (size_t k = index; k < order.size(); k += count) out.push_back(order[k].second);
    if (appendRankOf) {
        const size_t base = out.size();
        out.resize(base + order.size());
        for (size_t k = 0; k < order.size(); ++k) out[base + order[k].second] = (uint32_t)(k % count);
    }
}

static int renderImpl(kkd_ctx *ctx, const kkd_render_params *rp, float *devFilm, kkd_stats *stats) {
    if (!ctx->haveScene) return fail(ctx, KKD_ERR_INVALID, "no scene uploaded");
    if (!rp) return fail(ctx, KKD_ERR_INVALID, "render params are NULL");
    if (rp->spp == 0) return fail(ctx, KKD_ERR_INVALID, "spp must be positive");
    /* kk/direct.cpp:165-166 divides by both counts: zero is not a meaningful configuration */
    if (rp->integrator > KKD_INTEGRATOR_PATH) return fail(ctx, KKD_ERR_INVALID, "unknown integrator");
    if (rp->sampler > KKD_SAMPLER_SOBOL) return fail(ctx, KKD_ERR_INVALID, "unknown sampler");
    if ((rp->sampler == KKD_SAMPLER_HAMMERSLEY || rp->sampler == KKD_SAMPLER_HAMMERSLEY_PLAIN) && rp->integrator != KKD_INTEGRATOR_PATH && (rp->bsdf_samples > 1 || rp->light_samples > 1))
        return fail(ctx, KKD_ERR_UNSUPPORTED, "request2DArray(): Not supported for the Hammersley QMC sequence!");   /* hammersley.cpp:299-303 */
    if (rp->sample_count_total && (uint64_t)rp->sample_first + rp->spp > rp->sample_count_total)
        return fail(ctx, KKD_ERR_INVALID, "sample_first + spp exceeds sample_count_total");
    if ((rp->sampler == KKD_SAMPLER_LDSAMPLER || rp->sampler == KKD_SAMPLER_STRATIFIED) && (uint64_t)(rp->sample_count_total ? rp->sample_count_total : rp->spp) * 64u > 0x40000000ull)
        return fail(ctx, KKD_ERR_UNSUPPORTED, "ldsampler / stratified: more than 2^24 samples per pixel");
    const bool pathMode = rp->integrator == KKD_INTEGRATOR_PATH;
    /* MonteCarloIntegrator (librender/integrator.cpp:190-233): maxDepth = -1 or > 0, rrDepth > 0 */
    if (pathMode && (rp->max_depth == 0 || rp->max_depth < -1)) return fail(ctx, KKD_ERR_INVALID, "maxDepth must be -1 (unbounded) or positive");
    if (pathMode && rp->rr_depth <= 0) return fail(ctx, KKD_ERR_INVALID, "rrDepth must be greater than zero");
    if (rp->integrator == KKD_INTEGRATOR_KK && (rp->bsdf_samples == 0 || rp->light_samples == 0))
        return fail(ctx, KKD_ERR_UNSUPPORTED, "bsdfSamples and lightSamples must be >= 1");
    /* the stock plugin only asserts emitterSamples + bsdfSamples > 0 (integrators/direct/direct.cpp:107) */
    if (!pathMode && rp->bsdf_samples + rp->light_samples == 0) return fail(ctx, KKD_ERR_INVALID, "emitterSamples + bsdfSamples must be > 0");
    if (!pathMode && (rp->bsdf_samples > 64 || rp->light_samples > 64)) return fail(ctx, KKD_ERR_UNSUPPORTED, "bsdfSamples / lightSamples above 64");
    const uint32_t bs = rp->block_size ? rp->block_size : 32;                    /* scene.cpp:24 */
    if (bs < 2 || bs > 128) return fail(ctx, KKD_ERR_INVALID, "block size must be in [2, 128]");   /* mitsuba.cpp:142 */
    if ((int)bs < ctx->ds.filter.border) return fail(ctx, KKD_ERR_INVALID, "block size must be >= the filter border");   /* renderproc.cpp:173-174 */
    const uint32_t tileCount = rp->tile_count ? rp->tile_count : 1, tileIndex = rp->tile_count ? rp->tile_index : 0;
    if (tileIndex >= tileCount) return fail(ctx, KKD_ERR_INVALID, "tile_index must be < tile_count");
    KKD_HIP(ctx, hipSetDevice(ctx->device));
    ctx->cancel.store(0); ctx->flushReq.store(0);

    const int W = ctx->ds.sensor.cropW, H = ctx->ds.sensor.cropH;
    const int edge = ctx->ds.sensor.hqEdges ? ctx->ds.filter.border : 0;   /* Film::hasHighQualityEdges (renderproc.cpp:163-171) */
    const uint32_t nbx = (W + 2 * edge + bs - 1) / bs, nby = (H + 2 * edge + bs - 1) / bs, nTiles = nbx * nby;
    if (nbx > 0xffffu || nby > 0xffffu) return fail(ctx, KKD_ERR_UNSUPPORTED, "more than 65535 tiles along one axis");
    const uint32_t nTilesMine = tileIndex < nTiles ? (nTiles - tileIndex + tileCount - 1) / tileCount : 0;
    if (tileCount <= 1 || nTilesMine == 0) ctx->tileListHost.clear();      /* the list describes THIS render: none of an earlier one survives */
    if (tileCount > 1 && nTilesMine > 0) {
        const uint32_t key[4] = { nbx, nby, tileIndex, tileCount };
        if (std::memcmp(key, ctx->tileKey, sizeof(key)) != 0 || !ctx->tileList.p || ctx->tileListHost.size() != nTilesMine) {
            tilesOfRank(nbx, nby, tileIndex, tileCount, ctx->tileListHost, true);      /* device copy: [my tiles | rank of every tile] */
            int rcUp = upload(ctx, ctx->tileList, ctx->tileListHost.data(), ctx->tileListHost.size());
            ctx->tileListHost.resize(nTilesMine);
            if (rcUp != KKD_OK) return rcUp;
            KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            std::memcpy(ctx->tileKey, key, sizeof(key));
        }
    }
    const uint32_t g = std::min(64u, pow2ceil(rp->spp)), chunks = (rp->spp + g - 1) / g;
    uint32_t gShift = 0; while ((1u << gShift) < g) ++gShift;
    const uint64_t total = (uint64_t)nTilesMine * bs * bs * chunks * g;
    /* library default: 64 M camera samples in flight per lane, fewer when every sample carries many BSDF / emitter samples
     * (the ray and record buffers grow with max(bsdfSamples, lightSamples): ~0.2 KB per ray pair and lane, 14 GB per lane at
     * the default).  Every batch ends each of its traversal stages with a round that runs as long as the batch's longest
     * ray, ~0.4 ms with the device nearly idle, so fewer, larger batches are faster: C4 2720 / 2796 / 2841 / 2885 / 2898 / 2903
     * Msamples/s at 8 / 16 / 32 / 64 / 128 / 256 M samples per batch (C2: 3184 / 3227 / 3237 / 3260 / 3284 / 3297) */
    const uint64_t perSample = std::max<uint64_t>(1, std::max(pathMode ? 1u : rp->bsdf_samples, pathMode ? 1u : rp->light_samples));
    static const int envLanesDefault = [] { const char *e = getenv("KKD_STREAMS"); return envInt(e, 2, 1, kMaxLanes); }();
    const uint32_t nB = pathMode ? 1u : rp->bsdf_samples, nL = pathMode ? 1u : rp->light_samples;   /* path.cpp: one of each per vertex */
    /* $KKD_ITER_CAP_CLOSEST / _ANY pin the budgets (0 = one launch, no rounds); else they adapt, see below */
    static const int envCapClosest = [] { const char *e = getenv("KKD_ITER_CAP_CLOSEST"); return envInt(e, -1, 0, 4096); }();
    static const int envCapAny = [] { const char *e = getenv("KKD_ITER_CAP_ANY"); return envInt(e, -1, 0, 4096); }();
    const int capClosest = envCapClosest >= 0 ? envCapClosest : ctx->capClosest, capAny = envCapAny >= 0 ? envCapAny : ctx->capAny;
    /* Batch lanes: consecutive batches alternate between two streams with their own wavefront buffers, so the
     * issue-bound kernels of one batch (sample generation, shading, splat) fill the issue slots the latency-bound
     * traversal kernels of the other leave idle.  Batches only meet in the film and the counters (atomic adds).
     * $KKD_STREAMS=1 switches back to one lane. */
    int envLanes = (ctx->cfg.flags & KKD_FLAG_ONE_LANE) ? 1 : envLanesDefault;
    uint64_t batchCap = ctx->cfg.max_samples_in_flight ? ctx->cfg.max_samples_in_flight : std::max<uint64_t>(1ull << 20, (1ull << 28) / perSample);   /* 256 M rays of a kind per launch: ~42 GB of wavefront buffers per lane at one sample of each kind (round 3: 64 M -> 256 M is +1.2 % on C4, every batch ends each traversal stage with an almost idle device); less when the device cannot give that much, below */
    uint64_t batch = 0, nBatches = 0, contCap = 0; size_t S = 0; int nLanes = 1;
    hipError_t e = hipSuccess;
    auto laneP = [&](DevBuf &b, size_t bytes, int lane) { return (char *)b.p + ((bytes + 255) / 256 * 256) * lane; };
    /* The wavefront buffers are sized by the batch (~0.16 KB per camera sample and lane with the continuation records: 84 GB for two lanes of 256 M samples).
     * On a device that cannot give that much -- a smaller part, or one shared with other contexts or another tenant -- the
     * batch is halved down to 1 M samples, then the second lane is dropped, before the render gives up with KKD_ERR_NOMEM. */
    for (int attempt = 0;; ++attempt) {
        batch = std::max<uint64_t>(256, (batchCap + 255) / 256 * 256);
        batch = std::min<uint64_t>(batch, 1ull << 28);
        /* the rays of one kind of a batch are traced by one launch and numbered with 32 bits (continuation records, RaySource) */
        batch = std::min<uint64_t>(batch, std::max<uint64_t>(256, ((1ull << 31) / std::max<uint64_t>(1, std::max(nB, nL))) / 256 * 256));
        batch = std::min<uint64_t>(batch, std::max<uint64_t>(256, (total + 255) / 256 * 256));
        if ((!ctx->cfg.max_samples_in_flight || attempt > 0) && envLanes >= 2 && total >= (1ull << 21)) {
            /* library default: an even number of equal batches, so both batch lanes work to the end */
            uint64_t nb = (total + batch - 1) / batch;
            nb = std::max<uint64_t>(2, nb + (nb & 1));
            batch = ((total + nb - 1) / nb + 255) / 256 * 256;
        }
        S = (size_t)batch;
        nBatches = (total + batch - 1) / batch;
        nLanes = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min(envLanes, kMaxLanes), nBatches));
        e = hipSuccess;
        for (int l = 1; l < nLanes && e == hipSuccess; ++l)
            if (!ctx->laneStreams[l - 1]) e = hipStreamCreateWithFlags(&ctx->laneStreams[l - 1], hipStreamNonBlocking);
        if (e != hipSuccess) return fail(ctx, KKD_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        auto need = [&](DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.ensure(bytes); };
        auto needL = [&](DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.ensure(((bytes + 255) / 256 * 256) * nLanes); };   /* one copy per lane */
        needL(ctx->hits, S * sizeof(DHit)); needL(ctx->accum, S * 16); needL(ctx->refN, pathMode ? S * 16 : 16); needL(ctx->bflags, pathMode ? 16 : S * nB); needL(ctx->aflags, pathMode ? 16 : S); needL(ctx->origins, pathMode ? 16 : S * 16);
        needL(ctx->bsdfRays, S * nB * sizeof(DRay)); needL(ctx->bsdfHits, S * nB * sizeof(DHit)); needL(ctx->brec0, S * nB * 16);
        /* continuation buffer of the two-launch traversal: half of the largest ray launch may be parked */
        contCap = std::min<uint64_t>(((uint64_t)S * std::max(nB, nL) / 2 / kContShards + 128) * kContShards, 0x7fffffffull);
        needL(ctx->contRecs, contCap * sizeof(ContRec)); needL(ctx->contStacks, contCap * 16 * sizeof(uint32_t)); needL(ctx->contCount, 4 * kContShards * 32 * sizeof(uint32_t));   /* closest: 2 ping-pong counter arrays, any: 2 */
        needL(ctx->shadowRays, S * nL * sizeof(DRay)); needL(ctx->occluded, S * nL * 4); needL(ctx->lrec, S * nL * 16);
        need(ctx->counters, kCounterShards * sizeof(DeviceCounters));
        if (pathMode) {   /* slot counters and the second set of path arrays (compaction ping-pong) */
            needL(ctx->pathAlive, (kAliveShards + 1) * sizeof(uint32_t));
            needL(ctx->pBsdfRays, S * sizeof(DRay)); needL(ctx->pBrec, S * 16); needL(ctx->pRefN, S * 16);
            needL(ctx->pShadowRays, S * sizeof(DRay)); needL(ctx->pLrec, S * 16); needL(ctx->pSampleIdx, 2 * S * sizeof(uint32_t));
        }
        if (!devFilm) need(ctx->film, (size_t)W * H * 5 * sizeof(float));
        if (e == hipSuccess) {
            /* The buffers fit -- but a device left with next to nothing cannot run them: the runtime allocates the kernels'
             * private segments (the spilled part of the traversal stack, ~0.2 GB when the chip is full of waves) when they
             * are dispatched, and a failure THERE aborts the queue and the process.  Keep kMemHeadroom free. */
            size_t freeB = 0, totalB = 0;
            if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB < kMemHeadroom && (batch > (1ull << 20) || envLanes > 1)) e = hipErrorOutOfMemory;
        }
        if (e == hipSuccess) break;
        if (e != hipErrorOutOfMemory) return fail(ctx, KKD_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
        (void)hipGetLastError();
        /* give back what this attempt took (DevBuf::ensure frees before it grows, so the buffers that failed hold nothing) */
        DevBuf *work[] = { &ctx->hits, &ctx->accum, &ctx->refN, &ctx->bflags, &ctx->aflags, &ctx->origins, &ctx->bsdfRays, &ctx->bsdfHits, &ctx->brec0, &ctx->contRecs, &ctx->contStacks, &ctx->contCount,
                           &ctx->shadowRays, &ctx->occluded, &ctx->lrec, &ctx->pathAlive, &ctx->pBsdfRays, &ctx->pBrec, &ctx->pRefN, &ctx->pShadowRays, &ctx->pLrec, &ctx->pSampleIdx };
        for (DevBuf *b : work) b->release();
        if (getenv("KKD_VERBOSE")) fprintf(stderr, "kkd_render: out of device memory with %llu samples in flight on %d lane(s): backing off\n", (unsigned long long)batch, nLanes);
        if (batch > (1ull << 20)) batchCap = std::max<uint64_t>(1ull << 20, batch / 2);
        else if (envLanes > 1) envLanes = 1;
        else return fail(ctx, KKD_ERR_NOMEM, "hipMalloc: out of memory for the wavefront buffers of 1 M samples in flight");
    }
    float *film = devFilm ? devFilm : (float *)ctx->film.p;

    TraceContinuation contL[kMaxLanes]; WaveBuffers wbL[kMaxLanes];
    for (int l = 0; l < nLanes; ++l) {
        TraceContinuation &cont = contL[l]; WaveBuffers &wb = wbL[l];
        cont.recs = (ContRec *)laneP(ctx->contRecs, contCap * sizeof(ContRec), l); cont.stacks = (uint4 *)laneP(ctx->contStacks, contCap * 16 * sizeof(uint32_t), l);
        cont.count = (uint32_t *)laneP(ctx->contCount, 4 * kContShards * 32 * sizeof(uint32_t), l);
        cont.perShard = (uint32_t)(contCap / kContShards); cont.iterCap = 0;
        wb.hits = (DHit *)laneP(ctx->hits, S * sizeof(DHit), l); wb.accum = (float4 *)laneP(ctx->accum, S * 16, l); wb.refN = (float4 *)laneP(ctx->refN, pathMode ? S * 16 : 16, l); wb.bflags = (uint8_t *)laneP(ctx->bflags, pathMode ? 16 : S * nB, l); wb.aflags = (uint8_t *)laneP(ctx->aflags, pathMode ? 16 : S, l); wb.origins = (float4 *)laneP(ctx->origins, pathMode ? 16 : S * 16, l);
        wb.bsdfRays = (DRay *)laneP(ctx->bsdfRays, S * nB * sizeof(DRay), l); wb.bsdfHits = (DHit *)laneP(ctx->bsdfHits, S * nB * sizeof(DHit), l);
        wb.brec0 = (float4 *)laneP(ctx->brec0, S * nB * 16, l);
        wb.shadowRays = (DRay *)laneP(ctx->shadowRays, S * nL * sizeof(DRay), l); wb.occluded = (uint32_t *)laneP(ctx->occluded, S * nL * 4, l);
        wb.lrec = (float4 *)laneP(ctx->lrec, S * nL * 16, l);
        wb.sampleIdx = nullptr;
    }
    /* two rounds: all rays for the adaptive first budget, then the parked ones to the end in persistent waves (kTraceTail).  The
     * step counts are heavy-tailed (C4, BSDF rays: 11 % need more than 32 steps, 1.5 % more than 64, 0.3 % more than 128), so a
     * wave of 64 parked rays runs several times longer than its mean ray unless its idle lanes are refilled.  ($KKD_TRACE_ROUNDS
     * > 2 puts plain resume rounds of $KKD_ITER_CAP_LATER steps in between: the layout before the persistent tail.) */
    static const int capLater = [] { const char *e = getenv("KKD_ITER_CAP_LATER"); return envInt(e, 32, 1, 4096); }();
    static const int rounds = [] { const char *e = getenv("KKD_TRACE_ROUNDS"); return envInt(e, 2, 2, 8); }();
    for (int l = 0; l < nLanes; ++l) { contL[l].iterCapLater = capLater; contL[l].rounds = rounds; }
    DeviceCounters *ctr = (DeviceCounters *)ctx->counters.p;

    BatchParams bp;
    std::memset(&bp, 0, sizeof(bp));
    bp.seed = ctx->cfg.seed; bp.stride = (uint32_t)S;
    bp.g = g; bp.gShift = gShift; bp.chunks = chunks; bp.spp = rp->spp; bp.sampleFirst = rp->sample_first;
    bp.nB = nB; bp.nL = nL; bp.hideEmitters = rp->hide_emitters != 0;
    static const bool wholeRays = getenv("KKD_RAYS_WHOLE") != nullptr;     /* developer switch (A/B runs): 32-byte rays, the layout of round 2 */
    bp.splitRays = (!pathMode && !wholeRays) ? 1u : 0u;
    /* Sampler::getSampleCount(): of the whole render when this call renders a share of every pixel's samples */
    const uint32_t sampleCountTotal = rp->sample_count_total ? rp->sample_count_total : rp->spp;
    bp.diffScale = 1.0f / std::sqrt((float)sampleCountTotal);
    bp.sampler = rp->sampler; bp.samplerDim = rp->sampler_dimension ? rp->sampler_dimension : 4u;   /* ldsampler.cpp:81 */
    bp.samplerSet = pow2ceil(sampleCountTotal);                                                      /* ldsampler.cpp:83-87 */
    if (rp->sampler == KKD_SAMPLER_STRATIFIED) {                                                     /* stratified.cpp:62-79 */
        uint32_t res = 1; while ((uint64_t)res * res < sampleCountTotal) ++res;
        bp.samplerRes = res; bp.samplerSet = res * res;
    }
    if (rp->sampler >= KKD_SAMPLER_HALTON && rp->sampler <= KKD_SAMPLER_HAMMERSLEY_PLAIN) {
        if (!ctx->haltonTables.p) {
            std::vector<uint32_t> words; haltonTables(words);
            const int rcUp = upload(ctx, ctx->haltonTables, words.data(), words.size());
            if (rcUp != KKD_OK) return rcUp;
            KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));      /* `words` is read by an asynchronous copy */
        }
        /* Integrator::configureSampler (librender/integrator.cpp:37-41): the crop size, blocked; arrays as kk/direct.cpp:44-47 request them */
        if (rp->sampler >= KKD_SAMPLER_HAMMERSLEY) hammersleySetup(W, H, sampleCountTotal, rp->sampler == KKD_SAMPLER_HAMMERSLEY, (const uint32_t *)ctx->haltonTables.p, bp.halton);
        else haltonSetup(W, H, pathMode ? 0u : (nB > 1u ? 1u : 0u) + (nL > 1u ? 1u : 0u), rp->sampler == KKD_SAMPLER_HALTON, (const uint32_t *)ctx->haltonTables.p, bp.halton);
    }
    if (rp->sampler == KKD_SAMPLER_SOBOL) {
        /* Integrator::configureSampler (librender/integrator.cpp:37-41): the crop size, blocked; arrays as kk/direct.cpp:44-47 request them */
        sobolSetup(W, H, pathMode ? 0u : (nB > 1u ? 1u : 0u) + (nL > 1u ? 1u : 0u), nullptr, bp.sobol);
        if (!ctx->sobolTables.p || ctx->sobolM != bp.sobol.m) {
            std::vector<uint32_t> words;
            if (!sobolTables(bp.sobol.m, words)) return fail(ctx, KKD_ERR_UNSUPPORTED, "sobol: no pixel look-up for this film resolution");
            const int rcUp = upload(ctx, ctx->sobolTables, words.data(), words.size());
            if (rcUp != KKD_OK) return rcUp;
            KKD_HIP(ctx, hipStreamSynchronize(ctx->stream));      /* `words` is read by an asynchronous copy */
            ctx->sobolM = bp.sobol.m;
        }
        bp.sobol.tables = (const uint32_t *)ctx->sobolTables.p;
    }
    bp.integrator = rp->integrator; bp.strictNormals = rp->strict_normals != 0;
    bp.maxDepth = rp->max_depth; bp.rrDepth = rp->rr_depth; bp.depth = 1;
    bp.chunksShift = 32; bp.pixShift = 32;
    if ((chunks & (chunks - 1)) == 0) { bp.chunksShift = 0; while ((1u << bp.chunksShift) < chunks) ++bp.chunksShift; }
    if ((bs & (bs - 1)) == 0) { bp.pixShift = 0; while ((1u << bp.pixShift) < bs * bs) ++bp.pixShift; }
    bp.edge = edge;
    bp.bs = bs; bp.bsPow2 = (bs & (bs - 1)) == 0; bp.nbx = nbx; bp.nby = nby;
    bp.tileIndex = tileIndex; bp.tileCount = tileCount; bp.nTilesMine = nTilesMine; bp.W = W; bp.H = H;
    bp.tileList = (tileCount > 1 && nTilesMine > 0) ? (const uint32_t *)ctx->tileList.p : nullptr;
#ifdef KKD_ABLATION   /* ablation builds only (scripts/ab*.sh): a release library never reads result-changing switches */
    { const char *dbg = getenv("KKD_DEBUG"); bp.debug = dbg ? (uint32_t)atoi(dbg) : 0u; }
#endif

    const bool count = (ctx->cfg.flags & KKD_FLAG_COUNT_TRAVERSAL) != 0;
    /* one event between consecutive kernels of a batch: 6 per batch, on the stream the kernels run on; events 0 / 1 bracket
     * the render on the context's stream, 2 forks and 3... join the other lanes */
    const bool timeStages = stats != nullptr && nBatches <= 8192 && !pathMode;   /* a path batch is a loop of stages: only the total is timed */
    const size_t evBase = 3 + kMaxLanes;
    const size_t nEvents = evBase + (timeStages ? (size_t)nBatches * 6 : 0);
    while (ctx->events.size() < nEvents) { hipEvent_t ev; KKD_HIP(ctx, hipEventCreate(&ev)); ctx->events.push_back(ev); }

    const hipStream_t st0 = ctx->stream, laneStream[kMaxLanes] = { ctx->stream, ctx->laneStreams[0], ctx->laneStreams[1], ctx->laneStreams[2] };
    KKD_HIP(ctx, hipMemsetAsync(film, 0, (size_t)W * H * 5 * sizeof(float), st0));
    KKD_HIP(ctx, hipMemsetAsync(ctr, 0, kCounterShards * sizeof(DeviceCounters), st0));
    KKD_HIP(ctx, hipEventRecord(ctx->events[0], st0));
    if (nLanes > 1) {   /* the other lanes start after the film and the counters are cleared */
        KKD_HIP(ctx, hipEventRecord(ctx->events[2], st0));
        for (int l = 1; l < nLanes; ++l) KKD_HIP(ctx, hipStreamWaitEvent(laneStream[l], ctx->events[2], 0));
    }
    size_t ev = evBase;
    uint64_t launches[5] = { 0, 0, 0, 0, 0 };
    bool cancelled = false;
    uint64_t batchIndex = 0;
    /* progress: a ring of "batch done" events (a lane runs its batches in order, so at most a few are outstanding);
     * completed ones are polled between batches, never waited for */
    constexpr size_t kDoneRing = 16;
    const bool report = ctx->progressFn != nullptr || ctx->flushFn != nullptr;   /* a flush is answered between batches: same pacing */
    while (report && ctx->doneEvents.size() < kDoneRing) { hipEvent_t e; KKD_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->doneEvents.push_back(e); }
    uint64_t doneCount[kDoneRing] = {}; bool donePending[kDoneRing] = {};
    uint64_t finishedSamples = 0;
    auto pollProgress = [&](bool wait) {
        if (!report) return;
        uint64_t add = 0;
        for (size_t k = 0; k < kDoneRing; ++k) {
            if (!donePending[k]) continue;
            if (wait) hipEventSynchronize(ctx->doneEvents[k]);
            if (hipEventQuery(ctx->doneEvents[k]) == hipSuccess) { donePending[k] = false; add += doneCount[k]; }
        }
        if (add) { finishedSamples += add; if (ctx->progressFn) ctx->progressFn(ctx->progressUser, std::min(finishedSamples, total), total); }
    };
    /* Film::develop during the render (Scene::flush, scene.cpp:458-460): drain what is queued, hand out a copy of the film */
    auto serviceFlush = [&]() -> int {
        const size_t bytes = (size_t)W * H * 5 * sizeof(float);
        if (ctx->flushHostBytes < bytes) {
            if (ctx->flushHost) { hipHostFree(ctx->flushHost); ctx->flushHost = nullptr; ctx->flushHostBytes = 0; }
            if (hipHostMalloc((void **)&ctx->flushHost, bytes, hipHostMallocDefault) != hipSuccess) return fail(ctx, KKD_ERR_NOMEM, "flush: no pinned memory for the film copy");
            ctx->flushHostBytes = bytes;
        }
        for (int l = 0; l < nLanes; ++l) KKD_HIP(ctx, hipStreamSynchronize(laneStream[l]));
        pollProgress(false);
        KKD_HIP(ctx, hipMemcpyAsync(ctx->flushHost, film, bytes, hipMemcpyDeviceToHost, st0));
        KKD_HIP(ctx, hipStreamSynchronize(st0));
        ctx->flushFn(ctx->flushUser, ctx->flushHost, (uint32_t)W, (uint32_t)H, std::min(finishedSamples, total), total);
        return KKD_OK;
    };
    auto batchDone = [&](hipStream_t st, uint64_t index, uint64_t count) {      /* after the batch's splat has been queued */
        if (!report) return;
        const size_t k = (size_t)(index % kDoneRing);
        if (donePending[k]) { hipEventSynchronize(ctx->doneEvents[k]); pollProgress(false); }
        /* with a listener the host stays at most four batches ahead of the device, so that reports arrive while the render runs
         * (queueing a batch takes microseconds, running it milliseconds) */
        if (index >= 4 && donePending[(size_t)((index - 4) % kDoneRing)]) { hipEventSynchronize(ctx->doneEvents[(size_t)((index - 4) % kDoneRing)]); pollProgress(false); }
        hipEventRecord(ctx->doneEvents[k], st); doneCount[k] = count; donePending[k] = true;
    };
    for (uint64_t w0 = 0; w0 < total; w0 += batch, ++batchIndex) {
        if (ctx->cancel.load()) { cancelled = true; break; }
        pollProgress(false);
        if (ctx->flushFn && ctx->flushReq.exchange(0)) { const int rcFlush = serviceFlush(); if (rcFlush != KKD_OK) return rcFlush; }
        const int lane = (int)(batchIndex % (uint64_t)nLanes);
        const hipStream_t st = laneStream[lane];
        const WaveBuffers &wb = wbL[lane]; const TraceContinuation &cont = contL[lane];
        bp.w0 = w0; bp.count = (uint32_t)std::min<uint64_t>(batch, total - w0);
        {   /* w0 is a multiple of 256 and g divides 256: lane group = w0 / g + (i >> gShift) */
            const uint64_t group0 = w0 >> gShift, slot0 = group0 / chunks, pixPerTile = (uint64_t)bs * bs;
            bp.chunk0 = (uint32_t)(group0 % chunks); bp.pin0 = (uint32_t)(slot0 % pixPerTile); bp.tile0 = (uint32_t)(slot0 / pixPerTile);
        }
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
        launchPrimary(st, count, ctx->ds, bp, wb.hits, ctr); launches[KKD_STAGE_PRIMARY]++;
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
        launchShade(st, ctx->ds, bp, wb); launches[KKD_STAGE_SHADE]++;
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
        /* all nB planes are contiguous when count == stride; otherwise trace plane by plane */
        auto trace = [&](const RaySource &rays, uint64_t n, bool any, void *out, int cls) {
            TraceContinuation c = cont; c.iterCap = any ? capAny : capClosest;
            c.count = cont.count + (any ? 2u : 0u) * kContShards * 32u;
            /* stage_launches counts kernel launches: a traversal in rounds is cont.rounds kernels */
            if (c.iterCap > 0) { launchTracePhased(st, ctx->ds, rays, n, any, count, out, ctr, cls, c); launches[any ? KKD_STAGE_SHADOW : KKD_STAGE_BSDF] += std::max(2, c.rounds) - 1; }
            else launchTrace(st, ctx->ds, rays, n, any, count, any ? 2 : 0, out, ctr, cls);
        };
        auto whole = [](const DRay *rays) { return RaySource{ rays, nullptr, 0u }; };
        /* rays of plane k of a ray kind: whole rays, or directions beside the plane of the samples' origins */
        auto planeSrc = [&](const DRay *rays, uint32_t k) {
            if (!bp.splitRays) return whole(rays + (size_t)k * S);
            return RaySource{ reinterpret_cast<const DRay *>(wb.origins), reinterpret_cast<const float4 *>(rays) + (size_t)k * S, (uint32_t)S };
        };
        auto shadowSrc = [&](uint32_t l) { return planeSrc(wb.shadowRays, l); };
        if (bp.count == bp.stride) { trace(planeSrc(wb.bsdfRays, 0), (uint64_t)S * nB, false, wb.bsdfHits, KKD_RAYS_BSDF); launches[KKD_STAGE_BSDF]++; }
        else for (uint32_t b = 0; b < nB; ++b) { trace(planeSrc(wb.bsdfRays, b), bp.count, false, wb.bsdfHits + (size_t)b * S, KKD_RAYS_BSDF); launches[KKD_STAGE_BSDF]++; }
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
        if (bp.count == bp.stride) { trace(shadowSrc(0), (uint64_t)S * nL, true, wb.occluded, KKD_RAYS_SHADOW); launches[KKD_STAGE_SHADOW]++; }
        else for (uint32_t l = 0; l < nL; ++l) { trace(shadowSrc(l), bp.count, true, (uint32_t *)wb.occluded + (size_t)l * S, KKD_RAYS_SHADOW); launches[KKD_STAGE_SHADOW]++; }
        if (pathMode) {
            /* MIPathTracer: vertices at depth 2, 3, ... -- kPathBounce resolves the segments just traced and shades the next
             * vertices, then both ray kinds are traced again.  After every bounce the host reads back how many paths go on:
             * the loop ends when none does (Russian roulette ends every path) or at maxDepth, and once fewer than three
             * quarters of the slots are alive the next bounce writes the survivors densely into the other set of path
             * arrays, so that later launches only cover live paths. */
            uint32_t *counters = (uint32_t *)laneP(ctx->pathAlive, (kAliveShards + 1) * sizeof(uint32_t), lane);
            WaveBuffers sets[2] = { wb, wb };
            sets[1].bsdfRays = (DRay *)laneP(ctx->pBsdfRays, S * sizeof(DRay), lane); sets[1].brec0 = (float4 *)laneP(ctx->pBrec, S * 16, lane);
            sets[1].refN = (float4 *)laneP(ctx->pRefN, S * 16, lane); sets[1].shadowRays = (DRay *)laneP(ctx->pShadowRays, S * sizeof(DRay), lane);
            sets[1].lrec = (float4 *)laneP(ctx->pLrec, S * 16, lane);
            sets[0].sampleIdx = (uint32_t *)laneP(ctx->pSampleIdx, 2 * S * sizeof(uint32_t), lane); sets[1].sampleIdx = sets[0].sampleIdx + S;
            int cur = 0;
            uint32_t n = bp.count;
            bool identity = true, compactNext = false, everCompacted = false;
            static const bool noCompact = getenv("KKD_PATH_NOCOMPACT") != nullptr;   /* developer switch (A/B runs) */
            for (uint32_t depth = 2; (rp->max_depth < 0 || (int)depth < rp->max_depth) && depth < 4096u && n > 0; ++depth) {
                bp.depth = depth;
                KKD_HIP(ctx, hipMemsetAsync(counters, 0, (kAliveShards + 1) * sizeof(uint32_t), st));
                PathLaunch pl; pl.nIn = n; pl.identity = identity; pl.compact = compactNext; pl.terminate = 0; pl.counters = counters;
                launchPathBounce(st, ctx->ds, bp, sets[cur], sets[compactNext ? cur ^ 1 : cur], pl); launches[KKD_STAGE_SHADE]++;
                uint32_t host[kAliveShards + 1];
                KKD_HIP(ctx, hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, st));
                KKD_HIP(ctx, hipStreamSynchronize(st));
                uint64_t alive = 0; for (uint32_t k = 0; k <= kAliveShards; ++k) alive += host[k];
                if (compactNext) { cur ^= 1; n = (uint32_t)alive; everCompacted = true; }
                identity = false;
                if (alive == 0) { n = 0; break; }
                compactNext = !noCompact && alive < (uint64_t)n * 3 / 4;
                trace(whole(sets[cur].bsdfRays), n, false, wb.bsdfHits, KKD_RAYS_BSDF); launches[KKD_STAGE_BSDF]++;
                trace(whole(sets[cur].shadowRays), n, true, wb.occluded, KKD_RAYS_SHADOW); launches[KKD_STAGE_SHADOW]++;
                if (ctx->cancel.load()) break;
            }
            if (!everCompacted) {
                /* never compacted: slot j still holds sample j, so kResolve adds the last segments itself, as for the direct plugin */
                bp.depth = 1;
                launchResolve(st, ctx->ds, bp, wb, film, ctr); launches[KKD_STAGE_RESOLVE]++;
                batchDone(st, batchIndex, bp.count);
                continue;
            }
            /* what the last segments found goes into the per-sample sums; kResolve then only splats them */
            bp.depth = 0x7fffffff;
            PathLaunch pl; pl.nIn = n; pl.identity = identity; pl.compact = 0; pl.terminate = 1; pl.counters = counters;
            launchPathBounce(st, ctx->ds, bp, sets[cur], sets[cur], pl); launches[KKD_STAGE_SHADE]++;
            bp.depth = 1;
            BatchParams bpSplat = bp; bpSplat.nB = 0; bpSplat.nL = 0;
            launchResolve(st, ctx->ds, bpSplat, wb, film, ctr); launches[KKD_STAGE_RESOLVE]++;
            batchDone(st, batchIndex, bp.count);
            continue;
        }
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
        launchResolve(st, ctx->ds, bp, wb, film, ctr); launches[KKD_STAGE_RESOLVE]++;
        batchDone(st, batchIndex, bp.count);
        if (timeStages) hipEventRecord(ctx->events[ev++], st);
    }
    KKD_HIP(ctx, hipGetLastError());
    for (int l = 1; l < nLanes; ++l) {   /* join: the context's stream continues after the other lanes have drained */
        KKD_HIP(ctx, hipEventRecord(ctx->events[2 + l], laneStream[l]));
        KKD_HIP(ctx, hipStreamWaitEvent(st0, ctx->events[2 + l], 0));
    }
    KKD_HIP(ctx, hipEventRecord(ctx->events[1], st0));
    std::vector<DeviceCounters> shards(kCounterShards);
    KKD_HIP(ctx, hipMemcpyAsync(shards.data(), ctr, kCounterShards * sizeof(DeviceCounters), hipMemcpyDeviceToHost, st0));
    KKD_HIP(ctx, hipStreamSynchronize(st0));
    pollProgress(true);
    const DeviceCounters c = sumShards(shards.data(), shards.size());
    {
        /* Budget control over the whole render: too many parked rays and the tail round repeats their set-up for nothing, too
         * few and the first round idles in its tail.  With the persistent tail round the throughput of C4 is flat while 18-33 % of
         * the BSDF rays and 28-54 % of the live shadow rays are parked, that of C2 is best at 14 % / 17 % (budgets 26 / 19 on both) */
        auto adapt = [&](int &cap, int used, int envCap, int cls, double lo, double hi) {
            if (used <= 0 || c.rays[cls] < (1u << 16)) return;
            const double f = (double)c.parked[cls] / (double)c.rays[cls];
            if (envCap < 0) { if (f > hi) cap = std::min(64, cap + 2); else if (f < lo) cap = std::max(12, cap - 2); }
            static const bool dbg = getenv("KKD_DEBUG_ROUNDS") != nullptr;
            if (dbg) fprintf(stderr, "rounds: class %d budget %d parked %.3f -> budget %d\n", cls, used, f, envCap < 0 ? cap : used);
        };
        adapt(ctx->capClosest, capClosest, envCapClosest, KKD_RAYS_BSDF, 0.13, 0.23);
        adapt(ctx->capAny, capAny, envCapAny, KKD_RAYS_SHADOW, 0.16, 0.34);
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        float ms = 0.f;
        KKD_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[0], ctx->events[1]));
        fillCounterStats(stats, c, count);
        stats->seconds = ms * 1e-3;
        if (timeStages) for (size_t k = evBase; k + 5 < ev; k += 6)
            for (int sIdx = 0; sIdx < 5; ++sIdx) { float t = 0.f; if (hipEventElapsedTime(&t, ctx->events[k + sIdx], ctx->events[k + sIdx + 1]) == hipSuccess) stats->stage_seconds[sIdx] += t * 1e-3; }
        for (int sIdx = 0; sIdx < 5; ++sIdx) stats->stage_launches[sIdx] = launches[sIdx];
        stats->seconds_trace = stats->stage_seconds[KKD_STAGE_PRIMARY] + stats->stage_seconds[KKD_STAGE_BSDF] + stats->stage_seconds[KKD_STAGE_SHADOW];
    }
    if (cancelled) return fail(ctx, KKD_ERR_CANCELLED, "render cancelled");
    return KKD_OK;
}

int kkd_render_device(kkd_ctx *ctx, const kkd_render_params *params, void *dev_film, kkd_stats *stats) {
    if (!ctx) return KKD_ERR_INVALID;
    if (!dev_film) return fail(ctx, KKD_ERR_INVALID, "kkd_render_device: film buffer is NULL");
    return renderImpl(ctx, params, (float *)dev_film, stats);
}

int kkd_render(kkd_ctx *ctx, const kkd_render_params *params, float *out_film, kkd_stats *stats) {
    if (!ctx) return KKD_ERR_INVALID;
    if (!out_film) return fail(ctx, KKD_ERR_INVALID, "kkd_render: film buffer is NULL");
    int rc = renderImpl(ctx, params, nullptr, stats);
    if (rc != KKD_OK && rc != KKD_ERR_CANCELLED) return rc;
    const size_t bytes = (size_t)ctx->ds.sensor.cropW * ctx->ds.sensor.cropH * 5 * sizeof(float);
    hipError_t e = hipMemcpy(out_film, ctx->film.p, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, KKD_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    return rc;
}

} /* extern "C" */

/* ==== multi-device rendering inside the library =============================================================
 * One context per device, the scene flattened once and replicated, one host thread per device while rendering, no exchange
 * until the frame is done.  Replaces, for one node, what the reference does with its scheduler: workers render blocks
 * (src/librender/renderproc.cpp:68-86) and BlockedRenderProcess::processResult adds every block into the film under a
 * mutex (renderproc.cpp:142-149 -> Film::put, src/films/hdrfilm.cpp:391-393).
 *   tile mode:    rank r renders the tiles dealt to it (Morton round-robin, tilesOfRank) with their filter borders; its tiles
 *                 are packed, moved to the root device by ONE peer copy (hipMemcpyPeerAsync over xGMI: film size / N plus
 *                 borders per rank, not a whole film) and added into the root film there -- Film::put(block) per tile;
 *   sample mode:  every rank renders the whole frame with its share of the samples; the films are summed by ONE
 *                 ncclReduce (RCCL, loaded at run time) to the root.  Without RCCL, or with one device listed twice (the
 *                 one-GPU rehearsal of the tests), the films are copied to the root and added there. */
#include <dlfcn.h>
#include <thread>

namespace {
/* the few RCCL entry points used, bound at run time so that libkkdirect.so loads on machines without librccl */
struct Rccl {
    typedef void *Comm;
    int (*CommInitAll)(Comm *, int, const int *) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void *, void *, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    void *lib = nullptr; std::string loadedFrom;
    static constexpr int kFloat = 7 /* ncclFloat32 */, kSum = 0 /* ncclSum */;
    bool load() {
        if (lib) return true;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) { loadedFrom = name; break; } }
        if (!lib) return false;
        GetVersion = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
} /* namespace */

struct kkd_multi {
    std::vector<kkd_ctx *> ctx;          /* ctx[0] is the root: its device receives the merged film */
    uint32_t split = KKD_SPLIT_TILE;
    std::string err;
    bool haveScene = false;
    std::vector<DevBuf> packed;          /* per rank, on its device: its tiles, packed (tile mode) */
    std::vector<DevBuf> staging;         /* per rank, on the root device: what the rank sent */
    std::vector<DevBuf> rankTiles;       /* per rank, on the root device: the rank's tile list */
    DevBuf rootFilm;                     /* root film of kkd_multi_render (host output) */
    Rccl rccl; std::vector<Rccl::Comm> comms; bool distinct = false;
    std::string mergePath;               /* "single", "tile-gather", "rccl-reduce", "copy-add" (what the last render used) */
    std::string diagnostics;             /* what kkd_multi_create found: peer access per device pair, librccl (kkd_multi_diagnostics) */
    /* Film::develop during a render (kkd_multi_request_flush): every rank hands a copy of ITS film to onRankFlush (between two
     * of its batches, or when its share is done); the films of a tile-sharded or sample-split job add up to the job's film,
     * so the rank that completes a round calls the user's function with the sum */
    kkd_flush_fn flushFn = nullptr; void *flushUser = nullptr;
    struct RankFlush { kkd_multi *m; uint32_t rank; };
    std::vector<RankFlush> rankFlush;
    std::mutex flushMutex;
    std::atomic<uint32_t> flushWanted{0};    /* rounds requested so far */
    uint32_t flushRound = 0;                 /* round being collected (0 = none) */
    std::vector<uint32_t> flushGiven;        /* per rank: the last round it contributed to */
    std::vector<float> flushSum; uint32_t flushArrived = 0; uint64_t flushDone = 0, flushTotal = 0;
    bool rendering = false;
    static void onRankFlush(void *user, const float *film, uint32_t w, uint32_t h, uint64_t done, uint64_t total) {
        RankFlush *rf = (RankFlush *)user; rf->m->contributeFlush(rf->rank, film, w, h, done, total);
    }
    void contributeFlush(uint32_t rank, const float *film, uint32_t w, uint32_t h, uint64_t done, uint64_t total) {
        std::lock_guard<std::mutex> lock(flushMutex);
        const uint32_t wanted = flushWanted.load();
        if (flushRound == 0) {                 /* first film of a new round */
            if (flushGiven[rank] >= wanted) return;
            flushRound = wanted; flushArrived = 0; flushDone = flushTotal = 0;
            flushSum.assign((size_t)w * h * 5, 0.f);
        }
        if (flushGiven[rank] >= flushRound) return;      /* this rank is already in */
        flushGiven[rank] = flushRound;
        const size_t n = (size_t)w * h * 5;
        for (size_t i = 0; i < n; ++i) flushSum[i] += film[i];
        flushDone += done; flushTotal += total;
        if (++flushArrived == (uint32_t)ctx.size()) {
            if (flushFn) flushFn(flushUser, flushSum.data(), w, h, flushDone, flushTotal);
            flushRound = 0;
        }
    }
    /* progress of the whole job: the per-device reports are summed under a mutex and passed on */
    kkd_progress_fn progressFn = nullptr; void *progressUser = nullptr;
    struct RankProgress { kkd_multi *m; uint32_t rank; };
    std::vector<RankProgress> rankProgress; std::vector<uint64_t> done, total; std::mutex progressMutex;
    static void onRankProgress(void *user, uint64_t d, uint64_t t) {
        RankProgress *rp = (RankProgress *)user; kkd_multi *m = rp->m;
        std::lock_guard<std::mutex> lock(m->progressMutex);
        m->done[rp->rank] = d; m->total[rp->rank] = t;
        uint64_t sd = 0, stt = 0;
        for (size_t r = 0; r < m->done.size(); ++r) { sd += m->done[r]; stt += m->total[r]; }
        if (m->progressFn) m->progressFn(m->progressUser, sd, stt);
    }
};

static int mfail(kkd_multi *m, int code, const std::string &msg) { m->err = msg; return code; }

extern "C" {

kkd_multi *kkd_multi_create(const kkd_multi_config *cfg) {
    if (!cfg || cfg->n_devices == 0 || cfg->n_devices > 64 || (!cfg->devices && cfg->n_devices > 0) || cfg->split > KKD_SPLIT_SAMPLE) {
        g_globalError = "kkd_multi_create: bad configuration (1..64 devices, split = KKD_SPLIT_TILE | KKD_SPLIT_SAMPLE)"; return nullptr;
    }
    kkd_multi *m = new kkd_multi();
    m->split = cfg->split;
    for (uint32_t r = 0; r < cfg->n_devices; ++r) {
        kkd_config c{}; c.device = cfg->devices[r]; c.flags = cfg->flags; c.seed = cfg->seed; c.max_samples_in_flight = cfg->max_samples_in_flight;
        kkd_ctx *x = kkd_create(&c);
        if (!x) { kkd_multi_destroy(m); return nullptr; }       /* g_globalError holds the reason */
        m->ctx.push_back(x);
    }
    const size_t n = m->ctx.size();
    m->packed.resize(n); m->staging.resize(n); m->rankTiles.resize(n);
    m->distinct = true;
    for (size_t a = 0; a < n; ++a) for (size_t b = a + 1; b < n; ++b) if (m->ctx[a]->device == m->ctx[b]->device) m->distinct = false;
    /* What the first run on real devices needs to name the cause of a failure: peer access per pair and the RCCL library, kept
     * as text (kkd_multi_diagnostics) and printed to stderr when $KKD_MULTI_VERBOSE is set. */
    std::string &dg = m->diagnostics;
    dg = "devices:";
    for (size_t r = 0; r < n; ++r) dg += " " + std::to_string(m->ctx[r]->device);
    dg += m->distinct ? " (distinct)" : " (a device is listed more than once: films are merged by copies on it, RCCL is not used)";
    if (m->distinct && n > 1) {           /* direct xGMI copies between the root and every other device, where the platform allows them */
        for (size_t r = 1; r < n; ++r) {
            const int a = m->ctx[0]->device, b = m->ctx[r]->device;
            int canAB = 0, canBA = 0;
            const hipError_t qa = hipDeviceCanAccessPeer(&canAB, a, b), qb = hipDeviceCanAccessPeer(&canBA, b, a);
            hipError_t ea = hipErrorUnknown, eb = hipErrorUnknown;
            if (qa == hipSuccess && canAB) { hipSetDevice(a); ea = hipDeviceEnablePeerAccess(b, 0); }
            if (qb == hipSuccess && canBA) { hipSetDevice(b); eb = hipDeviceEnablePeerAccess(a, 0); }
            auto enabled = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
            dg += "; peer " + std::to_string(a) + "->" + std::to_string(b) + ": " + (qa != hipSuccess ? std::string("query failed: ") + hipGetErrorString(qa) : !canAB ? std::string("no") : enabled(ea) ? std::string("enabled") : std::string("enable failed: ") + hipGetErrorString(ea));
            dg += ", " + std::to_string(b) + "->" + std::to_string(a) + ": " + (qb != hipSuccess ? std::string("query failed: ") + hipGetErrorString(qb) : !canBA ? std::string("no") : enabled(eb) ? std::string("enabled") : std::string("enable failed: ") + hipGetErrorString(eb));
        }
        (void)hipGetLastError();          /* "already enabled" is not an error */
    }
    if (m->split == KKD_SPLIT_SAMPLE) {
        if (m->rccl.load()) {
            int v = 0;
            dg += "; librccl: " + m->rccl.loadedFrom;
            if (m->rccl.GetVersion && m->rccl.GetVersion(&v) == 0) dg += " version " + std::to_string(v);
        } else dg += std::string("; librccl: not loaded (") + (dlerror() ? "dlopen failed" : "symbols missing") + "): films are merged by copy + add";
    }
    m->flushGiven.assign(n, 0u);
    if (getenv("KKD_MULTI_VERBOSE")) fprintf(stderr, "kkd_multi_create: %s\n", dg.c_str());
    return m;
}

void kkd_multi_destroy(kkd_multi *m) {
    if (!m) return;
    for (Rccl::Comm c : m->comms) if (c && m->rccl.CommDestroy) m->rccl.CommDestroy(c);
    for (size_t r = 0; r < m->ctx.size(); ++r) {
        if (!m->ctx[r]) continue;
        hipSetDevice(m->ctx[r]->device);
        if (r < m->packed.size()) m->packed[r].release();
    }
    if (!m->ctx.empty() && m->ctx[0]) {
        hipSetDevice(m->ctx[0]->device);
        for (DevBuf &b : m->staging) b.release();
        for (DevBuf &b : m->rankTiles) b.release();
        m->rootFilm.release();
    }
    for (kkd_ctx *x : m->ctx) kkd_destroy(x);
    delete m;
}

int kkd_multi_set_progress_callback(kkd_multi *m, kkd_progress_fn fn, void *user) {
    if (!m) return KKD_ERR_INVALID;
    std::lock_guard<std::mutex> lock(m->progressMutex);
    m->progressFn = fn; m->progressUser = user;
    const size_t n = m->ctx.size();
    m->rankProgress.resize(n); m->done.assign(n, 0); m->total.assign(n, 0);
    for (size_t r = 0; r < n; ++r) {
        m->rankProgress[r] = { m, (uint32_t)r };
        kkd_set_progress_callback(m->ctx[r], fn ? &kkd_multi::onRankProgress : nullptr, &m->rankProgress[r]);
    }
    return KKD_OK;
}

const char *kkd_multi_last_error(kkd_multi *m) { return m ? m->err.c_str() : g_globalError.c_str(); }
void kkd_multi_request_cancel(kkd_multi *m) { if (m) for (kkd_ctx *x : m->ctx) kkd_request_cancel(x); }
uint32_t kkd_multi_device_count(kkd_multi *m) { return m ? (uint32_t)m->ctx.size() : 0u; }
const char *kkd_multi_merge_path(kkd_multi *m) { return m ? m->mergePath.c_str() : ""; }
const char *kkd_multi_diagnostics(kkd_multi *m) { return m ? m->diagnostics.c_str() : ""; }

int kkd_multi_set_flush_callback(kkd_multi *m, kkd_flush_fn fn, void *user) {
    if (!m) return KKD_ERR_INVALID;
    std::lock_guard<std::mutex> lock(m->flushMutex);
    m->flushFn = fn; m->flushUser = user;
    const size_t n = m->ctx.size();
    m->rankFlush.resize(n);
    for (size_t r = 0; r < n; ++r) {
        m->rankFlush[r] = { m, (uint32_t)r };
        kkd_set_flush_callback(m->ctx[r], fn ? &kkd_multi::onRankFlush : nullptr, &m->rankFlush[r]);
    }
    return KKD_OK;
}

void kkd_multi_request_flush(kkd_multi *m) {
    if (!m || !m->flushFn) return;
    m->flushWanted.fetch_add(1);
    for (kkd_ctx *x : m->ctx) kkd_request_flush(x);
}

int kkd_multi_scene_upload(kkd_multi *m, const kkd_scene_desc *sd) {
    if (!m) return KKD_ERR_INVALID;
    m->haveScene = false;
    HostScene hs;
    int rc = buildHostScene(m->ctx[0], sd, hs);                  /* the BVH and the tables are built ONCE ... */
    if (rc != KKD_OK) return mfail(m, rc, m->ctx[0]->err);
    std::vector<int> rcs(m->ctx.size(), KKD_OK);
    std::vector<std::thread> th;
    for (size_t r = 0; r < m->ctx.size(); ++r)                   /* ... and copied to every device side by side */
        th.emplace_back([&, r] { m->ctx[r]->haveScene = false; rcs[r] = uploadHostScene(m->ctx[r], hs, sd); });
    for (std::thread &t : th) t.join();
    for (size_t r = 0; r < m->ctx.size(); ++r) if (rcs[r] != KKD_OK) return mfail(m, rcs[r], "device " + std::to_string(m->ctx[r]->device) + ": " + m->ctx[r]->err);
    m->haveScene = true; m->err.clear();
    return KKD_OK;
}

static int multiRender(kkd_multi *m, const kkd_render_params *rp, float *rootDevFilm, kkd_stats *stats) {
    if (!m->haveScene) return mfail(m, KKD_ERR_INVALID, "no scene uploaded");
    if (!rp) return mfail(m, KKD_ERR_INVALID, "render params are NULL");
    const uint32_t n = (uint32_t)m->ctx.size();
    kkd_ctx *root = m->ctx[0];
    const int W = root->ds.sensor.cropW, H = root->ds.sensor.cropH;
    const size_t filmFloats = (size_t)W * H * 5;
    if (m->split == KKD_SPLIT_SAMPLE && rp->spp < n) return mfail(m, KKD_ERR_INVALID, "sample-split mode needs spp >= number of devices");
    /* the caller's own shard (tile_index / tile_count / sample_first of a larger job) is refined, not replaced */
    const uint32_t outerCount = rp->tile_count ? rp->tile_count : 1, outerIndex = rp->tile_count ? rp->tile_index : 0;
    std::vector<kkd_render_params> prm(n, *rp);
    uint32_t sampleAt = rp->sample_first;
    for (uint32_t r = 0; r < n; ++r) {
        if (m->split == KKD_SPLIT_TILE) { prm[r].tile_count = outerCount * n; prm[r].tile_index = outerIndex * n + r; }
        else { prm[r].spp = rp->spp / n + (r < rp->spp % n ? 1u : 0u); prm[r].sample_first = sampleAt; sampleAt += prm[r].spp;
               if (!prm[r].sample_count_total) prm[r].sample_count_total = rp->sample_first + rp->spp; }
    }
    std::vector<kkd_stats> st(n);
    std::vector<int> rcs(n, KKD_OK);
    std::vector<uint32_t> nTilesOf(n, 0);
    const uint32_t bs = rp->block_size ? rp->block_size : 32;
    const int border = root->ds.filter.border, edge = root->ds.sensor.hqEdges ? border : 0;
    TileGrid grid; grid.bs = bs; grid.nbx = (uint32_t)((W + 2 * edge + (int)bs - 1) / (int)bs); grid.nby = (uint32_t)((H + 2 * edge + (int)bs - 1) / (int)bs);
    grid.border = border; grid.edge = edge; grid.W = W; grid.H = H; grid.rankOf = nullptr; grid.myRank = 0;
    const size_t perTile = (size_t)(bs + 2 * border) * (bs + 2 * border) * 5;
    const bool gather = m->split == KKD_SPLIT_TILE && n > 1;
    auto worker = [&](uint32_t r) {
        kkd_ctx *x = m->ctx[r];
        rcs[r] = renderImpl(x, &prm[r], r == 0 ? rootDevFilm : nullptr, &st[r]);      /* ranks > 0 render into their own film buffer */
        if (rcs[r] == KKD_OK && m->flushFn && m->flushGiven[r] < m->flushWanted.load()) {
            /* a flush was requested that this rank did not get to answer between two batches (its share was done, or the
             * request came after its last batch was queued): its finished film completes the round */
            const size_t bytes = filmFloats * sizeof(float);
            hipSetDevice(x->device);
            if (x->flushHostBytes < bytes) {
                if (x->flushHost) { hipHostFree(x->flushHost); x->flushHost = nullptr; x->flushHostBytes = 0; }
                if (hipHostMalloc((void **)&x->flushHost, bytes, hipHostMallocDefault) == hipSuccess) x->flushHostBytes = bytes;
            }
            const float *src = r == 0 ? rootDevFilm : (const float *)x->film.p;
            if (x->flushHostBytes >= bytes && hipMemcpy(x->flushHost, src, bytes, hipMemcpyDeviceToHost) == hipSuccess)
                m->contributeFlush(r, x->flushHost, (uint32_t)W, (uint32_t)H, st[r].samples, st[r].samples);
        }
        if (rcs[r] != KKD_OK || r == 0 || !gather) return;
        /* tile mode: pack this rank's tiles (with borders) and send them to the root device in one copy */
        /* the rank's tile count is a function of the grid alone (nTilesOf, filled in with the staging buffers below): never
         * what an earlier render on this context left in its list */
        const uint32_t nt = nTilesOf[r];
        if (nt == 0) return;
        if (x->tileListHost.size() != nt) { x->err = "tile gather: the context's tile list does not match the grid"; rcs[r] = KKD_ERR_DEVICE; return; }
        hipError_t e = m->packed[r].ensure(nt * perTile * sizeof(float));
        TileGrid mine = grid; mine.rankOf = (const uint32_t *)x->tileList.p + nt; mine.myRank = prm[r].tile_index;
        if (e == hipSuccess) { launchPackTiles(x->stream, (const float *)x->film.p, (float *)m->packed[r].p, (const uint32_t *)x->tileList.p, nt, mine); e = hipGetLastError(); }
        if (e == hipSuccess) e = x->device == root->device ? hipMemcpyAsync(m->staging[r].p, m->packed[r].p, nt * perTile * sizeof(float), hipMemcpyDeviceToDevice, x->stream)
                                                           : hipMemcpyPeerAsync(m->staging[r].p, root->device, m->packed[r].p, x->device, nt * perTile * sizeof(float), x->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(x->stream);
        if (e != hipSuccess) { x->err = std::string("tile gather: ") + hipGetErrorString(e); rcs[r] = KKD_ERR_DEVICE; }
    };
    if (gather) {
        /* staging buffers and the ranks' tile lists on the root device, sized before the workers start (the lists are a
         * function of the grid alone) */
        hipSetDevice(root->device);
        const uint32_t nby = grid.nby;
        if (grid.nbx > 0xffffu || nby > 0xffffu) return mfail(m, KKD_ERR_UNSUPPORTED, "more than 65535 tiles along one axis");
        std::vector<uint32_t> list;
        for (uint32_t r = 1; r < n; ++r) {
            tilesOfRank(grid.nbx, nby, prm[r].tile_index, prm[r].tile_count, list);
            nTilesOf[r] = (uint32_t)list.size();
            hipError_t e = m->staging[r].ensure(std::max<size_t>(16, list.size() * perTile * sizeof(float)));
            if (e == hipSuccess) e = m->rankTiles[r].ensure(std::max<size_t>(16, list.size() * sizeof(uint32_t)));
            if (e == hipSuccess && !list.empty()) e = hipMemcpy(m->rankTiles[r].p, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) return mfail(m, e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE, std::string("tile gather buffers: ") + hipGetErrorString(e));
        }
    }
    {   /* flush rounds do not outlive a render */
        std::lock_guard<std::mutex> lock(m->flushMutex);
        m->flushWanted.store(0); m->flushRound = 0; m->flushGiven.assign(n, 0u);
    }
    if (m->progressFn) {   /* every rank's total is known only once it reports: start from "nothing done of nothing" */
        std::lock_guard<std::mutex> lock(m->progressMutex);
        m->done.assign(n, 0); m->total.assign(n, 0);
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < n; ++r) th.emplace_back(worker, r);
    worker(0);
    for (std::thread &t : th) t.join();
    int rc = KKD_OK;
    for (uint32_t r = 0; r < n; ++r) if (rcs[r] != KKD_OK && rc == KKD_OK) { rc = rcs[r]; m->err = "device " + std::to_string(m->ctx[r]->device) + ": " + m->ctx[r]->err; }
    if (rc != KKD_OK) return rc;

    /* ---- film merge on the root device ---- */
    hipSetDevice(root->device);
    hipError_t e = hipSuccess;
    m->mergePath = "single";
    if (gather) {
        m->mergePath = "tile-gather";
        for (uint32_t r = 1; r < n && e == hipSuccess; ++r) {
            launchUnpackAddTiles(root->stream, rootDevFilm, (const float *)m->staging[r].p, (const uint32_t *)m->rankTiles[r].p, nTilesOf[r], grid);
            e = hipGetLastError();
        }
    } else if (m->split == KKD_SPLIT_SAMPLE) {
        bool reduced = false;
        if (m->distinct && m->rccl.load()) {
            /* one communicator per device, created on first use; a single-process group call issues the per-device halves */
            if (m->comms.empty()) {
                std::vector<int> devs; for (kkd_ctx *x : m->ctx) devs.push_back(x->device);
                m->comms.assign(n, nullptr);
                const int r0 = m->rccl.CommInitAll(m->comms.data(), (int)n, devs.data());
                if (r0 != 0) { m->comms.clear(); m->err = std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r0) : "failed"); }
            }
            if (!m->comms.empty()) {
                int r1 = m->rccl.GroupStart();
                for (uint32_t r = 0; r < n && r1 == 0; ++r) {
                    const float *src = r == 0 ? rootDevFilm : (const float *)m->ctx[r]->film.p;
                    r1 = m->rccl.Reduce(src, r == 0 ? rootDevFilm : (float *)m->ctx[r]->film.p, filmFloats, Rccl::kFloat, Rccl::kSum, 0, m->comms[r], m->ctx[r]->stream);
                }
                const int r2 = m->rccl.GroupEnd();
                if (r1 == 0 && r2 == 0) {
                    for (uint32_t r = 0; r < n && e == hipSuccess; ++r) { hipSetDevice(m->ctx[r]->device); e = hipStreamSynchronize(m->ctx[r]->stream); }
                    hipSetDevice(root->device);
                    reduced = e == hipSuccess; if (reduced) m->mergePath = "rccl-reduce";
                } else return mfail(m, KKD_ERR_DEVICE, std::string("ncclReduce: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r1 ? r1 : r2) : "failed"));
            }
        }
        if (!reduced && n > 1 && e == hipSuccess) {
            m->mergePath = "copy-add";
            for (uint32_t r = 1; r < n && e == hipSuccess; ++r) {
                kkd_ctx *x = m->ctx[r];
                e = m->staging[r].ensure(filmFloats * sizeof(float));
                if (e == hipSuccess) e = x->device == root->device ? hipMemcpyAsync(m->staging[r].p, x->film.p, filmFloats * sizeof(float), hipMemcpyDeviceToDevice, root->stream)
                                                                   : hipMemcpyPeerAsync(m->staging[r].p, root->device, x->film.p, x->device, filmFloats * sizeof(float), root->stream);
                if (e == hipSuccess) { launchAddFilm(root->stream, rootDevFilm, (const float *)m->staging[r].p, filmFloats); e = hipGetLastError(); }
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(root->stream);
    if (e != hipSuccess) return mfail(m, KKD_ERR_DEVICE, std::string("film merge: ") + hipGetErrorString(e));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (uint32_t r = 0; r < n; ++r) {
            const kkd_stats &s = st[r];
            stats->samples += s.samples; stats->rays_closest += s.rays_closest; stats->rays_shadow += s.rays_shadow;
            stats->algorithmic_bytes += s.algorithmic_bytes; stats->nodes_visited += s.nodes_visited; stats->tris_tested += s.tris_tested;
            for (int k = 0; k < 3; ++k) { stats->class_rays[k] += s.class_rays[k]; stats->class_nodes[k] += s.class_nodes[k]; stats->class_tris[k] += s.class_tris[k]; }
            for (int k = 0; k < 5; ++k) { stats->stage_seconds[k] += s.stage_seconds[k]; stats->stage_launches[k] += s.stage_launches[k]; }
            stats->seconds_trace = std::max(stats->seconds_trace, s.seconds_trace);
        }
        stats->seconds = wall;           /* host wall clock of the whole job: the slowest device + the merge */
    }
    m->err.clear();
    return KKD_OK;
}

int kkd_multi_render_device(kkd_multi *m, const kkd_render_params *params, void *dev_film, kkd_stats *stats) {
    if (!m) return KKD_ERR_INVALID;
    if (!dev_film) return mfail(m, KKD_ERR_INVALID, "kkd_multi_render_device: film buffer is NULL");
    return multiRender(m, params, (float *)dev_film, stats);
}

int kkd_multi_render(kkd_multi *m, const kkd_render_params *params, float *out_film, kkd_stats *stats) {
    if (!m) return KKD_ERR_INVALID;
    if (!out_film) return mfail(m, KKD_ERR_INVALID, "kkd_multi_render: film buffer is NULL");
    if (!m->haveScene) return mfail(m, KKD_ERR_INVALID, "no scene uploaded");
    kkd_ctx *root = m->ctx[0];
    const size_t bytes = (size_t)root->ds.sensor.cropW * root->ds.sensor.cropH * 5 * sizeof(float);
    hipSetDevice(root->device);
    hipError_t e = m->rootFilm.ensure(bytes);
    if (e != hipSuccess) return mfail(m, e == hipErrorOutOfMemory ? KKD_ERR_NOMEM : KKD_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    int rc = multiRender(m, params, (float *)m->rootFilm.p, stats);
    if (rc != KKD_OK) return rc;
    hipSetDevice(root->device);
    e = hipMemcpy(out_film, m->rootFilm.p, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return mfail(m, KKD_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    return KKD_OK;
}

} /* extern "C" */
