/*
 * kkd_kernels.hip -- the gfx950 kernels of libkkdirect, wavefront style: every stage is one launch over
 * a batch of camera samples and hands the next stage small SoA records that stay resident in HBM / the
 * 256 MiB Infinity Cache.
 *
 *   kPrimary   sensor ray generation fused with closest-hit traversal      (perspective.cpp:271-298 +
 *                                                                           skdtree.cpp:112-142)
 *   kShade     hit record, Le, BSDF sampling, emitter sampling, MIS terms   (kk/direct.cpp:50-160)
 *   kTrace     closest-hit / any-hit traversal of a ray buffer              (skdtree.cpp:112-142, 207-226)
 *   kResolve   BSDF-ray emitter lookup + light pdf, MIS sum, filtered splat (kk/direct.cpp:105-123,162-178,
 *                                                                           integrator.cpp:183-184, ImageBlock::put)
 * Reference paths are relative to /root/reference/src.
 *
 * Work item -> camera sample mapping (replaces the Hilbert-curve pixel walk of renderproc.cpp:79 and the
 * spp loop of integrator.cpp:162-186): the `g` lanes of a lane group (g = min(64, pow2ceil(spp))) hold
 * g samples of ONE pixel, so primary rays of a wave are maximally coherent and the film splat can be
 * reduced across the group before any atomic is issued.
 */
#include <algorithm>
#include "kkd_kernels.h"
#include "kkd_trace.h"
#include "kkd_bsdf.h"
#include "kkd_shade.h"

namespace kkd {

/* ablation switches ($KKD_DEBUG bits, kkd_kernels.h) exist only in -DKKD_ABLATION builds; a release build folds them away */
#ifdef KKD_ABLATION
__host__ __device__ __forceinline__ bool ablation(const BatchParams &bp, uint32_t bit) { return (bp.debug & bit) != 0u; }
#else
__host__ __device__ __forceinline__ constexpr bool ablation(const BatchParams &, uint32_t) { return false; }
#endif

/* ---- work mapping ------------------------------------------------------------------------------ */
struct SampleId { int X, Y; uint32_t pixel, sample; bool active; };

__device__ __forceinline__ uint32_t compact1by1(uint32_t x) {
    x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x;
}

__device__ __forceinline__ SampleId mapWorkItem(const BatchParams &bp, uint32_t i) {
    SampleId s;
    /* 32-bit arithmetic only (64-bit division costs hundreds of instructions per lane, and this runs in three kernels):
     * the host has split the batch origin into (tile0, pin0, chunk0); shifts when the divisors are powers of two */
    const uint32_t j = i & (bp.g - 1u);
    const uint32_t c = bp.chunk0 + (i >> bp.gShift);
    uint32_t chunk, sAdd;
    if (bp.chunksShift < 32u) { sAdd = c >> bp.chunksShift; chunk = c & (bp.chunks - 1u); }
    else { sAdd = c / bp.chunks; chunk = c - sAdd * bp.chunks; }
    const uint32_t p = bp.pin0 + sAdd, pixPerTile = bp.bs * bp.bs;
    uint32_t tAdd, pin;
    if (bp.pixShift < 32u) { tAdd = p >> bp.pixShift; pin = p & (pixPerTile - 1u); }
    else { tAdd = p / pixPerTile; pin = p - tAdd * pixPerTile; }
    const uint32_t tileLocal = bp.tile0 + tAdd;
    const uint32_t tile = bp.tileList ? bp.tileList[min(tileLocal, bp.nTilesMine - 1u)] : tileLocal;   /* (lanes past the end are inactive) */
    const uint32_t ty = tile / bp.nbx, tx = tile - ty * bp.nbx;
    uint32_t px, py;
    if (bp.bsPow2) { px = compact1by1(pin); py = compact1by1(pin >> 1); }
    else { px = pin % bp.bs; py = pin / bp.bs; }
    s.X = (int)(tx * bp.bs + px) - bp.edge; s.Y = (int)(ty * bp.bs + py) - bp.edge;   /* tiles start at (-edge, -edge), renderproc.cpp:163-171 */
    s.sample = chunk * bp.g + j;
    s.active = i < bp.count && tileLocal < bp.nTilesMine && s.X < bp.W + bp.edge && s.Y < bp.H + bp.edge && s.sample < bp.spp;
    s.pixel = (uint32_t)((s.Y + bp.edge) * (bp.W + 2 * bp.edge) + (s.X + bp.edge));
    s.sample += bp.sampleFirst;
    return s;
}

/* Sampler::next2D / next2DArray / next1D of sample s: `flat` is the dimension of the independent stream (what every request
 * beyond the sampler's 'dimension' falls back to, ldsampler.cpp:214-228, stratified.cpp:186-187, 201-205), `req` the running
 * number of the 2D (1D) request within the sample, (array, n, e) element e of the sample's n-element slice of requested array
 * `array`.  mode: KKD_SAMPLER_* -- a compile-time constant in kShade, whose register budget has no room for all samplers. */
__device__ __forceinline__ bool isHalton(uint32_t mode) { return mode >= KKD_SAMPLER_HALTON && mode <= KKD_SAMPLER_HAMMERSLEY_PLAIN; }   /* halton and hammersley: HaltonSetup tells them apart */
/* halton: one running dimension over all requests of a sample (m_dimension): the kk / direct plugins only make 2D requests, so it is
 * 2 * req; the path plugin's requests sit at fixed dimensions per vertex, the independent stream's (`flat`).  Beyond the prime
 * table (the reference stops with an error there, halton.cpp:356-358) the independent stream takes over. */
__device__ __forceinline__ uint32_t haltonDim(const BatchParams &bp, uint32_t flat, uint32_t req2) {
    return haltonSkipArrays(bp.halton, bp.integrator == KKD_INTEGRATOR_PATH ? flat : req2);
}
/* sobol: the running dimension is counted as for halton; beyond the direction-number table (the reference stops with an error,
 * sobol.cpp:235-237, 250-252) the independent stream takes over */
__device__ __forceinline__ uint32_t sobolDim(const BatchParams &bp, uint32_t flat, uint32_t req2) {
    return sobolSkipArrays(bp.sobol, bp.integrator == KKD_INTEGRATOR_PATH ? flat : req2);
}
__device__ __forceinline__ f2 draw2D(const BatchParams &bp, const SampleId &s, uint32_t flat, uint32_t req, uint32_t mode) {
    if (mode == KKD_SAMPLER_SOBOL) { const uint32_t d = sobolDim(bp, flat, 2u * req); if (d + 1u < kSobolDimsMax) return sobolSample2D(bp.sobol, s.X, s.Y, s.sample, d); }
    if (mode == KKD_SAMPLER_LDSAMPLER && req < bp.samplerDim) return ldSample2D(bp.seed, s.pixel, 2u * req, s.sample, bp.samplerSet);
    if (isHalton(mode)) { const uint32_t d = haltonDim(bp, flat, 2u * req); if (d + 1u < kHaltonDims) return haltonSample2D(bp.halton, s.X, s.Y, s.sample, d); }
    const f2 u = sample2D(bp.seed, s.pixel, s.sample, flat);
    if (mode == KKD_SAMPLER_STRATIFIED && req < bp.samplerDim) return stratSample2D(bp.seed, s.pixel, 2u * req, s.sample, bp.samplerRes, u);
    return u;
}
__device__ __forceinline__ f2 draw2DArray(const BatchParams &bp, const SampleId &s, uint32_t flat, uint32_t array, uint32_t n, uint32_t e, uint32_t mode) {
    if (mode == KKD_SAMPLER_LDSAMPLER) return ldSample2D(bp.seed, s.pixel, 0x10000u + array, s.sample * n + e, bp.samplerSet * n);
    if (isHalton(mode)) return haltonArray2D(bp.halton, s.X, s.Y, array, (uint64_t)s.sample * n + e);
    if (mode == KKD_SAMPLER_SOBOL) return sobolArray2D(bp.sobol, s.X, s.Y, array, (uint64_t)s.sample * n + e);
    const f2 u = sample2D(bp.seed, s.pixel, s.sample, flat);
    if (mode == KKD_SAMPLER_STRATIFIED) return latinHypercube2D(bp.seed, s.pixel, 0x10000u + array, s.sample * n + e, bp.samplerSet * n, u);
    return u;
}
__device__ __forceinline__ float draw1D(const BatchParams &bp, const SampleId &s, uint32_t flat, uint32_t req, uint32_t mode) {
    if (mode == KKD_SAMPLER_LDSAMPLER && req < bp.samplerDim) return ldSample1D(bp.seed, s.pixel, 2u * req + 1u, s.sample, bp.samplerSet);
    if (isHalton(mode)) { const uint32_t d = haltonDim(bp, flat, flat); if (d < kHaltonDims) return haltonSample1D(bp.halton, s.X, s.Y, s.sample, d); }
    if (mode == KKD_SAMPLER_SOBOL) { const uint32_t d = sobolDim(bp, flat, flat); if (d < kSobolDimsMax) return sobolSample1D(bp.sobol, s.X, s.Y, s.sample, d); }
    const float u = sample2D(bp.seed, s.pixel, s.sample, flat).x;
    if (mode == KKD_SAMPLER_STRATIFIED && req < bp.samplerDim) return stratSample1D(bp.seed, s.pixel, 2u * req + 1u, s.sample, bp.samplerRes, u);
    return u;
}

__device__ __forceinline__ DRay primaryRay(const DScene &sc, const BatchParams &bp, const SampleId &s, float &sx, float &sy, uint32_t mode) {
    const f2 jit = draw2D(bp, s, 0u, 0u, mode);
    sx = (float)s.X + jit.x; sy = (float)s.Y + jit.y;      /* integrator.cpp:171 */
    return sensorRay(sc.sensor, sx, sy);
}

__device__ __forceinline__ DeviceCounters *counterShard(DeviceCounters *ctr) { return ctr + (blockIdx.x % kCounterShards); }
__device__ __forceinline__ void countRays(bool traced, unsigned long long *counter) {
    const unsigned long long m = __ballot(traced);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}

/* ---- K1+K2: primary rays ------------------------------------------------------------------------- */
__device__ __forceinline__ void countTraversal(uint32_t nn, uint32_t nt, DeviceCounters *ctr, int cls) {
    for (int m = 32; m > 0; m >>= 1) { nn += __shfl_xor(nn, m); nt += __shfl_xor(nt, m); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&ctr->nodes[cls], (unsigned long long)nn); atomicAdd(&ctr->tris[cls], (unsigned long long)nt); }
}

#ifndef KKD_PRIMARY_RAYS
#define KKD_PRIMARY_RAYS 2     /* rays per lane of the packet walk of the primary rays: the walk is bound by its scalar side, which is
                                  paid per wave -- measured on C2: 1: 28.7 ms, 2: 21.3 ms, 3: 22.0 ms, 4: 21.8 ms per frame */
#endif
/* INDEP: the independent sampler, compiled without the other samplers' code (with them behind a run-time branch the packet walk
 * of C4 took 179 instead of 173 ms per step) */
template <bool COUNT, bool PROJ, bool PACKET, bool INDEP, int STACK, int BLOCK>
__global__ void __launch_bounds__(BLOCK, !PACKET || KKD_PRIMARY_RAYS == 1 ? 8 : (KKD_PRIMARY_RAYS == 2 ? 5 : 3)) kPrimary(DScene sc, BatchParams bp, DHit *__restrict__ hits, DeviceCounters *ctr) {
    /* PACKET: one wave-uniform stack per wave; else one stack per lane */
    __shared__ uint32_t sstack[PACKET ? (BLOCK / 64) * kMaxTraversalStack : STACK * BLOCK];
    constexpr int R = PACKET ? KKD_PRIMARY_RAYS : 1;
    /* a wave takes R * 64 consecutive work items: lane l the items base + l (+ 64): R lane groups of one pixel when spp allows */
    const uint32_t base = (blockIdx.x * BLOCK + (threadIdx.x & ~63u)) * R + (threadIdx.x & 63u);
    DRay ray[R]; DHit h[R]; bool active[R];
    uint32_t nn = 0, nt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const SampleId s = mapWorkItem(bp, base + 64u * r);
        ray[r].maxt = -1.f; ray[r].mint = 0.f; ray[r].ox = ray[r].oy = ray[r].oz = 0.f; ray[r].dx = ray[r].dy = 0.f; ray[r].dz = 1.f;
        float sx, sy;
        active[r] = s.active;
        if (s.active) ray[r] = primaryRay(sc, bp, s, sx, sy, INDEP ? KKD_SAMPLER_INDEPENDENT : bp.sampler);
    }
    const void *triTable = PROJ ? (const void *)sc.projTris : (const void *)sc.tris;
    if (PACKET) traversePacket<COUNT, PROJ, R>(sc.wideNodes, triTable, ray, sstack + (threadIdx.x >> 6) * kMaxTraversalStack, h, nn, nt);
    else h[0] = traverse<false, COUNT, PROJ, STACK, BLOCK>(sc.nodes, triTable, ray[0], sstack, nn, nt);
    ctr = counterShard(ctr);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (base + 64u * r < bp.count) hits[base + 64u * r] = h[r];
        countRays(active[r], &ctr->rays[KKD_RAYS_PRIMARY]);
    }
    if (COUNT) countTraversal(nn, nt, ctr, KKD_RAYS_PRIMARY);
}

template <int OUT> __device__ __forceinline__ void storeTraceResult(void *__restrict__ out, uint64_t i, DHit h, const LeafTri *__restrict__ tris);

#ifndef KKD_SORT_OCTANT
#define KKD_SORT_OCTANT 8      /* bins of the first round's direction sort: 8 = octants (kept), 64 = octahedral cells, 0 = none */
#endif
#ifndef KKD_FIRST_BLOCK
#define KKD_FIRST_BLOCK 256    /* workgroup of the first traversal round: the rays that are sorted together */
#endif
#ifndef KKD_REPACK_AT
#define KKD_REPACK_AT 0        /* experiment: steps of the first round after which a workgroup moves its walking rays together (0 = never) */
#endif
#ifndef KKD_REPACK_WORDS
#define KKD_REPACK_WORDS 4
#endif
#ifndef KKD_SORT_ANY
#define KKD_SORT_ANY 1         /* the same for shadow rays (0: only their dead rays are squeezed out, the layout of round 2) */
#endif
#if KKD_SORT_OCTANT
/* direction -> one of 64 bins: the octahedral map of the unit sphere cut into 8 x 8 cells, numbered along a Morton curve so
 * that neighbouring bins hold neighbouring directions */
__device__ __forceinline__ uint32_t directionBin(float dx, float dy, float dz) {
#if KKD_SORT_OCTANT == 8
    return (dx < 0.f ? 1u : 0u) | (dy < 0.f ? 2u : 0u) | (dz < 0.f ? 4u : 0u);
#else
    const float inv = frcp(fabsf(dx) + fabsf(dy) + fabsf(dz));
    float u = dx * inv, v = dy * inv;
    if (dz < 0.f) { const float fu = (1.f - fabsf(v)) * (u < 0.f ? -1.f : 1.f), fv = (1.f - fabsf(u)) * (v < 0.f ? -1.f : 1.f); u = fu; v = fv; }
    const uint32_t ix = min((uint32_t)max((u + 1.f) * 4.f, 0.f), 7u), iy = min((uint32_t)max((v + 1.f) * 4.f, 0.f), 7u);
    return (ix & 1u) | (iy & 1u) << 1 | (ix & 2u) << 1 | (iy & 2u) << 2 | (ix & 4u) << 2 | (iy & 4u) << 3;
#endif
}
#endif

/* SHARE: RaySource with dirs (kkd_kernels.h) */
template <bool SHARE>
__device__ __forceinline__ DRay loadRay(const DRay *__restrict__ rays, const float4 *__restrict__ dirs, uint32_t stride, uint64_t k) {
    if (!SHARE) return rays[k];
    const uint32_t k32 = (uint32_t)k, oi = k32 < stride ? k32 : k32 % stride;
    const float4 o = reinterpret_cast<const float4 *>(rays)[oi], d = dirs[k];
    DRay r; r.ox = o.x; r.oy = o.y; r.oz = o.z; r.mint = o.w; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.maxt = d.w;
    return r;
}

/* ---- generic ray-buffer traversal ------------------------------------------------------------------ */
/* OUT: 0 = DHit with LeafTri slot (internal), 1 = DHit with global primitive id (C ABI), 2 = occlusion word */
template <bool ANY, bool COUNT, int OUT, bool PROJ, int STACK, int BLOCK, bool SHARE = false>
__global__ void __launch_bounds__(BLOCK, 8) kTrace(const Node4 *__restrict__ nodes, const LeafTri *__restrict__ tris, const ProjTri *__restrict__ projTris,
                                                const DRay *__restrict__ rays, const float4 *__restrict__ dirs, uint32_t stride, uint64_t n, void *__restrict__ out, DeviceCounters *ctr, int cls) {
    __shared__ uint32_t sstack[STACK * BLOCK];
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    DRay ray; ray.maxt = -1.f; ray.mint = 0.f; ray.ox = ray.oy = ray.oz = 0.f; ray.dx = ray.dy = 0.f; ray.dz = 1.f;
    if (i < n) ray = loadRay<SHARE>(rays, dirs, stride, i);
    uint32_t nn = 0, nt = 0;
    DHit h = traverse<ANY, COUNT, PROJ, STACK, BLOCK>(nodes, PROJ ? (const void *)projTris : (const void *)tris, ray, sstack, nn, nt);
    if (i < n) storeTraceResult<OUT>(out, i, h, tris);
    if (ctr) {
        ctr = counterShard(ctr);
        const bool traced = i < n && ray.maxt > adaptiveMinT<ANY>(ray.mint, ray.ox, ray.oy, ray.oz);
        countRays(traced, &ctr->rays[cls]);
        if (COUNT) countTraversal(nn, nt, ctr, cls);
    }
}

/* ---- the same walk in two launches --------------------------------------------------------------------------------
 * A wave of incoherent rays runs as long as its longest lane: on C2 ~37 steps for a mean path of ~22, so in the late
 * steps most lanes idle while every instruction is still issued for the whole wave.  the first round therefore stops every
 * wave after iterCap steps; a lane that is not done parks its state (current reference, stack, clipped interval, best
 * hit) in a continuation record, and the next round, launched over the parked rays only, packs them 64 to a wave and
 * walks on (kTraceRound).  Lanes whose stack has spilled into private memory, or that find the continuation buffer full, simply
 * walk on in the first launch. */
template <int OUT>
__device__ __forceinline__ void storeTraceResult(void *__restrict__ out, uint64_t i, DHit h, const LeafTri *__restrict__ tris) {
    if (OUT == 2) reinterpret_cast<uint32_t *>(out)[i] = h.id != kNoHit ? 1u : 0u;
    else {
        if (OUT == 1 && h.id != kNoHit) h.id = tris[h.id].prim;
        if (OUT == 0 && h.id != kNoHit && tris[h.id].emitter == 0u) h.id |= kHitNotEmitter;
        reinterpret_cast<DHit *>(out)[i] = h;
    }
}

/* the LDS part of a parked ray's stack travels as (STACK + 3) / 4 uint4s; entries past STACK do not exist */
template <int STACK, int BLOCK> __device__ __forceinline__ void stackToLds(uint32_t *sstack, int col, const uint4 *src) {
#pragma unroll
    for (int q = 0; q < (STACK + 3) / 4; ++q) {
        const uint4 s4 = src[q];
        sstack[(4 * q) * BLOCK + col] = s4.x;
        if (4 * q + 1 < STACK) sstack[(4 * q + 1) * BLOCK + col] = s4.y;
        if (4 * q + 2 < STACK) sstack[(4 * q + 2) * BLOCK + col] = s4.z;
        if (4 * q + 3 < STACK) sstack[(4 * q + 3) * BLOCK + col] = s4.w;
    }
}
template <int STACK, int BLOCK> __device__ __forceinline__ void stackFromLds(const uint32_t *sstack, int col, uint4 *dst) {
#pragma unroll
    for (int q = 0; q < (STACK + 3) / 4; ++q)
        dst[q] = make_uint4(sstack[(4 * q) * BLOCK + col], 4 * q + 1 < STACK ? sstack[(4 * q + 1) * BLOCK + col] : 0u,
                            4 * q + 2 < STACK ? sstack[(4 * q + 2) * BLOCK + col] : 0u, 4 * q + 3 < STACK ? sstack[(4 * q + 3) * BLOCK + col] : 0u);
}
static_assert(kStackLds >= 4 && kStackLds <= 16, "TraceContinuation stores up to 16 stack words per parked ray");
static_assert(kStackLdsAny >= 4 && kStackLdsAny <= 16, "TraceContinuation stores up to 16 stack words per parked ray");

/* One round.  FIRST: lanes take rays [0, n) of the ray buffer; else lanes take the records parked in `in` by the round
 * before.  next.iterCap > 0: stop after that many steps and park what is still walking in `next`; else walk to the end. */
template <bool ANY, bool COUNT, int OUT, bool PROJ, bool FIRST, int STACK, int BLOCK, bool SHARE = false>
__global__ void __launch_bounds__(BLOCK, 8) kTraceRound(const Node4 *__restrict__ nodes, const LeafTri *__restrict__ tris, const ProjTri *__restrict__ projTris,
                                                        const DRay *__restrict__ rays, const float4 *__restrict__ dirs, uint32_t stride, uint64_t n, void *__restrict__ out, DeviceCounters *ctr, int cls,
                                                        TraceContinuation in, TraceContinuation next) {
    __shared__ uint32_t sstack[STACK * BLOCK];
    DRay ray; ray.maxt = -1.f; ray.mint = 0.f; ray.ox = ray.oy = ray.oz = 0.f; ray.dx = ray.dy = 0.f; ray.dz = 1.f;
    ContRec r; r.ray = 0; r.cur = kTraversalDone; r.sp = 0; r.tmax = -1.f; r.pend = 0u; r.u = r.v = 0.f; r.id = kNoHit;
    bool mine;                      /* this lane holds a ray */
    size_t inSlot = 0;
#if KKD_SORT_OCTANT
    /* The block's live rays sorted by the octant of their direction (counting sort in LDS, ray -> lane only: results go to the ray's own slot), so that the lanes of a
     * wave walk the tree in the same order; dead rays get their miss here */
    if (FIRST && (!ANY || KKD_SORT_ANY)) {
        const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
        DRay own = ray;
        const bool have = i < n;
        if (have) own = loadRay<SHARE>(rays, dirs, stride, i);
        const bool live = have && own.maxt > adaptiveMinT<ANY>(own.mint, own.ox, own.oy, own.oz);
        if (have && !live) { DHit miss; miss.t = kInf; miss.u = 0.f; miss.v = 0.f; miss.id = kNoHit; storeTraceResult<OUT>(out, i, miss, tris); }
        uint32_t *perm = sstack, *cnt = sstack + BLOCK;
        if (threadIdx.x <= 64u) cnt[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t key = directionBin(own.dx, own.dy, own.dz);
        uint32_t pos = 0;
        if (live) pos = atomicAdd(&cnt[key], 1u);
        __syncthreads();
        if (threadIdx.x < 64u) {                                /* first wave: exclusive prefix sums of the 64 counters */
            const uint32_t c = cnt[threadIdx.x];
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((int)threadIdx.x >= d) incl += o; }
            cnt[threadIdx.x] = incl - c;
            if (threadIdx.x == 63u) cnt[64] = incl;
        }
        __syncthreads();
        const uint32_t base = cnt[key], total = cnt[64];
        if (live) perm[base + pos] = threadIdx.x;
        __syncthreads();
        if (total == 0u) return;                                /* block-uniform */
        mine = threadIdx.x < total;
        if (mine) {
            const uint32_t src = perm[threadIdx.x];
            r.ray = (uint32_t)((uint64_t)blockIdx.x * BLOCK + src);
            ray = loadRay<SHARE>(rays, dirs, stride, r.ray);
        }
    } else if (FIRST) {
#else
    if (FIRST && !ANY) {
        const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
        mine = i < n;
        if (mine) { ray = loadRay<SHARE>(rays, dirs, stride, i); r.ray = (uint32_t)i; }
    } else if (FIRST) {
#endif
        /* Shadow rays: a quarter of them is dead on C2 (shading queued nothing for the sample: maxt < mint).  Dead rays get
         * their miss right here, and when that empties at least one wave the live rays of the block are packed into its
         * first waves: a wave full of holes runs as long as a full one, an empty wave leaves at once (C2 shadow stage 37.5
         * -> 32.9 ms).  For BSDF rays (13 % dead) the same costs more than it saves (71.3 -> 72.5 ms).  perm / wcount
         * borrow the stack's LDS. */
        const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
        DRay own = ray;
        const bool have = i < n;
        if (have) own = loadRay<SHARE>(rays, dirs, stride, i);
        const bool live = have && own.maxt > adaptiveMinT<ANY>(own.mint, own.ox, own.oy, own.oz);   /* what Walk::begin tests */
        if (have && !live) { DHit miss; miss.t = kInf; miss.u = 0.f; miss.v = 0.f; miss.id = kNoHit; storeTraceResult<OUT>(out, i, miss, tris); }
        uint32_t *perm = sstack, *wcount = sstack + BLOCK;
        const unsigned long long lm = __ballot(live);
        const uint32_t wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63u) == 0u) wcount[wave] = (uint32_t)__popcll(lm);
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (uint32_t wv = 0; wv < BLOCK / 64; ++wv) { const uint32_t c = wcount[wv]; if (wv < wave) base += c; total += c; }
        if (live) perm[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))] = threadIdx.x;
        __syncthreads();
        if (total == 0u) return;                                /* block-uniform */
        if (total > BLOCK - 64u) {
            /* packing would not empty a wave: every lane keeps its own ray (and the order the shading kernel wrote them in) */
            mine = live;
            if (mine) { ray = own; r.ray = (uint32_t)i; }
        } else {
            mine = threadIdx.x < total;
            if (mine) {
                const uint32_t src = perm[threadIdx.x];
                r.ray = (uint32_t)((uint64_t)blockIdx.x * BLOCK + src);
                ray = loadRay<SHARE>(rays, dirs, stride, r.ray);
            }
        }
    } else {
        /* block -> (shard, chunk of BLOCK records of that shard); the grid is sized for full shards */
        const uint32_t shard = blockIdx.x % kContShards, chunk = blockIdx.x / kContShards;
        const uint32_t nParked = min(in.count[shard * 32u], in.perShard);
        if (chunk * BLOCK >= nParked) return;                   /* block-uniform */
        const uint32_t e = chunk * BLOCK + threadIdx.x;
        mine = e < nParked;
        inSlot = (size_t)shard * in.perShard + e;
        if (mine) { r = in.recs[inSlot]; ray = loadRay<SHARE>(rays, dirs, stride, r.ray); }
    }
    uint32_t nn = 0, nt = 0;
    const void *triTable = PROJ ? (const void *)projTris : (const void *)tris;
    Walk<ANY, COUNT, PROJ, STACK, BLOCK> w;
    uint32_t ovf[kStackOverflow];
    w.begin(ray);
    const bool traced = FIRST && mine && !w.finished();
    if (!FIRST) {
        w.resume(r.cur, r.pend, (int)r.sp, r.tmax, r.u, r.v, r.id, (mine && r.id != kNoHit) ? tris[r.id].prim : 0u);
        if (mine) {
            stackToLds<STACK, BLOCK>(sstack, (int)threadIdx.x, in.stacks + inSlot * ((STACK + 3) / 4));
        }
    }
    bool parked = false;
    int budget = next.iterCap;
#if KKD_REPACK_AT > 0
    /* Experiment: after KKD_REPACK_AT steps the rays of the workgroup that are still walking are moved together into its first
     * waves (the others have their result stored and their waves leave), so that the rest of the round does not issue for
     * lanes that are done.  A ray keeps its column of the LDS stack (Walk::col): only nine words of state travel, through a
     * 4 KB buffer in three passes; the ray itself is read again. */
    constexpr int RPW = KKD_REPACK_WORDS;            /* words per lane and pass of the exchange buffer */
    __shared__ uint32_t sx[RPW * BLOCK];
    if (FIRST && next.iterCap > KKD_REPACK_AT) {
        w.run(nodes, triTable, sstack, ovf, nn, nt, KKD_REPACK_AT);
        budget -= KKD_REPACK_AT;
        const bool live = mine && !w.finished();
        /* a stack that has spilled into the lane's private array cannot travel: such a workgroup (rare) walks on as it is */
        if (!__syncthreads_or(live && w.sp > STACK)) {
        if (mine && !live) { storeTraceResult<OUT>(out, r.ray, w.hit, tris); mine = false; }
        const unsigned long long lm = __ballot(live);
        const uint32_t wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63u) == 0u) sx[wave] = (uint32_t)__popcll(lm);
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (uint32_t wv = 0; wv < BLOCK / 64; ++wv) { const uint32_t c = sx[wv]; if (wv < wave) base += c; total += c; }
        __syncthreads();
        if (total == 0u) {                                       /* block-uniform: every ray of the workgroup is done */
            if (ctr) { ctr = counterShard(ctr); if (FIRST) countRays(traced, &ctr->rays[cls]); if (COUNT) countTraversal(nn, nt, ctr, cls); }
            return;
        }
        const uint32_t dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
        const bool take = threadIdx.x < total;
        uint32_t g[10] = { r.ray, w.cur, w.pend, (uint32_t)w.sp, __float_as_uint(w.tmax), __float_as_uint(w.hit.u), __float_as_uint(w.hit.v), w.hit.id, w.hprim, (uint32_t)w.col };
#pragma unroll
        for (int pass = 0; pass < (10 + RPW - 1) / RPW; ++pass) {
            if (live) { for (int j = 0; j < RPW; ++j) if (RPW * pass + j < 10) sx[j * BLOCK + dst] = g[RPW * pass + j]; }
            __syncthreads();
            if (take) { for (int j = 0; j < RPW; ++j) if (RPW * pass + j < 10) g[RPW * pass + j] = sx[j * BLOCK + threadIdx.x]; }
            __syncthreads();
        }
        mine = take;
        if (mine) {
            r.ray = g[0];
            ray = loadRay<SHARE>(rays, dirs, stride, r.ray);
            w.begin(ray);
            w.resume(g[1], g[2], (int)g[3], __uint_as_float(g[4]), __uint_as_float(g[5]), __uint_as_float(g[6]), g[7], g[8]);
            w.col = (int)g[9];
        } else w.stop();
        }
    }
#endif
    if (next.iterCap > 0) {
        w.run(nodes, triTable, sstack, ovf, nn, nt, budget);
        const bool unfinished = !w.finished();
        if (__ballot(unfinished) != 0ull) {
            const bool want = unfinished && w.sp <= STACK;      /* a stack that has spilled into private memory stays here */
            const unsigned long long m = __ballot(want);
            const uint32_t shard = blockIdx.x % kContShards;
            uint32_t base = 0;
            if (m != 0ull) {
                const int leader = __ffsll((long long)m) - 1;
                if ((int)(threadIdx.x & 63) == leader) {
                    base = atomicAdd(next.count + shard * 32u, (uint32_t)__popcll(m));
                }
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            }
            const uint32_t rank = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            parked = want && rank < next.perShard;              /* a full shard: walk on instead */
            if (FIRST && ctr) {
                /* what steers the budgets is the rays that WERE parked: not the lanes a full shard turned away, nor those whose
                 * stack had spilled */
                const unsigned long long pm = __ballot(parked);
                if (pm != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)pm) - 1) atomicAdd(&counterShard(ctr)->parked[cls], (unsigned long long)__popcll(pm));
            }
            const size_t slot = (size_t)shard * next.perShard + rank;
            if (parked) {
                ContRec o; o.ray = r.ray; o.cur = w.cur; o.sp = (uint32_t)w.sp; o.tmax = w.tmax; o.pend = w.pend; o.u = w.hit.u; o.v = w.hit.v; o.id = w.hit.id;
                next.recs[slot] = o;
                stackFromLds<STACK, BLOCK>(sstack, w.col, next.stacks + slot * ((STACK + 3) / 4));
                w.stop();
            }
        }
    }
    if (__ballot(!w.finished()) != 0ull) w.run(nodes, triTable, sstack, ovf, nn, nt, 1 << 24);
    if (mine && !parked) storeTraceResult<OUT>(out, r.ray, w.hit, tris);
    if (ctr) {
        ctr = counterShard(ctr);
        if (FIRST) countRays(traced, &ctr->rays[cls]);
        if (COUNT) countTraversal(nn, nt, ctr, cls);
    }
}

/* The LAST round, as persistent waves.  A wave draws the records of its shard's queue itself (consume counter = word 2 of the
 * shard's counter line, zero since the launcher's memset); whenever KKD_TAIL_REFILL of its lanes are done it stores their
 * results and hands them new records, so that the heavy tail of one ray does not keep 63 lanes idle: the parked rays are
 * few, long and unlike each other, and a wave of 64 of them runs several times longer than its mean ray (13-16 of 64 lanes
 * active per instruction in the plain resume kernel).  Every wave leaves: the queue is finite (a wave that finds it empty
 * stops drawing), each walk is finite, and the outer loop is bounded. */
#ifndef KKD_TAIL_REFILL
#define KKD_TAIL_REFILL 24
#endif
#ifndef KKD_TAIL_CHUNK
#define KKD_TAIL_CHUNK 8
#endif
template <bool ANY, bool COUNT, int OUT, bool PROJ, int STACK, int BLOCK, bool SHARE = false>
__global__ void __launch_bounds__(BLOCK, 8) kTraceTail(const Node4 *__restrict__ nodes, const LeafTri *__restrict__ tris, const ProjTri *__restrict__ projTris,
                                                       const DRay *__restrict__ rays, const float4 *__restrict__ dirs, uint32_t stride, void *__restrict__ out, DeviceCounters *ctr, int cls, TraceContinuation in) {
    __shared__ uint32_t sstack[STACK * BLOCK];
    const uint32_t shard = blockIdx.x % kContShards;
    const uint32_t nParked = min(in.count[shard * 32u], in.perShard);
    uint32_t *consume = in.count + shard * 32u + 2u;
    const void *triTable = PROJ ? (const void *)projTris : (const void *)tris;
    Walk<ANY, COUNT, PROJ, STACK, BLOCK> w;
    uint32_t ovf[kStackOverflow];
    DRay ray; ray.maxt = -1.f; ray.mint = 0.f; ray.ox = ray.oy = ray.oz = 0.f; ray.dx = ray.dy = 0.f; ray.dz = 1.f;
    w.begin(ray);                       /* finished: maxt < mint */
    uint32_t myRay = 0; bool have = false, empty = nParked == 0u;
    uint32_t nn = 0, nt = 0;
    for (int outer = 1 << 22; outer > 0; --outer) {
        const unsigned long long idle = __ballot(w.finished());
        const int nIdle = __popcll(idle);
        if (!empty && (nIdle >= KKD_TAIL_REFILL || nIdle == 64)) {
            if (have && w.finished()) { storeTraceResult<OUT>(out, myRay, w.hit, tris); have = false; }
            uint32_t base = 0;
            const int leader = __ffsll((long long)idle) - 1;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(consume, (uint32_t)nIdle);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (base + (uint32_t)nIdle >= nParked) empty = true;              /* wave-uniform */
            if (w.finished()) {
                const uint32_t e = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (e < nParked) {
                    const size_t slot = (size_t)shard * in.perShard + e;
                    const ContRec r = in.recs[slot];
                    ray = loadRay<SHARE>(rays, dirs, stride, r.ray); myRay = r.ray; have = true;
                    w.begin(ray);
                    w.resume(r.cur, r.pend, (int)r.sp, r.tmax, r.u, r.v, r.id, r.id != kNoHit ? tris[r.id].prim : 0u);
                    stackToLds<STACK, BLOCK>(sstack, (int)threadIdx.x, in.stacks + slot * ((STACK + 3) / 4));
                }
            }
        }
        if (__ballot(!w.finished()) == 0ull) { if (empty) break; else continue; }
        w.run(nodes, triTable, sstack, ovf, nn, nt, empty ? (1 << 24) : KKD_TAIL_CHUNK);
    }
    if (have) storeTraceResult<OUT>(out, myRay, w.hit, tris);
    if (ctr && COUNT) countTraversal(nn, nt, counterShard(ctr), cls);
}

/* ---- camera rays that leave the scene: filtered environment lookup (kk/direct.cpp:55-57 -> envmap.cpp:379-409) -------------
 * Runs after kShade (which has stored alpha and zero radiance for these samples) when the scene has an environment emitter
 * and emitters are not hidden. */
__global__ void __launch_bounds__(256) kEnvMiss(DScene sc, BatchParams bp, WaveBuffers wb) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= bp.count) return;
    if (wb.hits[i].id != kNoHit) return;
    const SampleId s = mapWorkItem(bp, i);
    if (!s.active) return;
    float sx, sy;
    const DRay ray = primaryRay(sc, bp, s, sx, sy, bp.sampler);
    const f3 d(ray.dx, ray.dy, ray.dz); f3 rxD, ryD;
    sensorDifferentials(sc.sensor, sx, sy, bp.diffScale, d, rxD, ryD);
    const f3 e = envEvalFiltered(sc.env, d, rxD, ryD);
    float4 a = wb.accum[i]; a.x = e.x; a.y = e.y; a.z = e.z;
    wb.accum[i] = a;
}

/* ---- K4: shade ------------------------------------------------------------------------------------ */
__device__ __forceinline__ void killRay(DRay &r) { r.ox = r.oy = r.oz = 0.f; r.dx = r.dy = 0.f; r.dz = 1.f; r.mint = 0.f; r.maxt = -1.f; }
__device__ __forceinline__ void storeRay(DRay *dst, const DRay &r) {
    float4 *p = reinterpret_cast<float4 *>(dst);
    p[0] = make_float4(r.ox, r.oy, r.oz, r.mint); p[1] = make_float4(r.dx, r.dy, r.dz, r.maxt);
}

/* kShade's rays: whole, or (BatchParams::splitRays) direction and maxt only -- the origin is the sample's, stored once */
__device__ __forceinline__ void storeRayOrDir(DRay *plane, const BatchParams &bp, size_t slot, const DRay &r) {
    if (bp.splitRays) reinterpret_cast<float4 *>(plane)[slot] = make_float4(r.dx, r.dy, r.dz, r.maxt);
    else storeRay(plane + slot, r);
}

#ifndef KKD_SHADE_WAVES
#define KKD_SHADE_WAVES 4      /* blocks of 256 per CU = waves per SIMD */
#endif
#ifndef KKD_SHADE_BLOCK
#define KKD_SHADE_BLOCK 256    /* (one-wave workgroups were measured: with two batch lanes kShade then finishes in 52 instead of 130 ms per
                                  64-spp C4 frame, but kResolve is starved instead, 120 instead of 42 ms: the frame time does not move, 198.0 vs
                                  197.6 ms -- the pipeline as a whole is bound by VALU issue, lanes only redistribute it) */
#endif
template <uint32_t LD>  /* KKD_SAMPLER_* in use */
__global__ void __launch_bounds__(KKD_SHADE_BLOCK, KKD_SHADE_WAVES * 256 / KKD_SHADE_BLOCK) kShade(DScene sc, BatchParams bp, WaveBuffers wb) {
    const uint32_t i = blockIdx.x * KKD_SHADE_BLOCK + threadIdx.x;
    if (i >= bp.count) return;
    const SampleId s = mapWorkItem(bp, i);
    const uint32_t nB = bp.nB, nL = bp.nL;
    float4 accum = make_float4(0.f, 0.f, 0.f, 1.f);
    const DHit hit = wb.hits[i];
    bool shade = false;
    Its its; DRay ray; float sx, sy;
    if (s.active) {
        ray = primaryRay(sc, bp, s, sx, sy, LD);
        if (hit.id == kNoHit) {                                    /* kk/direct.cpp:55-60 */
            if (sc.sensor.filmHasAlpha) accum.w = 0.f;
            /* (the environment seen by a camera ray that leaves the scene is added by kEnvMiss, which runs after this kernel) */
        } else shade = true;
    }
    if (!shade) {
        DRay dead; killRay(dead);
        if (bp.splitRays) wb.origins[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t b = 0; b < nB; ++b) { storeRayOrDir(wb.bsdfRays, bp, (size_t)b * bp.stride + i, dead); wb.brec0[(size_t)b * bp.stride + i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        for (uint32_t l = 0; l < nL; ++l) { storeRayOrDir(wb.shadowRays, bp, (size_t)l * bp.stride + i, dead); wb.lrec[(size_t)l * bp.stride + i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        /* a camera ray that left the scene: kEnvMiss adds the environment to what is stored here */
        if (bp.integrator == KKD_INTEGRATOR_PATH || s.active) wb.accum[i] = accum;
        if (bp.integrator != KKD_INTEGRATOR_PATH) wb.aflags[i] = s.active ? 1u : 0u;
        return;
    }
    const f3 rd(ray.dx, ray.dy, ray.dz);
    fillIntersection<true>(sc, rd, hit, its);
    const bool seesEmitter = its.emitter >= 0 && !bp.hideEmitters;
    if (seesEmitter) {                                               /* :66-67 */
        const f3 le = areaEval(sc.emitters[its.emitter], its.sh.n, -rd);
        accum.x += le.x; accum.y += le.y; accum.z += le.z;
    }
    const uint32_t btype = sc.materials[its.material].typeFlags;
    const f3 refN = (btype & (ETransmission | EBackSide)) ? f3(0.f) : its.sh.n;   /* DirectSamplingRecord(its) */
    if (bp.splitRays) wb.origins[i] = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);   /* every ray of the sample starts here (scene.cpp:840-841 for the shadow rays) */
    const float cB = fdiv((float)nB, (float)(nB + nL)), cL = fdiv((float)nL, (float)(nB + nL));            /* :163-166 */
    const float invNL = frcp((float)nL);
    /* sibling integrator on the same kernels: MIDirectIntegrator::Li (integrators/direct/direct.cpp:176-311) draws the
     * emitter samples before the BSDF samples, applies the geometric-normal tests only with 'strictNormals', samples
     * emitters only for smooth BSDFs and weights with the power heuristic */
    const bool stock = bp.integrator != KKD_INTEGRATOR_KK;      /* the path plugin's first vertex is the direct plugin's with one sample each */
    const bool geoTest = !stock || bp.strictNormals;
    const bool dead = (stock && bp.strictNormals && dot(its.geoN, rd) * its.wi.z >= 0.f)   /* direct.cpp:177-190, path.cpp:156-165 */
                      || (bp.integrator == KKD_INTEGRATOR_PATH && bp.maxDepth == 1);
    float pathEta = 1.0f;

    /* integrateBrdf :74-104 -- the part before the ray is traced */
    for (uint32_t b = 0; b < nB; ++b) {
        /* kk: BSDF samples are requested first (kk/direct.cpp:44-47, 80-85), the stock plugins ask for the emitter samples first
         * (direct.cpp:139-143, 219); a count of one is a plain next2D, more are a slice of a requested array */
        const uint32_t flatB = 2u + 2u * b + (stock ? 2u * nL : 0u);
        const f2 u = nB > 1u ? draw2DArray(bp, s, flatB, (stock && nL > 1u) ? 1u : 0u, nB, b, LD)
                             : draw2D(bp, s, flatB, 1u + ((stock && nL == 1u) ? 1u : 0u), LD);
        f3 woLocal; float bsdfPdf; uint32_t sampledType;
        woLocal = f3(0.f); bsdfPdf = 0.f; sampledType = 0u;
        const f3 bsdfVal = ablation(bp, 8u) ? f3(0.f) : sampleBsdf(sc, its.material, its.wi, u, woLocal, bsdfPdf, sampledType);   /* ($KKD_DEBUG bit 8, ablation build: no BSDF sampling) */
        DRay r; killRay(r);
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!isZero(bsdfVal) && !dead) {
            const f3 wo = its.sh.toWorld(woLocal);
            if (!geoTest || dot(its.geoN, wo) * woLocal.z > 0.f) {   /* :96-99; direct.cpp:267-271 */
                r.ox = its.p.x; r.oy = its.p.y; r.oz = its.p.z; r.mint = kEpsilon;
                r.dx = wo.x; r.dy = wo.y; r.dz = wo.z; r.maxt = kInf;
                rec = make_float4(bsdfVal.x, bsdfVal.y, bsdfVal.z, (sampledType & EDelta) ? -bsdfPdf : bsdfPdf);
                if (sampledType == EDeltaTransmission) { const float e = sc.materials[its.material].eta; pathEta = its.wi.z > 0.f ? e : frcp(e); }   /* bRec.eta, dielectric.cpp:300-308 */
            }
        }
        storeRayOrDir(wb.bsdfRays, bp, (size_t)b * bp.stride + i, r);
        wb.brec0[(size_t)b * bp.stride + i] = rec;
        /* kResolve needs refN only for the sign test of the emitters' pdfDirect on this ray: one byte instead of the normal */
        if (bp.integrator != KKD_INTEGRATOR_PATH) wb.bflags[(size_t)b * bp.stride + i] = dot(f3(r.dx, r.dy, r.dz), refN) >= 0.f ? 1u : 0u;
    }
    if (bp.integrator == KKD_INTEGRATOR_PATH) wb.refN[i] = make_float4(refN.x, refN.y, refN.z, pathEta);

    /* integrateLight :127-160 -- everything but the visibility test, already MIS-weighted (:172-177) */
    for (uint32_t l = 0; l < nL; ++l) {
        DRay r; killRay(r);
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((stock ? (btype & ESmooth) != 0u && !dead : btype != 0u) && sc.nEmitters && !ablation(bp, 16u)) {      /* :140; direct.cpp:215 ($KKD_DEBUG bit 16, ablation build: no emitter sampling) */
            const uint32_t flatL = 2u + 2u * l + (stock ? 0u : 2u * nB);
            const f2 u = nL > 1u ? draw2DArray(bp, s, flatL, (!stock && nB > 1u) ? 1u : 0u, nL, l, LD)
                                 : draw2D(bp, s, flatL, 1u + ((!stock && nB == 1u) ? 1u : 0u), LD);
            DirectRec dRec; dRec.ref = its.p; dRec.refN = refN; dRec.emitter = -1;
            f3 value;
            if (ablation(bp, 128u)) {     /* ($KKD_DEBUG bit 128, ablation build: a fixed direction instead of the emitter sample) */
                dRec.d = normalize(f3(0.3f + u.x * 0.01f, 0.9f, 0.2f + u.y * 0.01f)); dRec.dist = 100.f; dRec.pdf = 1.f; dRec.measure = 0; dRec.emitter = 0;
                dRec.p = dRec.ref + dRec.d * 100.f; dRec.n = -dRec.d; value = f3(1.f);
            } else value = sampleEmitterDirectUnoccluded(sc, dRec, u);
            if (!isZero(value)) {
                const f3 woLocal = its.sh.toLocal(dRec.d);
                /* value and density of the BSDF for this pair of directions share their half vector, D, G1 (kkd_bsdf.h: evalPdfBsdf) */
                float bsdfPdfBoth = 0.5f;
                const f3 bsdfVal = ablation(bp, 32u) ? f3(0.5f) : evalPdfBsdf(sc, its.material, its.wi, woLocal, bsdfPdfBoth);   /* ($KKD_DEBUG bit 32: no BSDF eval / pdf for the emitter sample) */
                if (!isZero(bsdfVal) && (!geoTest || dot(its.geoN, dRec.d) * woLocal.z > 0.f)) {   /* :154; direct.cpp:233-234 */
                    const bool onSurface = sc.emitters[dRec.emitter].type != KKD_EMITTER_POINT;
                    const float bsdfPdf = onSurface ? bsdfPdfBoth : 0.f;
                    const float pa = cL * dRec.pdf, pb = cB * bsdfPdf;
                    const float weight = stock ? fdiv(pa * pa, pa * pa + pb * pb) : fdiv(pa, pb + pa);   /* direct.cpp:241-243, 313-316 */
                    const f3 contrib = value * bsdfVal * (weight * invNL);
                    rec = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
                    r.ox = its.p.x; r.oy = its.p.y; r.oz = its.p.z; r.mint = kEpsilon;   /* scene.cpp:840-841 */
                    r.dx = dRec.d.x; r.dy = dRec.d.y; r.dz = dRec.d.z; r.maxt = dRec.dist * (1.f - kShadowEpsilon);
                }
            }
        }
        storeRayOrDir(wb.shadowRays, bp, (size_t)l * bp.stride + i, r);
        wb.lrec[(size_t)l * bp.stride + i] = rec;
    }
    /* most camera rays hit a surface that does not emit: their (0, 0, 0, 1) is a flag byte, not 16 bytes written here and read by kResolve */
    if (bp.integrator == KKD_INTEGRATOR_PATH || seesEmitter) wb.accum[i] = accum;
    if (bp.integrator != KKD_INTEGRATOR_PATH) wb.aflags[i] = seesEmitter ? 1u : 0u;
}

/* ---- K4b: next vertex of the path plugin (MIPathTracer::Li, integrators/path/path.cpp:135-287) -------------------------------
 * One launch per depth >= 2, after the BSDF and shadow rays of the previous vertex have been traced: adds the emitter
 * sample of the previous vertex if unoccluded, the emitter (or environment) the BSDF ray reached with its MIS weight
 * (:226-264), plays Russian roulette (:276-286), then shades the vertex the BSDF ray hit exactly like kShade does for the
 * first one: emitter sample (:172-200) and BSDF sample (:206-225), both scaled by the path throughput.  Path state lives
 * in slots (rays, records, refN + eta, sample index); radiance is summed per sample in accum.  A launch either rewrites its
 * slots in place (ended paths get dead rays and zero records) or, `compact`, writes the paths that go on densely into the
 * other set of arrays.  Sample dimensions of the vertex at depth d: 2 + 6 (d - 1) emitter sample, + 2 BSDF sample, + 4
 * Russian roulette (after that vertex). */
__global__ void __launch_bounds__(256, 4) kPathBounce(DScene sc, BatchParams bp, WaveBuffers wb, WaveBuffers wo, PathLaunch pl) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const bool valid = j < pl.nIn;
    bool goesOn = false, need = false;
    uint32_t i = 0;
    DRay nr, sr; killRay(nr); killRay(sr);
    float4 nrec = make_float4(0.f, 0.f, 0.f, 0.f), nl = make_float4(0.f, 0.f, 0.f, 0.f), nrefN = make_float4(0.f, 0.f, 0.f, 1.f);
    if (valid) {
        i = pl.identity ? j : wb.sampleIdx[j];
        const SampleId s = mapWorkItem(bp, i);
        float4 acc = wb.accum[i];
        {   /* emitter sample of the previous vertex */
            const float4 lr = wb.lrec[j];
            if (wb.occluded[j] == 0u) { acc.x += lr.x; acc.y += lr.y; acc.z += lr.z; }
        }
        const float4 rec = wb.brec0[j];
        if (s.active && rec.w != 0.f) {
            DHit h = wb.bsdfHits[j];
            const DRay r = wb.bsdfRays[j];
            const float4 rn = wb.refN[j];
            const f3 rd(r.dx, r.dy, r.dz);
            f3 T(rec.x, rec.y, rec.z);
            float eta = rn.w;
            const float bsdfPdf = fabsf(rec.w);
            DirectRec dRec; dRec.ref = f3(r.ox, r.oy, r.oz); dRec.refN = f3(rn.x, rn.y, rn.z); dRec.emitter = -1;
            Its its;
            bool cont = true, hitEmitter = false;
            f3 value(0.f);
            float envPdf = 0.f; bool envLooked = false;   /* the environment's density in this direction, looked up together with its value */
            if (h.id == kNoHit) {                                                        /* :233-248 */
                cont = false;
                if (sc.envEmitter >= 0) {
                    envEvalWithPdf(sc.env, rd, value, envPdf); envLooked = true;
                    const f3 center(sc.env.center[0], sc.env.center[1], sc.env.center[2]);
                    float nearT, farT;
                    if (sphereRayIntersect(center, sc.env.radius, dRec.ref, rd, nearT, farT) && !(nearT > 0.f) && !(farT < 0.f)) hitEmitter = true;
                    /* (a direction the lookup refuses leaves value = 0 and envPdf = 0, as envEval and envPdfDirection do) */
                }
            } else {                                                                     /* :226-232 */
                h.id &= ~kHitNotEmitter;
                fillIntersection<true>(sc, rd, h, its);
                if (its.emitter >= 0) {
                    value = areaEval(sc.emitters[its.emitter], its.sh.n, -rd);
                    dRec.p = its.p; dRec.n = its.sh.n; dRec.d = rd; dRec.dist = h.t; dRec.measure = 0; dRec.emitter = its.emitter;
                    hitEmitter = true;
                }
            }
            if (hitEmitter) {                                                            /* :257-264 */
                const float lumPdf = rec.w > 0.f ? (envLooked ? envPdf * (sc.emitters[sc.envEmitter].weight * sc.emitterNormalization) : pdfEmitterDirect(sc, dRec)) : 0.f;
                const float w = fdiv(bsdfPdf * bsdfPdf, bsdfPdf * bsdfPdf + lumPdf * lumPdf);
                const f3 c = T * value * w;
                acc.x += c.x; acc.y += c.y; acc.z += c.z;
            }
            if (pl.terminate) cont = false;
            if (cont && (int)bp.depth - 1 >= bp.rrDepth) {                               /* :276-286, depth before the increment */
                const float q = fminf(maxComp(T) * eta * eta, 0.95f);
                const float u = draw1D(bp, s, 2u + 6u * (bp.depth - 2u) + 4u, bp.depth - 1u - (uint32_t)bp.rrDepth, bp.sampler);   /* the roulette is the path's only 1D request */
                if (u >= q) cont = false; else T = T / q;
            }
            if (cont && ((bp.maxDepth > 0 && (int)bp.depth >= bp.maxDepth) || (bp.strictNormals && dot(rd, its.geoN) * its.wi.z >= 0.f))) cont = false;   /* :156-165 */
            if (cont) {
                const uint32_t dim0 = 2u + 6u * (bp.depth - 1u);
                const uint32_t btype = sc.materials[its.material].typeFlags;
                const f3 refN = (btype & (ETransmission | EBackSide)) ? f3(0.f) : its.sh.n;
                nrefN = make_float4(refN.x, refN.y, refN.z, eta);
                if ((btype & ESmooth) != 0u && sc.nEmitters) {                           /* :172-200 */
                    const f2 u = draw2D(bp, s, dim0, 1u + 2u * (bp.depth - 1u), bp.sampler);
                    DirectRec lRec; lRec.ref = its.p; lRec.refN = refN; lRec.emitter = -1;
                    const f3 lval = sampleEmitterDirectUnoccluded(sc, lRec, u);
                    if (!isZero(lval)) {
                        const f3 woLocal = its.sh.toLocal(lRec.d);
                        float bsdfPdfBoth;
                        const f3 bsdfVal = evalPdfBsdf(sc, its.material, its.wi, woLocal, bsdfPdfBoth);   /* value and density from one pass (kkd_bsdf.h) */
                        if (!isZero(bsdfVal) && (!bp.strictNormals || dot(its.geoN, lRec.d) * woLocal.z > 0.f)) {
                            const bool onSurface = sc.emitters[lRec.emitter].type != KKD_EMITTER_POINT;
                            const float bp2 = onSurface ? bsdfPdfBoth : 0.f;
                            const float w = fdiv(lRec.pdf * lRec.pdf, lRec.pdf * lRec.pdf + bp2 * bp2);
                            const f3 c = T * lval * bsdfVal * w;
                            nl = make_float4(c.x, c.y, c.z, 0.f);
                            sr.ox = its.p.x; sr.oy = its.p.y; sr.oz = its.p.z; sr.mint = kEpsilon;
                            sr.dx = lRec.d.x; sr.dy = lRec.d.y; sr.dz = lRec.d.z; sr.maxt = lRec.dist * (1.f - kShadowEpsilon);
                            need = true;
                        }
                    }
                }
                const f2 u = draw2D(bp, s, dim0 + 2u, 2u + 2u * (bp.depth - 1u), bp.sampler);             /* :206-225 */
                f3 woLocal; float pdf; uint32_t sampledType;
                const f3 bsdfWeight = sampleBsdf(sc, its.material, its.wi, u, woLocal, pdf, sampledType);
                if (!isZero(bsdfWeight)) {
                    const f3 wo = its.sh.toWorld(woLocal);
                    if (!bp.strictNormals || dot(its.geoN, wo) * woLocal.z > 0.f) {
                        nr.ox = its.p.x; nr.oy = its.p.y; nr.oz = its.p.z; nr.mint = kEpsilon;
                        nr.dx = wo.x; nr.dy = wo.y; nr.dz = wo.z; nr.maxt = kInf;
                        const f3 nT = T * bsdfWeight;                                     /* :252-253 */
                        nrec = make_float4(nT.x, nT.y, nT.z, (sampledType & EDelta) ? -pdf : pdf);
                        if (sampledType == EDeltaTransmission) { const float e = sc.materials[its.material].eta; nrefN.w = eta * (its.wi.z > 0.f ? e : frcp(e)); }
                        goesOn = true; need = true;
                    }
                }
            }
        }
        wb.accum[i] = acc;
    }
    if (pl.terminate) return;
    const unsigned long long m = __ballot(need);
    uint32_t slot = j;
    if (pl.compact) {
        /* dense slots for the paths that go on: one atomic per block on the slot counter */
        __shared__ uint32_t waveCount[4], blockBase;
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        if (lane == 0) waveCount[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t total = waveCount[0] + waveCount[1] + waveCount[2] + waveCount[3]; blockBase = total ? atomicAdd(pl.counters + kAliveShards, total) : 0u; }
        __syncthreads();
        uint32_t base = blockBase;
        for (uint32_t w = 0; w < wave; ++w) base += waveCount[w];
        slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    } else if ((threadIdx.x & 63) == 0 && m) atomicAdd(pl.counters + (blockIdx.x % kAliveShards), (uint32_t)__popcll(m));
    if (pl.compact ? need : valid) {
        storeRay(wo.bsdfRays + slot, nr); wo.brec0[slot] = nrec; wo.refN[slot] = nrefN;
        storeRay(wo.shadowRays + slot, sr); wo.lrec[slot] = nl;
        wo.sampleIdx[slot] = i;
    }
}

/* ---- filtered splat of the G samples a lane group holds of one pixel (ImageBlock::put, reduced across the group) -----------------------
 * Every lane holds one sample of pixel (X, Y); its footprint lies inside the (2B+1)^2 window around the pixel, i.e.
 * NV = (2B+1)^2 * 5 film words.  The words are summed over the 64 lanes by a halving butterfly: at each of the six
 * steps a lane hands one half of the words it still holds to a partner and adds the partner's other half, so 64 words
 * cost 63 exchanges (v_permlane32_swap / v_permlane16_swap / DPP row_mirror, row_half_mirror, quad_perm) instead of
 * 64 x 6, and lane l ends up owning the total of word (pass * 64 + l): one global_atomic_add_f32 per lane and pass,
 * contiguous within a film row. */
__device__ __forceinline__ void swap32(float &a, float &b) {   /* a.lanes[32:63] <-> b.lanes[0:31] */
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) {   /* odd rows of a <-> even rows of b */
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int CTRL> __device__ __forceinline__ float dppMove(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
template <int N, int CTRL> __device__ __forceinline__ void butterflyDpp(float *a, bool upper) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float send = upper ? a[i] : a[i + N], keep = upper ? a[i + N] : a[i];
        a[i] = keep + dppMove<CTRL>(send);
    }
}

/* G = lanes per pixel (a power of two, 8..64): the butterfly starts at the stage that halves a G-lane group, every group
 * of the wave reduces its own pixel's window and lane l of a group ends up owning word (pass * G + l). */
/* BLOCK4 (G = 64, spp a multiple of 256): the four waves of a block hold samples of the SAME pixel, so their totals are
 * added in LDS (sred: 4 x NPASS x 64 words) and only the first wave issues atomics -- a quarter of the film atomics, which
 * queue up on the pixel's 125 addresses when many waves share a pixel (C4 at 2048 spp: resolve 1059 -> 738 ms per frame). */
template <int B, int G, bool BLOCK4 = false>
__device__ __forceinline__ void splatGroup(float *__restrict__ film, const float *sfilter, float radius, float scale,
                                           int X, int Y, int W, int H, float posx, float posy, const float *val, bool noAtomics, float *sred = nullptr) {
    constexpr int WIN = 2 * B + 1, NV = WIN * WIN * 5, NPASS = (NV + G - 1) / G;
    const int lane = threadIdx.x & 63;
    const int lox = (int)ceilf(posx - radius), hix = (int)floorf(posx + radius);
    const int loy = (int)ceilf(posy - radius), hiy = (int)floorf(posy + radius);
    float fx[WIN], fy[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        const int xi = X - B + i, yi = Y - B + i;
        fx[i] = (xi >= lox && xi <= hix) ? sfilter[min((int)fabsf(((float)xi - posx) * scale), 31)] : 0.f;
        fy[i] = (yi >= loy && yi <= hiy) ? sfilter[min((int)fabsf(((float)yi - posy) * scale), 31)] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        float a[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = pass * G + i;
            a[i] = k < NV ? fx[(k / 5) % WIN] * fy[(k / 5) / WIN] * val[k % 5] : 0.f;
        }
        if (G >= 64) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { swap32(a[i], a[i + 32]); a[i] += a[i + 32]; }
        }
        if (G >= 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { swap16(a[i], a[i + 16]); a[i] += a[i + 16]; }
        }
        if (G >= 16) butterflyDpp<8, 0x140>(a, (lane & 8) != 0);   /* row_mirror: partner differs in bit 3          */
        butterflyDpp<4, 0x141>(a, (lane & 4) != 0);                /* row_half_mirror: partner differs in bit 2     */
        butterflyDpp<2, 0x4E>(a, (lane & 2) != 0);                 /* quad_perm [2,3,0,1]                           */
        butterflyDpp<1, 0xB1>(a, (lane & 1) != 0);                 /* quad_perm [1,0,3,2]                           */
        const int k = pass * G + (lane & (G - 1));
        if (BLOCK4) { sred[((threadIdx.x >> 6) * NPASS + pass) * 64 + lane] = a[0]; continue; }
        if (k < NV && a[0] != 0.f) {
            const int c = k % 5, pix = k / 5, xo = X - B + pix % WIN, yo = Y - B + pix / WIN;
            if (xo >= 0 && xo < W && yo >= 0 && yo < H) { if (!noAtomics) atomicAdd(film + ((size_t)yo * W + xo) * 5 + c, a[0]); else if (a[0] == 12345.678f) film[0] = a[0]; }
        }
    }
    if (BLOCK4) {
        __syncthreads();                       /* block-uniform path: every wave of the block is here */
        if (threadIdx.x < 64) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int k = pass * 64 + lane;
                const float v = (sred[(0 * NPASS + pass) * 64 + lane] + sred[(1 * NPASS + pass) * 64 + lane])
                              + (sred[(2 * NPASS + pass) * 64 + lane] + sred[(3 * NPASS + pass) * 64 + lane]);
                if (k < NV && v != 0.f) {
                    const int c = k % 5, pix = k / 5, xo = X - B + pix % WIN, yo = Y - B + pix / WIN;
                    if (xo >= 0 && xo < W && yo >= 0 && yo < H) { if (!noAtomics) atomicAdd(film + ((size_t)yo * W + xo) * 5 + c, v); else if (v == 12345.678f) film[0] = v; }
                }
            }
        }
    }
}

/* ---- K5: resolve + splat ---------------------------------------------------------------------------- */
template <bool INDEP>   /* as kPrimary: the independent sampler without the others' code */
__global__ void __launch_bounds__(256) kResolve(DScene sc, BatchParams bp, WaveBuffers wb, float *__restrict__ film, DeviceCounters *ctr) {
    __shared__ float sfilter[32];
    __shared__ float sred[4 * 4 * 64];         /* splatGroup<.., BLOCK4>: 4 waves x up to 4 passes x 64 words */
    if (threadIdx.x < 32) sfilter[threadIdx.x] = sc.filter.values[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const SampleId s = mapWorkItem(bp, i);
    const uint32_t nB = bp.nB, nL = bp.nL;
    float val[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
    float sx = 0.f, sy = 0.f;
    if (s.active) {
        const f2 jit = draw2D(bp, s, 0u, 0u, INDEP ? KKD_SAMPLER_INDEPENDENT : bp.sampler);
        sx = (float)s.X + jit.x; sy = (float)s.Y + jit.y;
        const float4 acc = (bp.integrator == KKD_INTEGRATOR_PATH || wb.aflags[i]) ? wb.accum[i] : make_float4(0.f, 0.f, 0.f, 1.f);
        f3 L(acc.x, acc.y, acc.z);
        const float cB = fdiv((float)nB, (float)(nB + nL)), cL = fdiv((float)nL, (float)(nB + nL));
        const float invNB = frcp((float)nB);
        for (uint32_t b = 0; b < nB; ++b) {                        /* kk/direct.cpp:105-123 + :167-171 */
            const float4 rec = wb.brec0[(size_t)b * bp.stride + i];
            if (rec.w == 0.f || ablation(bp, 2u)) continue;
            DHit h = wb.bsdfHits[(size_t)b * bp.stride + i];
            if (h.id != kNoHit && (h.id & kHitNotEmitter)) continue;   /* hit a surface that does not emit: Le = 0 */
            if (h.id == kNoHit && sc.envEmitter < 0) continue;
            DRay r;
            if (bp.splitRays) {
                const float4 o = wb.origins[i], d = reinterpret_cast<const float4 *>(wb.bsdfRays)[(size_t)b * bp.stride + i];
                r.ox = o.x; r.oy = o.y; r.oz = o.z; r.mint = o.w; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.maxt = d.w;
            } else r = wb.bsdfRays[(size_t)b * bp.stride + i];
            const f3 rd(r.dx, r.dy, r.dz);
            DirectRec dRec; dRec.ref = f3(r.ox, r.oy, r.oz);
            /* pdfDirect only asks for the sign of dot(d, refN): the direct plugins carry that bit (kShade), the path plugin the normal */
            if (bp.integrator == KKD_INTEGRATOR_PATH) { const float4 rn = wb.refN[i]; dRec.refN = f3(rn.x, rn.y, rn.z); }
            else dRec.refN = wb.bflags[(size_t)b * bp.stride + i] ? rd : -rd;
            f3 value;
            float lightPdf;
            if (h.id != kNoHit) {
                Its bIts; fillIntersection<false>(sc, rd, h, bIts);
                if (bIts.emitter < 0) continue;
                value = areaEval(sc.emitters[bIts.emitter], bIts.sh.n, -rd);
                dRec.p = bIts.p; dRec.n = bIts.sh.n; dRec.d = rd; dRec.dist = h.t; dRec.measure = 0; dRec.emitter = bIts.emitter;
                lightPdf = rec.w > 0.f ? pdfEmitterDirect(sc, dRec) : 0.f;
            } else {
                if (sc.envEmitter < 0) continue;
                /* value and density of the environment in this direction share their lookup (pdfEmitterDirect's environment
                 * branch, scene.cpp:949-952 + envmap.cpp:603-633, written out) */
                float envPdf;
                envEvalWithPdf(sc.env, rd, value, envPdf);
                const f3 center(sc.env.center[0], sc.env.center[1], sc.env.center[2]);   /* envmap.cpp:358-374 */
                float nearT, farT;
                if (!sphereRayIntersect(center, sc.env.radius, dRec.ref, rd, nearT, farT) || nearT > 0.f || farT < 0.f) continue;
                lightPdf = rec.w > 0.f ? envPdf * (sc.emitters[sc.envEmitter].weight * sc.emitterNormalization) : 0.f;
            }
            const float bsdfPdf = fabsf(rec.w);
            const float pa = cB * bsdfPdf, pb = cL * lightPdf;
            const float weight = bp.integrator != KKD_INTEGRATOR_KK ? fdiv(pa * pa, pa * pa + pb * pb) : fdiv(pa, pa + pb);   /* direct.cpp:303-305, path.cpp:263 */
            L += value * f3(rec.x, rec.y, rec.z) * (weight * invNB);
        }
        for (uint32_t l = 0; l < nL; ++l) {
            if (wb.occluded[(size_t)l * bp.stride + i] != 0u) continue;     /* (the record of an occluded sample is not read) */
            const float4 rec = wb.lrec[(size_t)l * bp.stride + i];
            L.x += rec.x; L.y += rec.y; L.z += rec.z;
        }
        /* ImageBlock::put rejects samples with negative or non-finite components */
        const bool ok = isfinite(L.x) && isfinite(L.y) && isfinite(L.z) && L.x >= 0.f && L.y >= 0.f && L.z >= 0.f;
        if (ok) { val[0] = L.x; val[1] = L.y; val[2] = L.z; val[3] = acc.w; val[4] = 1.f; }
    }
    /* ---- filtered splat, reduced across the g lanes that share this pixel ---- */
    const int b = sc.filter.border, win = 2 * b + 1, nVals = win * win * 5;
    const float radius = sc.filter.radius, scale = sc.filter.scaleFactor;
    const float posx = sx - 0.5f, posy = sy - 0.5f;
    const int lox = (int)ceilf(posx - radius), hix = (int)floorf(posx + radius);
    const int loy = (int)ceilf(posy - radius), hiy = (int)floorf(posy + radius);
    const uint32_t g = bp.g, j = threadIdx.x & (g - 1u);
    /* every lane of a group maps to the same pixel, also the inactive ones */
    const int X = s.X, Y = s.Y, W = bp.W, H = bp.H;
    if (g >= 8u && b <= 3) {   /* wave-uniform: groups of g lanes hold the samples of one pixel each */
        const bool na = ablation(bp, 1u);
        /* four waves per pixel: g = 64 lanes per group and the number of groups per pixel a multiple of 4 (batches start
         * at multiples of 256 samples, so a block never straddles two pixels then) */
        const bool block4 = g == 64u && (bp.chunks & 3u) == 0u && !ablation(bp, 4u);
#define KKD_SPLAT(BB) switch (g) { \
            case 64u: if (block4) splatGroup<BB, 64, true>(film, sfilter, radius, scale, X, Y, W, H, posx, posy, val, na, sred); \
                      else splatGroup<BB, 64>(film, sfilter, radius, scale, X, Y, W, H, posx, posy, val, na); break; \
            case 32u: splatGroup<BB, 32>(film, sfilter, radius, scale, X, Y, W, H, posx, posy, val, na); break; \
            case 16u: splatGroup<BB, 16>(film, sfilter, radius, scale, X, Y, W, H, posx, posy, val, na); break; \
            default: splatGroup<BB, 8>(film, sfilter, radius, scale, X, Y, W, H, posx, posy, val, na); break; }
        switch (b) {
            case 0: KKD_SPLAT(0) break;
            case 1: KKD_SPLAT(1) break;
            case 2: KKD_SPLAT(2) break;
            default: KKD_SPLAT(3) break;
        }
#undef KKD_SPLAT
        if (ctr) countRays(s.active, &counterShard(ctr)->samples);
        return;
    }
    float keep = 0.f; int k = 0;
    for (int wy = 0; wy < win; ++wy) {
        const int yi = Y - b + wy;
        const float fy = (yi >= loy && yi <= hiy) ? sfilter[min((int)fabsf(((float)yi - posy) * scale), 31)] : 0.f;
        for (int wx = 0; wx < win; ++wx) {
            const int xi = X - b + wx;
            const float fx = (xi >= lox && xi <= hix) ? sfilter[min((int)fabsf(((float)xi - posx) * scale), 31)] : 0.f;
            const float w = fx * fy;
#pragma unroll
            for (int c = 0; c < 5; ++c, ++k) {
                float v = w * val[c];
                for (uint32_t m = 1; m < g; m <<= 1) v += __shfl_xor(v, (int)m);
                const uint32_t kk = (uint32_t)k & (g - 1u);
                if (kk == j) keep = v;
                if (kk == g - 1u || k == nVals - 1) {
                    const int kmine = (k - (int)kk) + (int)j;
                    if (kmine < nVals && keep != 0.f) {
                        const int c2 = kmine % 5, pix = kmine / 5, wx2 = pix % win, wy2 = pix / win;
                        const int xo = X - b + wx2, yo = Y - b + wy2;
                        if (xo >= 0 && xo < W && yo >= 0 && yo < H)
                            atomicAdd(film + ((size_t)yo * W + xo) * 5 + c2, keep);
                    }
                    keep = 0.f;
                }
            }
        }
    }
    if (ctr) countRays(s.active, &counterShard(ctr)->samples);
}

/* ---- film develop: Bitmap::convert(ESpectrumAlphaWeight -> fmt), src/libcore/fmtconv.cpp:955-1034 ----------------
 * One pixel per lane: 20 bytes in, 4-16 bytes out, HBM streaming.  Operation order of the reference, unfused. */
template <int FMT>
__global__ void __launch_bounds__(256) kDevelop(const float *__restrict__ film, uint64_t n, float *__restrict__ out) {
#pragma clang fp contract(off)
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = film + i * 5;
    const float r = p[0], g = p[1], b = p[2], a = p[3], w = p[4];
    const float inv = (w != 0.f) ? 1.0f / w : w;
    constexpr int C = (FMT == KKD_PIXEL_LUMINANCE) ? 1 : (FMT == KKD_PIXEL_LUMINANCE_ALPHA) ? 2 : (FMT == KKD_PIXEL_RGB || FMT == KKD_PIXEL_XYZ) ? 3 : 4;
    float *o = out + i * C;
    if (FMT == KKD_PIXEL_LUMINANCE || FMT == KKD_PIXEL_LUMINANCE_ALPHA) {
        o[0] = (r * 0.212671f + g * 0.715160f + b * 0.072169f) * inv;      /* spec.getLuminance() * invWeight */
        if (FMT == KKD_PIXEL_LUMINANCE_ALPHA) o[1] = a * inv;
    } else if (FMT == KKD_PIXEL_RGB || FMT == KKD_PIXEL_RGBA) {
        o[0] = r * inv; o[1] = g * inv; o[2] = b * inv;                     /* Spectrum(spec * invWeight).toLinearRGB */
        if (FMT == KKD_PIXEL_RGBA) o[3] = a * inv;
    } else {
        const float x = r * inv, y = g * inv, z = b * inv;                  /* Spectrum(spec * invWeight * 1).toXYZ, spectrum.cpp:229-234 */
        o[0] = x * 0.412453f + y * 0.357580f + z * 0.180423f;
        o[1] = x * 0.212671f + y * 0.715160f + z * 0.072169f;
        o[2] = x * 0.019334f + y * 0.119193f + z * 0.950227f;
        if (FMT == KKD_PIXEL_XYZA) o[3] = a * inv;
    }
}

/* ---- film merge of multi-device renders ---------------------------------------------------------------------------- */
template <bool UNPACK>
__global__ void __launch_bounds__(256) kTilesPackUnpack(float *__restrict__ film, float *__restrict__ packed, const uint32_t *__restrict__ tileList,
                                                        uint32_t nTiles, TileGrid g) {
    const uint32_t side = g.bs + 2u * (uint32_t)g.border, perTile = side * side;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;          /* one thread per packed pixel (5 floats) */
    if (i >= (uint64_t)nTiles * perTile) return;
    const uint32_t k = (uint32_t)(i / perTile), r = (uint32_t)(i - (uint64_t)k * perTile), py = r / side, px = r - py * side;
    const uint32_t tile = tileList[k], ty = tile / g.nbx, tx = tile - ty * g.nbx;
    const int x = (int)(tx * g.bs + px) - g.border - g.edge, y = (int)(ty * g.bs + py) - g.border - g.edge;
    bool inside = x >= 0 && x < g.W && y >= 0 && y < g.H;
    if (!UNPACK && inside) {
        /* the owner of the pixel and its eight neighbours are the only tiles whose regions can cover it (border < bs) */
        const int ox = (x + g.edge) / (int)g.bs, oy = (y + g.edge) / (int)g.bs;
        for (int dy = -1; dy <= 1 && inside; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = ox + dx, ny = oy + dy;
                if (nx < 0 || ny < 0 || nx >= (int)g.nbx || ny >= (int)g.nby) continue;
                const uint32_t other = (uint32_t)ny * g.nbx + (uint32_t)nx;
                if (other >= tile || g.rankOf[other] != g.myRank) continue;
                const int x0 = nx * (int)g.bs - g.edge - g.border, y0 = ny * (int)g.bs - g.edge - g.border, side2 = (int)g.bs + 2 * g.border;
                if (x >= x0 && x < x0 + side2 && y >= y0 && y < y0 + side2) { inside = false; break; }
            }
    }
    float *p = packed + i * 5;
    float *f = film + ((size_t)(inside ? y : 0) * g.W + (inside ? x : 0)) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        if (UNPACK) { if (inside && p[c] != 0.f) atomicAdd(f + c, p[c]); }      /* borders of neighbouring tiles overlap */
        else p[c] = inside ? f[c] : 0.f;
    }
}

__global__ void __launch_bounds__(256) kAddFilm(float *__restrict__ dst, const float *__restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

/* ---- host launchers ---------------------------------------------------------------------------------- */
static inline uint32_t blocksFor(uint64_t n, uint32_t block) { return (uint32_t)((n + block - 1) / block); }

void launchPrimary(hipStream_t st, bool count, const DScene &sc, const BatchParams &bp, DHit *hits, DeviceCounters *ctr) {
    const bool packet = !ablation(bp, 64u);
    const dim3 grid(blocksFor(bp.count, 256 * (packet ? KKD_PRIMARY_RAYS : 1))), block(256);     /* $KKD_DEBUG bit 64: per-lane walk for primary rays (A/B measurements) */
#define KKD_LP(C, P, K) do { if (bp.sampler == KKD_SAMPLER_INDEPENDENT) hipLaunchKernelGGL((kPrimary<C, P, K, true, kStackLds, 256>), grid, block, 0, st, sc, bp, hits, ctr); \
                             else hipLaunchKernelGGL((kPrimary<C, P, K, false, kStackLds, 256>), grid, block, 0, st, sc, bp, hits, ctr); } while (0)
    if (sc.projTris) {
        if (count) { if (packet) KKD_LP(true, true, true); else KKD_LP(true, true, false); }
        else { if (packet) KKD_LP(false, true, true); else KKD_LP(false, true, false); }
    } else {
        if (count) { if (packet) KKD_LP(true, false, true); else KKD_LP(true, false, false); }
        else { if (packet) KKD_LP(false, false, true); else KKD_LP(false, false, false); }
    }
#undef KKD_LP
}

template <bool ANY, bool COUNT, int OUT>
static void launchTraceT(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, void *out, DeviceCounters *ctr, int cls) {
    constexpr bool canShare = (ANY && OUT == 2) || (!ANY && OUT == 0);       /* only the render path's rays come without origins */
#define KKD_LT(P, S) hipLaunchKernelGGL((kTrace<ANY, COUNT, OUT, P, (ANY ? kStackLdsAny : kStackLds), 256, S>), dim3(blocksFor(n, 256)), dim3(256), 0, st, sc.nodes, sc.tris, sc.projTris, rays.rays, rays.dirs, rays.stride, n, out, ctr, cls)
    if (canShare && rays.dirs) { if (sc.projTris) KKD_LT(true, canShare); else KKD_LT(false, canShare); }
    else { if (sc.projTris) KKD_LT(true, false); else KKD_LT(false, false); }
#undef KKD_LT
}

void launchTrace(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, bool anyHit, bool count, int outMode, void *out, DeviceCounters *ctr, int cls) {
    if (n == 0) return;
    if (anyHit) {
        if (outMode == 2) { if (count) launchTraceT<true, true, 2>(st, sc, rays, n, out, ctr, cls); else launchTraceT<true, false, 2>(st, sc, rays, n, out, ctr, cls); }
        else { if (count) launchTraceT<true, true, 1>(st, sc, rays, n, out, ctr, cls); else launchTraceT<true, false, 1>(st, sc, rays, n, out, ctr, cls); }
    } else {
        if (outMode == 0) { if (count) launchTraceT<false, true, 0>(st, sc, rays, n, out, ctr, cls); else launchTraceT<false, false, 0>(st, sc, rays, n, out, ctr, cls); }
        else { if (count) launchTraceT<false, true, 1>(st, sc, rays, n, out, ctr, cls); else launchTraceT<false, false, 1>(st, sc, rays, n, out, ctr, cls); }
    }
}

template <bool ANY, bool COUNT, int OUT, bool PROJ, bool SHARE = false>
static void launchPhasedT(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, void *out, DeviceCounters *ctr, int cls, const TraceContinuation &cont) {
    /* rounds: [0, cap) on all rays, then (rounds > 2 only) cap more steps on what is left, ...; the last round walks to the end
     * as persistent waves that draw the parked rays themselves (kTraceTail).  With more than two rounds the two halves of the
     * continuation buffer alternate as input and output; with two the first round parks into the whole buffer. */
    static const int tailBlocksAll = [] { const char *e = getenv("KKD_TAIL_PERSIST"); return e ? atoi(e) : 1; }();   /* blocks per shard; 0: developer switch, plain resume kernel */
    static const int tailBlocksAny = [] { const char *e = getenv("KKD_TAIL_PERSIST_ANY"); return e ? atoi(e) : -1; }();   /* the same for shadow rays only (A/B runs) */
    const int tailBlocks = (ANY && tailBlocksAny >= 0) ? tailBlocksAny : tailBlocksAll;
    const int rounds = cont.rounds < 2 ? 2 : cont.rounds;
    TraceContinuation bufs[2] = { cont, cont };
    if (rounds > 2) {
        const uint32_t half = cont.perShard / 2;
        bufs[0].perShard = bufs[1].perShard = half;
        bufs[1].recs = cont.recs + (size_t)kContShards * half; bufs[1].stacks = cont.stacks + (size_t)kContShards * half * (((ANY ? kStackLdsAny : kStackLds) + 3) / 4);
    }
    bufs[1].count = cont.count + kContShards * 32;
    hipMemsetAsync(cont.count, 0, 2 * kContShards * 32 * sizeof(uint32_t), st);   /* both counter arrays of this ray kind (fill counters and kTraceTail's consume counters) */
    /* a shard cannot hold more than its share of the rays of this launch */
    const uint64_t perShardMax = std::min<uint64_t>(bufs[0].perShard, (n + kContShards - 1) / kContShards + 256);
    const dim3 resumeGrid(kContShards * blocksFor(perShardMax, 256)), block(256);
    for (int r = 0; r < rounds; ++r) {
        TraceContinuation in = bufs[(r + 1) & 1], next = bufs[r & 1];
        next.iterCap = (r + 1 < rounds) ? (r == 0 ? cont.iterCap : cont.iterCapLater) : 0;
        if (r >= 2 && next.iterCap > 0) hipMemsetAsync(next.count, 0, kContShards * 32 * sizeof(uint32_t), st);
        if (r == 0) hipLaunchKernelGGL((kTraceRound<ANY, COUNT, OUT, PROJ, true, (ANY ? kStackLdsAny : kStackLds), KKD_FIRST_BLOCK, SHARE>), dim3(blocksFor(n, KKD_FIRST_BLOCK)), dim3(KKD_FIRST_BLOCK), 0, st,
                                       sc.nodes, sc.tris, sc.projTris, rays.rays, rays.dirs, rays.stride, n, out, ctr, cls, in, next);
        else if (next.iterCap == 0 && tailBlocks > 0)
            hipLaunchKernelGGL((kTraceTail<ANY, COUNT, OUT, PROJ, (ANY ? kStackLdsAny : kStackLds), 256, SHARE>), dim3(kContShards * (uint32_t)std::min(tailBlocks, 8)), block, 0, st,
                               sc.nodes, sc.tris, sc.projTris, rays.rays, rays.dirs, rays.stride, out, ctr, cls, in);
        else hipLaunchKernelGGL((kTraceRound<ANY, COUNT, OUT, PROJ, false, (ANY ? kStackLdsAny : kStackLds), 256, SHARE>), resumeGrid, block, 0, st,
                                sc.nodes, sc.tris, sc.projTris, rays.rays, rays.dirs, rays.stride, n, out, ctr, cls, in, next);
    }
}

void launchTracePhased(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, bool anyHit, bool count, void *out, DeviceCounters *ctr,
                       int cls, const TraceContinuation &cont) {
    if (n == 0) return;
    const bool proj = sc.projTris != nullptr;
#define KKD_PH(A, C, O, P) launchPhasedT<A, C, O, P>(st, sc, rays, n, out, ctr, cls, cont)
    if (anyHit && rays.dirs) {
        if (count) { if (proj) launchPhasedT<true, true, 2, true, true>(st, sc, rays, n, out, ctr, cls, cont); else launchPhasedT<true, true, 2, false, true>(st, sc, rays, n, out, ctr, cls, cont); }
        else { if (proj) launchPhasedT<true, false, 2, true, true>(st, sc, rays, n, out, ctr, cls, cont); else launchPhasedT<true, false, 2, false, true>(st, sc, rays, n, out, ctr, cls, cont); }
    } else if (rays.dirs) {
        if (count) { if (proj) launchPhasedT<false, true, 0, true, true>(st, sc, rays, n, out, ctr, cls, cont); else launchPhasedT<false, true, 0, false, true>(st, sc, rays, n, out, ctr, cls, cont); }
        else { if (proj) launchPhasedT<false, false, 0, true, true>(st, sc, rays, n, out, ctr, cls, cont); else launchPhasedT<false, false, 0, false, true>(st, sc, rays, n, out, ctr, cls, cont); }
    } else if (anyHit) {
        if (count) { if (proj) KKD_PH(true, true, 2, true); else KKD_PH(true, true, 2, false); }
        else { if (proj) KKD_PH(true, false, 2, true); else KKD_PH(true, false, 2, false); }
    } else {
        if (count) { if (proj) KKD_PH(false, true, 0, true); else KKD_PH(false, true, 0, false); }
        else { if (proj) KKD_PH(false, false, 0, true); else KKD_PH(false, false, 0, false); }
    }
#undef KKD_PH
}

void launchShade(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &wb) {
    const dim3 grid(blocksFor(bp.count, KKD_SHADE_BLOCK)), block(KKD_SHADE_BLOCK);
    if (bp.sampler == KKD_SAMPLER_LDSAMPLER) hipLaunchKernelGGL(kShade<KKD_SAMPLER_LDSAMPLER>, grid, block, 0, st, sc, bp, wb);
    else if (bp.sampler == KKD_SAMPLER_STRATIFIED) hipLaunchKernelGGL(kShade<KKD_SAMPLER_STRATIFIED>, grid, block, 0, st, sc, bp, wb);
    else if (bp.sampler >= KKD_SAMPLER_HALTON && bp.sampler <= KKD_SAMPLER_HAMMERSLEY_PLAIN) hipLaunchKernelGGL(kShade<KKD_SAMPLER_HALTON>, grid, block, 0, st, sc, bp, wb);
    else if (bp.sampler == KKD_SAMPLER_SOBOL) hipLaunchKernelGGL(kShade<KKD_SAMPLER_SOBOL>, grid, block, 0, st, sc, bp, wb);
    else hipLaunchKernelGGL(kShade<KKD_SAMPLER_INDEPENDENT>, grid, block, 0, st, sc, bp, wb);
    if (!bp.hideEmitters && sc.envEmitter >= 0) hipLaunchKernelGGL(kEnvMiss, dim3(blocksFor(bp.count, 256)), dim3(256), 0, st, sc, bp, wb);
}

void launchPathBounce(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &in, const WaveBuffers &out, const PathLaunch &pl) {
    if (pl.nIn == 0) return;
    hipLaunchKernelGGL(kPathBounce, dim3(blocksFor(pl.nIn, 256)), dim3(256), 0, st, sc, bp, in, out, pl);
}

void launchDevelop(hipStream_t st, const float *film, uint64_t n, int fmt, float *out) {
    if (n == 0) return;
    const dim3 grid(blocksFor(n, 256)), block(256);
    switch (fmt) {
        case KKD_PIXEL_LUMINANCE: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_LUMINANCE>, grid, block, 0, st, film, n, out); break;
        case KKD_PIXEL_LUMINANCE_ALPHA: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_LUMINANCE_ALPHA>, grid, block, 0, st, film, n, out); break;
        case KKD_PIXEL_RGB: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_RGB>, grid, block, 0, st, film, n, out); break;
        case KKD_PIXEL_RGBA: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_RGBA>, grid, block, 0, st, film, n, out); break;
        case KKD_PIXEL_XYZ: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_XYZ>, grid, block, 0, st, film, n, out); break;
        default: hipLaunchKernelGGL(kDevelop<KKD_PIXEL_XYZA>, grid, block, 0, st, film, n, out); break;
    }
}

void launchPackTiles(hipStream_t st, const float *film, float *packed, const uint32_t *tileList, uint32_t nTiles, const TileGrid &g) {
    const uint64_t side = g.bs + 2u * (uint32_t)g.border, n = (uint64_t)nTiles * side * side;
    if (n) hipLaunchKernelGGL(kTilesPackUnpack<false>, dim3(blocksFor(n, 256)), dim3(256), 0, st, const_cast<float *>(film), packed, tileList, nTiles, g);
}

void launchUnpackAddTiles(hipStream_t st, float *film, const float *packed, const uint32_t *tileList, uint32_t nTiles, const TileGrid &g) {
    const uint64_t side = g.bs + 2u * (uint32_t)g.border, n = (uint64_t)nTiles * side * side;
    if (n) hipLaunchKernelGGL(kTilesPackUnpack<true>, dim3(blocksFor(n, 256)), dim3(256), 0, st, film, const_cast<float *>(packed), tileList, nTiles, g);
}

void launchAddFilm(hipStream_t st, float *dst, const float *src, uint64_t n) {
    if (n) hipLaunchKernelGGL(kAddFilm, dim3(blocksFor(n, 256)), dim3(256), 0, st, dst, src, n);
}

void launchResolve(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &wb, float *film, DeviceCounters *ctr) {
    if (bp.sampler == KKD_SAMPLER_INDEPENDENT) hipLaunchKernelGGL(kResolve<true>, dim3(blocksFor(bp.count, 256)), dim3(256), 0, st, sc, bp, wb, film, ctr);
    else hipLaunchKernelGGL(kResolve<false>, dim3(blocksFor(bp.count, 256)), dim3(256), 0, st, sc, bp, wb, film, ctr);
}

} /* namespace kkd */
