/* kkd_kernels.h -- launch interface between the C-ABI host code (kkd_api.hip) and the gfx950 kernels. */
#pragma once
#include <hip/hip_runtime.h>
#include "kkd_types.h"

namespace kkd {

/* per ray class (KKD_RAYS_PRIMARY / _BSDF / _SHADOW) */
/* traversal stack capacity per lane: kStackLds entries in LDS + kStackOverflow in private memory (kkd_trace.h) */
static constexpr int kMaxTraversalStack = 12 + 80;   /* the shallower of the two walks: shadow rays keep 12 entries in LDS (kkd_trace.h), the others 16 */

/* One word takes only ~88 M atomic adds per second, far fewer than the waves a render retires, so the counters are
 * sharded: a wave adds into shard (blockIdx.x % kCounterShards), each shard on its own 128-byte line; the host sums. */
static constexpr int kCounterShards = 1024;
struct alignas(128) DeviceCounters { unsigned long long samples, pad; unsigned long long rays[3], nodes[3], tris[3]; unsigned long long parked[3]; /* rays the first traversal round parked, per class */ unsigned long long pad2[2]; };

/* One batch = work items [w0, w0 + count) of the render; see mapWorkItem() in kkd_kernels.hip. */
struct BatchParams {
    uint64_t w0; uint64_t seed;
    uint32_t count, stride;          /* stride: element pitch between the per-sample planes of multi-sample buffers */
    uint32_t g, gShift, chunks, spp, sampleFirst;
    /* position of the batch's first lane group, split on the host so that the device needs 32-bit arithmetic only:
     * group(w0) = (tile0 * bs^2 + pin0) * chunks + chunk0 */
    uint32_t chunk0, pin0, tile0, chunksShift /* log2(chunks) or 32 */, pixShift /* log2(bs^2) or 32 */;
    uint32_t nB, nL, hideEmitters;
    uint32_t splitRays;              /* direct plugins: BSDF and shadow rays are stored as float4 {direction, maxt} planes beside ONE plane of {shading point, mint} per sample (RaySource); path plugin: whole rays, its slots are compacted */
    uint32_t sampler, samplerDim, samplerSet;   /* KKD_SAMPLER_*; 'dimension'; size of the per-pixel point sets (ldsampler: a power of two, stratified: samplerRes^2) */
    uint32_t samplerRes;             /* stratified: cells per side (stratified.cpp:64-74) */
    HaltonSetup halton;              /* KKD_SAMPLER_HALTON / _HALTON_PLAIN / _HAMMERSLEY* */
    SobolSetup sobol;                /* KKD_SAMPLER_SOBOL */
    float diffScale;                 /* 1 / sqrt(spp): RayDifferential::scaleDifferential of the camera ray (integrator.cpp:144-145, 181) */
    uint32_t integrator, strictNormals;   /* KKD_INTEGRATOR_*; 'strictNormals' of the stock direct / path plugins */
    int32_t maxDepth, rrDepth;            /* path plugin: 'maxDepth', 'rrDepth' */
    uint32_t depth;                       /* path plugin: depth of the vertex a kPathBounce launch shades (2, 3, ...) */
    uint32_t bs, bsPow2, nbx, nby, tileIndex, tileCount, nTilesMine;
    int32_t W, H;
    int32_t edge;                    /* Film::hasHighQualityEdges: filter border by which the sampled area grows on every side, else 0 */
    /* tile sharding (tileCount > 1): the tiles of this context, row-major tile ids in the order they are rendered
     * (nTilesMine entries, device memory); nullptr = all tiles, tile id = running index.  The host deals the tiles out
     * round-robin along the Morton curve of the tile grid, so every rank's tiles are spread over the whole image. */
    const uint32_t *tileList;
    uint32_t debug;                  /* developer ablation bits from $KKD_DEBUG: 1 = no film atomics, 2 = no BSDF-hit evaluation, 4 = no block-level film reduction, 8 = no BSDF sampling, 16 = no emitter sampling, 32 = no BSDF eval / pdf of the emitter sample, 64 = per-lane primary walk, 128 = fixed direction instead of the emitter sample */
};

/* SoA work buffers of one batch (all device memory). */
struct WaveBuffers {
    DHit *hits;          /* primary hits                       [stride]      */
    float4 *accum;       /* Le / environment + alpha           [stride]      */
    uint8_t *aflags;     /* direct plugins: 1 = accum[i] is stored; 0 = it is (0, 0, 0, 1), the camera ray hit a surface that does not emit, and was not written [stride] */
    float4 *refN;        /* path plugin only: DirectSamplingRecord::refN; w = relative index of refraction along the path [stride] */
    uint8_t *bflags;     /* direct plugins: all a BSDF sample needs of refN, 1 = dot(wo, refN) >= 0 (the test of Emitter::pdfDirect of the shape emitters) [nB * stride] */
    DRay *bsdfRays;      /* splitRays: float4 {direction, maxt} instead [nB * stride] */
    float4 *origins;     /* splitRays: {shading point, mint}    [stride]      */
    DHit *bsdfHits;      /*                                    [nB * stride] */
    float4 *brec0;       /* bsdf weight rgb, +-pdf (<0: delta) [nB * stride] */
    DRay *shadowRays;    /* splitRays: float4 {direction, maxt} instead [nL * stride] */
    uint32_t *occluded;  /*                                    [nL * stride] */
    float4 *lrec;        /* MIS-weighted light contribution    [nL * stride] */
    uint32_t *sampleIdx; /* path plugin: sample (index into accum) of the path in this slot [stride] */
};

/* Where a traversal launch reads its rays.  dirs == nullptr: ray k = rays[k].  Otherwise (BatchParams::splitRays) ray k takes
 * direction and maxt from dirs[k] and origin and mint from origins[k mod stride], a float4 plane that `rays` points to: the
 * BSDF and shadow rays of a sample all start at its shading point, which kShade stores once. */
struct RaySource { const DRay *rays; const float4 *dirs; uint32_t stride; };

void launchPrimary(hipStream_t st, bool count, const DScene &sc, const BatchParams &bp, DHit *hits, DeviceCounters *ctr);
/* outMode: 0 = DHit with LeafTri slot, 1 = DHit with global primitive id, 2 = uint32 occlusion flags (anyHit only) */
/* rayClass: which DeviceCounters slot the launch accounts to */
void launchTrace(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, bool anyHit, bool count, int outMode, void *out, DeviceCounters *ctr, int rayClass);
inline void launchTrace(hipStream_t st, const DScene &sc, const DRay *rays, uint64_t n, bool anyHit, bool count, int outMode, void *out, DeviceCounters *ctr, int rayClass) {
    launchTrace(st, sc, RaySource{ rays, nullptr, 0u }, n, anyHit, count, outMode, out, ctr, rayClass);
}
/* A ray parked by kTraceRound after its iteration budget: everything Walk needs to go on (32 bytes + 16 stack words; the best
 * hit's distance is tmax). */
struct alignas(16) ContRec { uint32_t ray, cur, sp; float tmax; uint32_t pend; float u, v; uint32_t id; };
/* stacks: 16 words (= kStackLds, kkd_trace.h) per record */
/* The buffer is cut into kContShards regions of perShard records, each with its own fill counter on its own 128-byte
 * line (count[shard * 32]): one counter for all waves would serialise at ~88 M atomics/s, like the ray counters did. */
static constexpr uint32_t kContShards = 1024;
struct TraceContinuation { ContRec *recs; uint4 *stacks; uint32_t *count; uint32_t perShard; int iterCap /* first round */, iterCapLater, rounds; };
/* Two-launch variant of launchTrace for the render path (outMode 0 or 2): waves stop after cont.iterCap steps, the few lanes
 * still walking are parked in cont and finished by a second, densely packed launch.  cont.count must be zero on entry
 * (the launcher clears the counters). */
void launchTracePhased(hipStream_t st, const DScene &sc, const RaySource &rays, uint64_t n, bool anyHit, bool count, void *out, DeviceCounters *ctr,
                       int rayClass, const TraceContinuation &cont);
void launchShade(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &wb);
/* path plugin: resolve the segment traced last, play Russian roulette, shade the vertex at bp.depth; alive: kAliveShards counters of paths that go on */
static constexpr uint32_t kAliveShards = 64;
/* One kPathBounce launch over the nIn path slots of `in`.  identity: slot j holds sample j (right after kShade).  compact:
 * paths that go on (or still have a shadow ray pending) are written densely into `out` (other arrays than `in`), slots
 * handed out through counters[kAliveShards]; otherwise every slot is rewritten in place (out == in) and the paths that go
 * on are counted in counters[0 .. kAliveShards).  terminate: only add what the last segments found, shade nothing. */
struct PathLaunch { uint32_t nIn, identity, compact, terminate; uint32_t *counters; };
void launchPathBounce(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &in, const WaveBuffers &out, const PathLaunch &pl);
void launchDevelop(hipStream_t st, const float *film, uint64_t nPixels, int pixelFormat, float *out);
/* film merge of the tile-sharded multi-device render (BlockedRenderProcess::processResult -> Film::put(block),
 * reference src/librender/renderproc.cpp:142-149, src/films/hdrfilm.cpp:391-393): a rank's tiles, each with its filter
 * border, are packed into a dense buffer ((bs + 2 border)^2 * 5 floats per tile), moved to the root device in one copy
 * and added into the root's film there.  A rank renders all its tiles into ONE film, so where the regions of two of its
 * tiles overlap the film already holds their sum: such a pixel is packed only with the lowest-numbered tile of the rank
 * whose region covers it. */
struct TileGrid { uint32_t bs, nbx, nby; int32_t border, edge, W, H; const uint32_t *rankOf; /* rank of every tile (pack only) */ uint32_t myRank; };
void launchPackTiles(hipStream_t st, const float *film, float *packed, const uint32_t *tileList, uint32_t nTiles, const TileGrid &g);
void launchUnpackAddTiles(hipStream_t st, float *film, const float *packed, const uint32_t *tileList, uint32_t nTiles, const TileGrid &g);
/* dst[i] += src[i]: merge of whole films (sample-split mode without RCCL) */
void launchAddFilm(hipStream_t st, float *dst, const float *src, uint64_t n);
void launchResolve(hipStream_t st, const DScene &sc, const BatchParams &bp, const WaveBuffers &wb, float *film, DeviceCounters *ctr);

} /* namespace kkd */
