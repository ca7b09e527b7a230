/*
 * kkdirect.h -- C ABI of libkkdirect.so, the MI355X (gfx950) implementation of the
 * "kkIntegrator" direct-lighting path of akaterin/ray-tracer (a Mitsuba 0.5 fork).
 *
 * The boundary: the reference's plugin `plugins/kkIntegrator.so`
 * (src/integrators/SConscript:44, src/integrators/kk/direct.cpp:15-199) is a
 * SamplingIntegrator whose per-sample work is SamplingIntegrator::renderBlock
 * (src/librender/integrator.cpp:140-188) + DirectIntegrator::Li
 * (src/integrators/kk/direct.cpp:50-178).  A replacement plugin overrides
 * Integrator::render() (src/librender/integrator.cpp:95-129; precedent
 * src/integrators/kk/vcm.cpp:196-218), flattens the scene through the public
 * Scene/TriMesh/BSDF/Emitter API and calls the functions below.  Every entry
 * point cites the reference interface it replaces.
 *
 * Plain C: pointers and sizes only, no C++ types, no exceptions cross it.
 * All functions return KKD_OK (0) or a negative KKD_ERR_* code; the text of the
 * last error of a context is available from kkd_last_error().  One context is
 * used from one host thread at a time (kkd_request_cancel excepted).
 *
 * Conventions (same as the reference, SINGLE_PRECISION / SPECTRUM_SAMPLES=3):
 *   Float = float32, Spectrum = linear RGB triple, matrices row-major,
 *   film pixel = 5 floats {R*w, G*w, B*w, alpha*w, w}  (Bitmap::ESpectrumAlphaWeight,
 *   src/librender/renderproc.cpp:121, src/films/hdrfilm.cpp:352).
 */
#ifndef KKDIRECT_H
#define KKDIRECT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KKD_ABI_VERSION 1

/* ---- error codes ---------------------------------------------------------------- */
#define KKD_OK              0
#define KKD_ERR_INVALID    (-1)  /* bad argument / inconsistent scene description      */
#define KKD_ERR_DEVICE     (-2)  /* HIP runtime error (text in kkd_last_error)          */
#define KKD_ERR_CANCELLED  (-3)  /* kkd_request_cancel() was honoured                   */
#define KKD_ERR_NOMEM      (-4)
#define KKD_ERR_UNSUPPORTED (-5) /* feature outside the hot path (e.g. bsdfSamples==0)  */

/* ---- enums ----------------------------------------------------------------------- */
/* BSDF models on the path: src/bsdfs/{diffuse,roughconductor,roughplastic,dielectric,twosided}.cpp */
#define KKD_BSDF_DIFFUSE         0
#define KKD_BSDF_ROUGHCONDUCTOR  1
#define KKD_BSDF_ROUGHPLASTIC    2
#define KKD_BSDF_DIELECTRIC      3
#define KKD_BSDF_TWOSIDED        4

/* MicrofacetDistribution::EType, src/bsdfs/microfacet.h:48-57 */
#define KKD_DISTR_BECKMANN 0
#define KKD_DISTR_GGX      1
#define KKD_DISTR_PHONG    2

/* Emitters on the path: src/emitters/{area,point,envmap}.cpp */
#define KKD_EMITTER_AREA   0
#define KKD_EMITTER_POINT  1
#define KKD_EMITTER_ENVMAP 2

/* reconstruction filters: src/rfilters/ (tabulated by src/libcore/rfilter.cpp:37-55) */
#define KKD_RFILTER_GAUSSIAN   0
#define KKD_RFILTER_BOX        1
#define KKD_RFILTER_TENT       2
#define KKD_RFILTER_MITCHELL   3
#define KKD_RFILTER_CATMULLROM 4
#define KKD_RFILTER_LANCZOS    5

/* shape flags */
#define KKD_SHAPE_HAS_NORMALS   1u  /* per-vertex normals valid for this shape (TriMesh::m_normals != NULL) */
#define KKD_SHAPE_HAS_TEXCOORDS 2u  /* per-vertex uv valid (TriMesh::m_texcoords != NULL)                   */

/* ---- scene description ------------------------------------------------------------ */

/* One Mitsuba Shape after Scene::addShape expansion (src/librender/scene.cpp:558-591); the
 * order of this array is the ShapeKDTree::addShape order (src/librender/skdtree.cpp:49-66), so
 * global primitive id = first_prim + local primIndex (skdtree.cpp:57-58,69-70). */
typedef struct kkd_shape {
    uint32_t first_prim;    /* index of the shape's first triangle in kkd_scene_desc.indices   */
    uint32_t n_prims;       /* TriMesh::getTriangleCount()                                      */
    int32_t  material_id;   /* index into materials (Shape::getBSDF())                          */
    int32_t  emitter_id;    /* index into emitters or -1 (Shape::getEmitter())                  */
    uint32_t flags;         /* KKD_SHAPE_*                                                      */
    uint32_t reserved;
} kkd_shape;

/* BSDF parameter block.  Holds the *configured* state of the plugin, i.e. what the reference
 * serialises (diffuse.cpp:161-165, roughconductor.cpp:216-226, roughplastic.cpp:243-253,
 * dielectric.cpp:176-182, twosided.cpp:77-82); textures must be constant. */
typedef struct kkd_material {
    int32_t type;              /* KKD_BSDF_*                                                  */
    int32_t distribution;      /* KKD_DISTR_* (rough models)                                  */
    int32_t sample_visible;    /* MicrofacetDistribution::m_sampleVisible (default 1)         */
    int32_t nonlinear;         /* roughplastic 'nonlinear'                                    */
    int32_t nested[2];         /* twosided: material ids of m_nestedBRDF[0..1] (-1: same as 0) */
    float   alpha_u, alpha_v;  /* roughness (clamped to >=1e-4 by the library as microfacet.h:70-71) */
    float   eta;               /* dielectric / roughplastic: intIOR/extIOR                    */
    float   reflectance[3];            /* diffuse 'reflectance' / roughplastic 'diffuseReflectance' */
    float   specular_reflectance[3];   /* 'specularReflectance' (default 1)                   */
    float   specular_transmittance[3]; /* dielectric 'specularTransmittance' (default 1)      */
    float   cond_eta[3];       /* roughconductor eta / extEta  (roughconductor.cpp:189)       */
    float   cond_k[3];         /* roughconductor k   / extEta  (roughconductor.cpp:190)       */
} kkd_material;

/* Emitter record.  Area lights reference their parent shape (emitters/area.cpp:185-202). */
typedef struct kkd_emitter {
    int32_t type;              /* KKD_EMITTER_*                                               */
    int32_t shape_id;          /* area: parent shape index; else -1                           */
    float   sampling_weight;   /* Emitter::getSamplingWeight() (emitter.cpp:100-104)          */
    float   radiance[3];       /* area: 'radiance'; point: 'intensity'; envmap: unused        */
    float   position[3];       /* point: world position                                       */
    float   reserved;
} kkd_emitter;

/* Environment map (emitters/envmap.cpp).  Texels are the level-0 MIP array, linear RGB float32,
 * row-major, row 0 = +Y pole (theta = 0).  The library rounds them to half precision like the
 * reference's TMIPMap<Spectrum, SpectrumHalf> storage (envmap.cpp:102-103) and builds the MIP pyramid
 * of the filtered lookup for camera rays (envmap.cpp:166-182, 394-407) at upload. */
typedef struct kkd_envmap {
    uint32_t width, height;    /* 0,0 => no environment emitter                               */
    const float *texels;       /* width*height*3                                              */
    float   scale;             /* 'scale' (envmap.cpp:188)                                    */
    float   to_world[9];       /* rotation part of toWorld, row-major 3x3                     */
    float   bsphere_center[3]; /* m_sceneBSphere as set by EnvironmentMap::createShape        */
    float   bsphere_radius;    /* (envmap.cpp:331-341): 1.5 x radius of the scene AABB sphere */
} kkd_envmap;

/* Perspective sensor + film + filter (sensors/perspective.cpp:126-180, librender/sensor.cpp:84-108,
 * librender/film.cpp:25-96). */
typedef struct kkd_sensor {
    float   to_world[16];      /* camera-to-world, row-major 4x4, no scale (perspective.cpp:116-118) */
    float   xfov_deg;          /* PerspectiveCamera::getXFov()                                 */
    float   near_clip, far_clip;
    uint32_t film_width, film_height;       /* Film::getSize()                                 */
    uint32_t crop_offset_x, crop_offset_y;  /* Film::getCropOffset()                           */
    uint32_t crop_width, crop_height;       /* Film::getCropSize()                             */
    int32_t  film_has_alpha;   /* Film::hasAlpha() (integrator.cpp:159-160)                    */
    int32_t  rfilter_type;     /* KKD_RFILTER_*                                                */
    float    rfilter_radius;   /* gaussian: 4*stddev (=2); box: 0.5(+1e-5); tent 1; mitchell/catmullrom 2; lanczos: lobes */
    float    rfilter_param[2]; /* gaussian: stddev,-; mitchell: B,C; lanczos: lobes,-          */
    int32_t  film_high_quality_edges; /* Film::hasHighQualityEdges(): the sampled area grows by the filter border on every side
                                         (librender/film.cpp:54, librender/renderproc.cpp:166-171)  */
    int32_t  reserved;
} kkd_sensor;

typedef struct kkd_scene_desc {
    uint32_t abi_version;      /* = KKD_ABI_VERSION                                            */
    uint32_t reserved0;
    /* geometry: all TriMeshes concatenated, indices already offset into the global vertex arrays */
    uint64_t n_verts, n_tris;
    const float    *positions; /* n_verts*3  (TriMesh::getVertexPositions)                     */
    const float    *normals;   /* n_verts*3 or NULL (TriMesh::getVertexNormals)                */
    const float    *texcoords; /* n_verts*2 or NULL (TriMesh::getVertexTexcoords)              */
    const uint32_t *indices;   /* n_tris*3   (TriMesh::getTriangles)                           */
    uint32_t n_shapes;  const kkd_shape    *shapes;
    uint32_t n_materials; const kkd_material *materials;
    uint32_t n_emitters;  const kkd_emitter  *emitters;   /* Scene::getEmitters() order        */
    kkd_envmap envmap;         /* referenced by the emitter of type KKD_EMITTER_ENVMAP         */
    kkd_sensor sensor;
} kkd_scene_desc;

/* ---- context / render parameters ----------------------------------------------------- */

#define KKD_FLAG_COUNT_TRAVERSAL 1u  /* count visited nodes / tested triangles in every traversal (measurement runs) */
#define KKD_FLAG_PROJECTION_TEST 2u  /* (the default, accepted for callers that set it) leaf test with the reference's own
                                        arithmetic: TriAccel = Wald's projection test, src/librender/skdtree.cpp:77-104, 294-300.
                                        Hit ids and (t, u, v) then equal ShapeKDTree's except for equal-distance ties; like the
                                        reference, not watertight on shared edges */
#define KKD_FLAG_WATERTIGHT_TEST 16u /* leaf test = Woop / Benthin / Wald's watertight test instead: no ray slips between two
                                        triangles that share an edge; 2e-5 ... 5e-4 of the rays then report another primitive or
                                        (t, u, v) than the reference (a few ulp apart: float64 ties).  No counterpart in the
                                        reference; 0.5 % faster, and no second triangle table (48 bytes per triangle less) */

#define KKD_FLAG_ONE_LANE 4u         /* renders normally keep two batches of camera samples in flight on two HIP streams (the
                                        issue-bound kernels of one fill the slots the latency-bound traversal of the other leaves
                                        idle); with this flag batches run one after the other on the context's stream, so that
                                        kkd_stats.stage_seconds are the kernels' exclusive durations (measurement runs) */

#define KKD_FLAG_GPU_BUILD 8u        /* build the acceleration structure on the device (Morton-ordered linear BVH, milliseconds for
                                        10 M triangles) instead of on the host (binned SAH, seconds): for callers whose scenes change
                                        from frame to frame; the host tree is tighter and stays the default.  Replaces the same step,
                                        ShapeKDTree::build (src/librender/skdtree.cpp:68-110); hits do not depend on the choice */

typedef struct kkd_config {
    int32_t  device;           /* HIP device ordinal                                           */
    uint32_t flags;            /* KKD_FLAG_*                                                   */
    uint64_t seed;             /* RNG seed: key = (seed, pixel), counter = (sample, dimension) */
    uint64_t max_samples_in_flight; /* 0 = library default; bounds the wavefront buffers       */
} kkd_config;

/* integrator parameters = the XML parameters of the plugin (kk/direct.cpp:17-21) + the
 * work-unit description that replaces RectangularWorkUnit (librender/rectwu.cpp:33-49). */
typedef struct kkd_render_params {
    uint32_t spp;              /* Sampler::getSampleCount() (integrator.cpp:103-104)           */
    uint32_t bsdf_samples;     /* 'bsdfSamples'  (default 1)                                   */
    uint32_t light_samples;    /* 'lightSamples' (default 1)                                   */
    int32_t  hide_emitters;    /* 'hideEmitters' (default false)                               */
    /* sharding: render only the tiles of rank tile_index out of tile_count (tile mode), and only sample
     * indices [sample_first, sample_first+spp) of each pixel (sample-split mode).               */
    uint32_t block_size;       /* tile edge in pixels, 0 = 32 (scene.cpp:24)                   */
    uint32_t tile_index, tile_count;  /* 0,1 (or 0,0) = all tiles; else the tiles are dealt out round-robin along the Morton
                                         curve of the tile grid and this call renders the share of rank tile_index */
    uint32_t sample_first;     /* first RNG sample index (0)                                   */
    /* which Li: KKD_INTEGRATOR_KK = the path of this library, DirectIntegrator::Li (src/integrators/kk/direct.cpp:50-178);
     * KKD_INTEGRATOR_DIRECT = its sibling on the same kernels, MIDirectIntegrator::Li of the stock 'direct' plugin
     * (src/integrators/direct/direct.cpp:147-311): emitter samples first, power heuristic, 'strictNormals'; there
     * light_samples is 'emitterSamples' and one of the two counts may be 0;
     * KKD_INTEGRATOR_PATH = the next sibling, MIPathTracer::Li of the stock 'path' plugin (src/integrators/path/path.cpp:
     * 119-294): one emitter sample and one BSDF sample per vertex, power heuristic, 'maxDepth' / 'rrDepth' /
     * 'strictNormals' / 'hideEmitters' of MonteCarloIntegrator (src/librender/integrator.cpp:190-233); bsdf_samples and
     * light_samples are ignored. */
    uint32_t integrator;
    int32_t  strict_normals;   /* 'strictNormals' of the direct / path plugins (direct.cpp:103); ignored by KKD_INTEGRATOR_KK */
    int32_t  max_depth;        /* path: 'maxDepth' (-1 = unbounded, 1 = emitted radiance only, 2 = direct illumination ...); 0 is invalid */
    int32_t  rr_depth;         /* path: 'rrDepth', first depth at which Russian roulette is played (> 0; Mitsuba's default 5) */
    /* which Sampler: KKD_SAMPLER_INDEPENDENT = samplers/independent.cpp:82-103 (every dimension a fresh uniform number);
     * KKD_SAMPLER_LDSAMPLER = samplers/ldsampler.cpp:148-229, Mitsuba's default: per pixel, each of the first
     * sampler_dimension 1D / 2D requests and every requested sample array (kk/direct.cpp:44-47: bsdfSamples / lightSamples > 1)
     * is one (0,2)-sequence over the pixel's samples, randomly scrambled and shuffled; later requests are independent;
     * KKD_SAMPLER_STRATIFIED = samplers/stratified.cpp:147-207: each of the first sampler_dimension 2D (1D) requests visits the
     * res x res cells (res^2 intervals) once over the pixel's samples in a shuffled order, jittered inside the cell; requested
     * arrays are Latin hypercubes (libcore/util.cpp:565-576).  res^2 = the sample count rounded up to a perfect square
     * (stratified.cpp:62-79 renders that many; this library renders the spp it is given and leaves the last strata empty, so
     * callers pass a perfect square, as the Python mirror and the Mitsuba adapter do);
     * KKD_SAMPLER_HALTON / _HALTON_PLAIN = samplers/halton.cpp:240-384 with libcore/qmc.cpp:136-150 and samplers/faure.cpp:101-131:
     * deterministic (the seed is not used).  Dimension d of a sample is the radical inverse of its index in the d-th prime base,
     * digits permuted by Faure's permutations (HALTON, the plugin's default scramble = -1) or not (HALTON_PLAIN, scramble = 0;
     * random permutations, scramble > 0, come from Mitsuba's SFMT and are not offered); dimensions 0 and 1 address the pixel
     * through the 128-pixel tiling of setFilmResolution(res, blocked = true); sample arrays take dimensions 5 ... as in
     * generate().  The prime table ends at dimension 1024, where the reference stops with an error; the path integrator's
     * deeper vertices draw independent numbers instead;
     * KKD_SAMPLER_HAMMERSLEY / _HAMMERSLEY_PLAIN = samplers/hammersley.cpp:181-283 on the same functions: dimension 0 is
     * index / (sampleCount x tile size), dimension d > 0 the radical inverse in the (d - 1)-th prime base; tiles of power-of-two
     * size up to 128 pixels.  sampleCount = sample_count_total (or spp).  The plugin refuses sample arrays
     * (hammersley.cpp:296-303), so bsdf_samples and light_samples above 1 are rejected with KKD_ERR_UNSUPPORTED. */
    uint32_t sampler;
    uint32_t sampler_dimension;   /* ldsampler / stratified 'dimension' (0 = their default 4) */
    /* Sampler::getSampleCount() of the whole render when this call renders only the share [sample_first, sample_first + spp) of
     * every pixel's samples (sample-split mode); 0 = spp.  Sets the size of the ldsampler's point sets (rounded up to a power of
     * two, ldsampler.cpp:83-87) and the scale of the ray differentials (integrator.cpp:144-145). */
    uint32_t sample_count_total;
    uint32_t reserved;
} kkd_render_params;
#define KKD_SAMPLER_INDEPENDENT 0u
#define KKD_SAMPLER_LDSAMPLER   1u
#define KKD_SAMPLER_STRATIFIED  2u
#define KKD_SAMPLER_HALTON      3u   /* halton plugin with its default scramble = -1: Faure permutations */
#define KKD_SAMPLER_HALTON_PLAIN 4u  /* halton plugin with scramble = 0: plain radical inverses           */
#define KKD_SAMPLER_HAMMERSLEY  5u   /* hammersley plugin, scramble = -1 (its default)                    */
#define KKD_SAMPLER_HAMMERSLEY_PLAIN 6u /* hammersley plugin, scramble = 0                                */
#define KKD_SAMPLER_SOBOL       7u   /* sobol plugin (samplers/sobol.cpp:82-279), scramble = 0 (its default): Joe and Kuo's direction numbers for
                                        1024 dimensions (samplers/sobolseq.cpp); the first two dimensions address the pixel through
                                        sobol::look_up (sobolseq.h:102-137) on a 2^m x 2^m grid covering the film, sample arrays take dimensions
                                        5, 6 (7, 8); deterministic, so the values ARE the reference's */
#define KKD_INTEGRATOR_KK     0u
#define KKD_INTEGRATOR_DIRECT 1u
#define KKD_INTEGRATOR_PATH   2u

/* ray classes / kernel stages of one render (indices of the arrays in kkd_stats) */
#define KKD_RAYS_PRIMARY 0   /* sensor rays, closest hit   (kernel kPrimary)                      */
#define KKD_RAYS_BSDF    1   /* BSDF-sampled rays, closest hit (kernel kTrace<closest>)           */
#define KKD_RAYS_SHADOW  2   /* emitter-sampled rays, any hit  (kernel kTrace<any>)               */
#define KKD_STAGE_PRIMARY 0
#define KKD_STAGE_SHADE   1
#define KKD_STAGE_BSDF    2
#define KKD_STAGE_SHADOW  3
#define KKD_STAGE_RESOLVE 4

typedef struct kkd_stats {
    uint64_t samples;          /* camera samples processed                                     */
    uint64_t rays_closest;     /* "Normal rays traced"  (skdtree.cpp:46,123)                    */
    uint64_t rays_shadow;      /* "Shadow rays traced"  (skdtree.cpp:47,151,210)                */
    double   seconds;          /* device time of the render loop (HIP events on the context's stream) */
    double   seconds_trace;    /* part of it spent in the traversal kernels                    */
    double   algorithmic_bytes;/* SURVEY 8d, traversal kernels: sum over rays of B_ray (0 unless traversal counting is on) */
    uint64_t nodes_visited;    /* totals of the per-class counters below                        */
    uint64_t tris_tested;
    uint64_t class_rays[3];    /* rays actually traced per class (KKD_RAYS_*)                   */
    uint64_t class_nodes[3];   /* BVH nodes visited per class   (only with KKD_FLAG_COUNT_TRAVERSAL) */
    uint64_t class_tris[3];    /* triangles tested per class    (only with KKD_FLAG_COUNT_TRAVERSAL) */
    double   stage_seconds[5]; /* device time per kernel stage (KKD_STAGE_*), summed over launches.  A render keeps two batches
                                  in flight on two streams, so a stage's time includes the other batch's kernels sharing the GPU
                                  and the five add up to more than `seconds`; with KKD_FLAG_ONE_LANE they are exclusive.  Zero
                                  for KKD_INTEGRATOR_PATH (a batch is a loop of stages there; only `seconds` is timed) */
    uint64_t stage_launches[5];/* kernel launches per stage */
} kkd_stats;

typedef struct kkd_hit {       /* IntersectionCache of skdtree.h: {shapeIndex, primIndex, u, v} + t */
    float    t;                /* +inf on miss                                                 */
    float    u, v;             /* barycentrics of vertices 1 and 2 (triaccel.h)                */
    uint32_t prim;             /* global primitive id (shape first_prim + primIndex), 0xFFFFFFFF on miss */
} kkd_hit;

typedef struct kkd_ctx kkd_ctx;

/* ---- entry points ---------------------------------------------------------------------- */

/* Number of usable HIP devices.  Replaces Scheduler::getCoreCount() in the log line of
 * SamplingIntegrator::render (src/librender/integrator.cpp:102-109). */
int kkd_device_count(void);

/* Create a context on one device.  Replaces the per-process state BlockedRenderProcess /
 * BlockRenderer hold (src/librender/renderproc.cpp:26-66).  NULL on failure. */
kkd_ctx *kkd_create(const kkd_config *cfg);

/* Copy the flattened scene to the device, build the BVH, the emitter CDFs
 * (Scene::initialize, src/librender/scene.cpp:376-380; TriMesh::prepareSamplingTable,
 * src/librender/trimesh.cpp:392-407; EnvironmentMap::configure, src/emitters/envmap.cpp:260-329),
 * the rough-transmittance tables (src/bsdfs/rtrans.h:288-384) and the filter table
 * (src/libcore/rfilter.cpp:37-55).  Replaces Scene::preprocess -> ShapeKDTree::build
 * (src/librender/skdtree.cpp:68-110).  The caller keeps ownership of all input arrays. */
int kkd_scene_upload(kkd_ctx *ctx, const kkd_scene_desc *scene);

/* Render: the body of SamplingIntegrator::render + renderBlock + DirectIntegrator::Li
 * (src/librender/integrator.cpp:95-188, src/integrators/kk/direct.cpp:50-178).
 * out_film: crop_width*crop_height*5 floats, host memory, overwritten with the ImageBlock
 * sums {R*w,G*w,B*w,alpha*w,w} of the samples this call rendered -- hand it to Film::put /
 * Film::setBitmap (src/films/hdrfilm.cpp:391-397).  stats may be NULL. */
int kkd_render(kkd_ctx *ctx, const kkd_render_params *params, float *out_film, kkd_stats *stats);

/* Same, but the film stays in device memory: dev_film is a caller-owned device buffer of
 * crop_width*crop_height*5 floats (e.g. a torch tensor) that is overwritten with this call's sums, so
 * that multi-GPU callers can reduce it with RCCL without a host round trip (sample-split mode,
 * SURVEY 8e; the reduce plays the role of Film::put(block), src/films/hdrfilm.cpp:391-393).  The call
 * returns after the work has completed on the context's own stream. */
int kkd_render_device(kkd_ctx *ctx, const kkd_render_params *params, void *dev_film, kkd_stats *stats);

/* Fixed-ray-set entry point: ShapeKDTree::rayIntersect(ray, its) when any_hit==0
 * (src/librender/skdtree.cpp:112-142) and ShapeKDTree::rayIntersect(ray) when any_hit!=0
 * (skdtree.cpp:207-226), including their adaptive ray epsilon.  rays: n x 8 floats
 * {o.x,o.y,o.z,mint,d.x,d.y,d.z,maxt} in host memory; out: n kkd_hit in host memory
 * (any_hit: only prim!=0xFFFFFFFF / t<inf is meaningful). */
int kkd_trace(kkd_ctx *ctx, const float *rays, uint64_t n, int any_hit, kkd_hit *out);

/* Same queries (src/librender/skdtree.cpp:112-142, 207-226) with rays and hits resident in device memory, as
 * the measurement protocol of src/utils/kdbench.cpp:210-242 times them. */
int kkd_trace_device(kkd_ctx *ctx, const void *dev_rays, uint64_t n, int any_hit, void *dev_out, kkd_stats *stats);

/* kkd_trace_device for callers that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb an allocator while other work is being captured or replayed.  No counterpart in the reference
 * (its CPU path has no device memory to allocate).  Same queries, same results and statistics as kkd_trace_device; on the
 * success path the call makes no hipMalloc / hipFree / event or stream creation and no host allocation (an error builds its
 * message in a std::string): the sharded ray counters of the launch, which kkd_trace_device keeps in a buffer of the context,
 * live in a file-scope device array of the library (128 KiB on every GPU the library runs on, held until the process ends;
 * looked up once per device by kkd_create).  The call is NOT itself capturable into a HIP graph: it runs on the context's
 * private stream, records timing events, copies the counters back and synchronises before it returns.  dev_rays / dev_out are caller-owned device memory as for
 * kkd_trace_device.  At most KKD_NOALLOC_MAX_RAYS rays per call (one launch), more is refused with KKD_ERR_UNSUPPORTED.
 * All calls on one GPU share that GPU's array: it is cleared at the start of every call, and calls that overlap on one GPU
 * (two contexts, two threads) race on the counters -- their hits stay correct, their statistics do not. */
#define KKD_NOALLOC_MAX_RAYS (1ull << 30)
int kkd_trace_device_noalloc(kkd_ctx *ctx, const void *dev_rays, uint64_t n, int any_hit, void *dev_out, kkd_stats *stats);

/* Counting variant of kkd_trace (host buffers): also reports visited nodes / tested triangles in
 * stats->nodes_visited / tris_tested, the measured N_nodes and N_tris of SURVEY 8d.  Plays the role of the
 * reference's ray statistics counters (src/librender/skdtree.cpp:46-47,123,151,210). */
int kkd_trace_counted(kkd_ctx *ctx, const float *rays, uint64_t n, int any_hit, kkd_hit *out, kkd_stats *stats);

/* Sizes of the acceleration-structure records, for the algorithmic-bytes formula of SURVEY 8d (the reference's
 * records: 8-byte KDNode, 4-byte index, 48-byte TriAccel, src/librender/skdtree.cpp:77-79). */
int kkd_accel_info(kkd_ctx *ctx, uint32_t *node_bytes, uint32_t *tri_bytes, uint64_t *n_nodes, uint64_t *n_leaf_tris);

/* Host-only probes of the tables kkd_scene_upload builds (no GPU needed; used by the CPU tests).
 * Rough-dielectric transmittance for one (distribution, alpha, eta): 100 knots at cosTheta^(1/4) = i/99 and the
 * diffuse transmittance, i.e. what RoughTransmittance yields after setEta()/setAlpha() (src/bsdfs/rtrans.h:183-280)
 * from the data file written by src/utils/rdielprec.cpp:66-190. */
int kkd_probe_rough_transmittance(int distribution, float alpha, float eta, float *trans100, float *diffuse_transmittance);

/* Discretised reconstruction filter: 32 table values, scale factor and border size as computed by
 * ReconstructionFilter::configure (src/libcore/rfilter.cpp:37-55) for the filters of src/rfilters/. */
int kkd_probe_rfilter(int rfilter_type, float radius, const float *param2, float *values32, float *scale_factor, int *border);

/* ldsampler point sets (src/samplers/ldsampler.cpp:148-183): the values the device hands to samples 0 ... count-1 of `pixel`
 * for one stream -- kind 0: 2D request `index` (out: count x 2), kind 1: 1D request `index` (out: count), kind 2: requested
 * array `index` with `array_n` entries per sample (out: count x array_n x 2) -- when the point sets hold set_size samples. */
int kkd_probe_ldsampler(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t array_n, uint32_t set_size, uint32_t count, float *out);
/* The same for KKD_SAMPLER_STRATIFIED (StratifiedSampler::next2D / next1D, stratified.cpp:181-207, and the Latin hypercube
 * arrays of generate(), :158-162) with `resolution`^2 samples per pixel; flat_dim (even) = the dimension of the independent
 * stream whose values jitter the request inside its cell, as in a render. */
int kkd_probe_stratified(uint64_t seed, uint32_t pixel, int kind, uint32_t index, uint32_t array_n, uint32_t resolution, uint32_t flat_dim, uint32_t count, float *out);
/* HaltonSampler on a film of res_x x res_y pixels (setFilmResolution(res, true), generate(x, y), halton.cpp:240-384), samples
 * 0 ... count-1 of pixel (x, y) -- kind 0: next2D at running dimension `dim` (out: count x 2; dim 0 = the position inside the
 * pixel), kind 1: next1D at `dim` (out: count), kind 2: requested 2D array `dim` with array_n entries per sample when
 * n_arrays are requested (out: count x array_n x 2).  faure != 0: scramble = -1 (Faure permutations), else scramble = 0.
 * n_arrays = 0xFFFFFFFF selects the hammersley plugin (kinds 0 and 1 only) with sampleCount = array_n. */
int kkd_probe_halton(uint32_t res_x, uint32_t res_y, int32_t x, int32_t y, int faure, int kind, uint32_t dim, uint32_t array_n, uint32_t n_arrays, uint32_t count, float *out);
/* The Faure permutation of `base` (2 ... 8161) as PermutationStorage::computeFaurePermutations builds it (samplers/faure.cpp:101-131);
 * out: base entries. */
int kkd_probe_faure_permutation(uint32_t base, uint16_t *out);
/* The sobol sampler on the host, for tests (samplers/sobol.cpp:158-262 on a res_x x res_y film; arguments as kkd_probe_halton,
 * faure unused), and its two building blocks: sobol::sampleSingle(index, dim) as its 32 result bits, sobol::look_up(m, frame, px,
 * py, 0) (samplers/sobolseq.h:43-60, 102-137); crc32 (optional) receives the CRC-32 of the regenerated 1024 x 52 matrix words. */
int kkd_probe_sobol(uint32_t res_x, uint32_t res_y, int32_t x, int32_t y, int kind, uint32_t dim, uint32_t array_n, uint32_t n_arrays, uint32_t count, float *out);
int kkd_probe_sobol_raw(uint64_t index, uint32_t dim, uint32_t m, uint32_t frame, uint32_t px, uint32_t py, uint32_t *sample_bits, uint64_t *look_up_index, uint32_t *crc32);

/* ---- film develop and output: the step right after the path ---- */

/* Bitmap::EPixelFormat values the develop step can produce (SPECTRUM_SAMPLES = 3: spectrum = linear RGB) */
#define KKD_PIXEL_LUMINANCE       0   /* 1 channel  */
#define KKD_PIXEL_LUMINANCE_ALPHA 1   /* 2 channels */
#define KKD_PIXEL_RGB             2   /* 3 channels */
#define KKD_PIXEL_RGBA            3   /* 4 channels */
#define KKD_PIXEL_XYZ             4   /* 3 channels */
#define KKD_PIXEL_XYZA            5   /* 4 channels */

/* HDRFilm::develop / MFilm::develop (src/films/hdrfilm.cpp:481-497, src/films/mfilm.cpp:264-265): Bitmap::convert of
 * the ESpectrumAlphaWeight film {R*w,G*w,B*w,alpha*w,w} to pixel_format, float32, gamma 1, i.e. the
 * ESpectrumAlphaWeight source branch of FormatConverterImpl::convert (src/libcore/fmtconv.cpp:955-1034):
 * invWeight = w != 0 ? 1/w : 0, colour and alpha times invWeight, luminance / XYZ by the Rec. 709 weights of
 * src/libcore/spectrum.cpp:229-234.  dev_film: width*height*5 floats, dev_out: width*height*channels floats, both in
 * device memory.  The Mitsuba banner (hdrfilm.cpp:493-504) is not drawn. */
int kkd_develop_device(kkd_ctx *ctx, const void *dev_film, uint32_t width, uint32_t height, int32_t pixel_format, void *dev_out);

/* Same conversion (src/libcore/fmtconv.cpp:955-1034) with film and result in host memory. */
int kkd_develop(kkd_ctx *ctx, const float *film, uint32_t width, uint32_t height, int32_t pixel_format, float *out);

/* Bitmap::writePFM (src/libcore/bitmap.cpp:3804-3843): "PF" (RGB / RGBA, alpha stripped) or "Pf" (luminance) header,
 * size, "-1" for little endian, then the scanlines bottom-up as float32.  Host only, no context needed. */
int kkd_write_pfm(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format);

/* Bitmap::writeRGBE (src/libcore/bitmap.cpp:3679-3736; HDRFilm fileFormat = "rgbe", src/films/hdrfilm.cpp:224-225, 314-324:
 * pixel format forced to RGB float32): "#?RGBE", "FORMAT=32-bit_rle_rgbe", "-Y h +X w", then every scanline as four
 * run-length-encoded byte planes (R, G, B, shared exponent; RGBE_FromFloat / RGBE_WriteBytes_RLE, bitmap.cpp:3492-3576), flat
 * pixels when the width is below 8 or above 32767.  pixel_format: KKD_PIXEL_RGB or KKD_PIXEL_RGBA (alpha is skipped).  No
 * metadata comments (what the reference writes without annotations and with attachLog = false).  Host only. */
int kkd_write_rgbe(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format);

/* Bitmap::writeOpenEXR (src/libcore/bitmap.cpp:3171-3318; HDRFilm's default fileFormat, src/films/hdrfilm.cpp:213-223, with
 * componentFormat float16 by default): a single-part scanline OpenEXR 2 file with the header Imf::Header(width, height) carries
 * (channels, compression, dataWindow, displayWindow, lineOrder = increasing y, pixelAspectRatio 1, screenWindowCenter 0 0,
 * screenWindowWidth 1), the attribute "generatedBy", chromaticities as bitmap.cpp:3223-3231 adds them (Rec. 709 for RGB, the XYZ
 * primaries for XYZ), channels Y | R G B (+ A) stored in alphabetical order, ZIP compression in blocks of 16 scanlines
 * (Imf::Header's default) or none.  The OpenEXR library (third party, not in the checkout, not installed here) is replaced by
 * a writer that follows the published file layout; no OpenEXR reader is available to check the files against, parity unpinned.
 * component_format: KKD_COMPONENT_FLOAT16 (round to nearest even, as half.h) or KKD_COMPONENT_FLOAT32.  Host only. */
#define KKD_COMPONENT_FLOAT16 0
#define KKD_COMPONENT_FLOAT32 1
#define KKD_EXR_NO_COMPRESSION 0
#define KKD_EXR_ZIP_COMPRESSION 3
int kkd_write_exr(const char *path, const float *pixels, uint32_t width, uint32_t height, int32_t pixel_format,
                  int32_t component_format, int32_t compression, const char *generated_by);

/* MFilm::develop with fileFormat = numpy (src/films/mfilm.cpp:336-352): a version-1.0 .npy file holding a C-ordered
 * little-endian float32 array of shape (height, width, channels), or (height, width) when channels == 1, as
 * cnpy::npy_save writes it.  Host only, no context needed. */
int kkd_write_npy(const char *path, const float *pixels, uint32_t width, uint32_t height, uint32_t channels);

/* MFilm::develop with fileFormat = matlab (MFilm's default) or mathematica (src/films/mfilm.cpp:269-334): a .m text file that
 * assigns the image to `variable` (NULL = "data", mfilm.cpp:135), one matrix per channel, numbers printed with `digits`
 * significant digits (0 = 4, mfilm.cpp:134) the way `ostream << float` with setprecision(digits) prints them; Mathematica
 * numbers use "*^" for the exponent.  Same bytes as the reference's loop.  Host only, no context needed. */
#define KKD_MFILM_MATLAB      0
#define KKD_MFILM_MATHEMATICA 1
int kkd_write_mfilm_text(const char *path, const float *pixels, uint32_t width, uint32_t height, uint32_t channels,
                         int32_t file_format, int32_t digits, const char *variable);

/* ---- parameter extraction for the Mitsuba-side adapter ---- */

/* Decodes the byte stream InstanceManager::serialize(stream, bsdf) writes (src/libcore/serialization.cpp:73-87) for
 * the BSDFs of this path -- SmoothDiffuse (src/bsdfs/diffuse.cpp:161-165), RoughConductor (roughconductor.cpp:216-226),
 * RoughPlastic (roughplastic.cpp:243-253), SmoothDielectric (dielectric.cpp:176-182), TwoSidedBRDF (twosided.cpp:77-82)
 * with constant textures (src/libhw/basicshader.cpp:29-49) -- into kkd_material blocks, nested BSDFs as further
 * entries.  Several BSDFs serialised one after the other with the same InstanceManager share sub-objects by id and
 * are parsed in one call; the k-th top-level BSDF is the first entry not referenced as `nested` before it.  Host
 * only; errors through kkd_last_error(NULL). */
int kkd_parse_bsdf_stream(const void *bytes, uint64_t n_bytes, kkd_material *out, uint32_t capacity, uint32_t *count);

/* ---- on-disk mesh formats: the step right before the path ---- */

#define KKD_MESH_FACE_NORMALS 1u   /* TriMesh::m_faceNormals: shade with face normals (EFaceNormals, src/librender/trimesh.cpp:93) */

/* One triangle mesh as TriMesh holds it (src/librender/trimesh.cpp:39-54); the arrays are malloc'ed by the loader
 * and released by kkd_mesh_free.  They are laid out as kkd_scene_desc wants them. */
typedef struct kkd_mesh {
    uint64_t  n_verts, n_tris;
    float    *positions;      /* n_verts * 3                                     */
    float    *normals;        /* n_verts * 3 or NULL                             */
    float    *texcoords;      /* n_verts * 2 or NULL                             */
    float    *colors;         /* n_verts * 3 or NULL (vertex colours, linear RGB) */
    uint32_t *indices;        /* n_tris * 3                                      */
    uint32_t  flags;          /* KKD_MESH_*                                      */
    uint32_t  reserved;
    float     aabb_min[3], aabb_max[3];
    char      name[256];
} kkd_mesh;

/* Number of meshes in a .serialized file: TriMesh::readOffsetDictionary (src/librender/trimesh.cpp:296-337); a file
 * without a valid dictionary holds one mesh (src/shapes/serialized.cpp:232-236). */
int kkd_mesh_count_serialized(const char *path, uint32_t *count);

/* The serialized plugin: TriMesh::loadCompressed (src/librender/trimesh.cpp:174-249; versions 3 and 4, single or
 * double precision, zlib stream) for sub-mesh shape_index, then the toWorld handling of SerializedMesh
 * (src/shapes/serialized.cpp:183-201: points, normalised normals, winding flipped when det < 0).  to_world: 16 floats
 * row-major or NULL.  Errors: kkd_last_error(NULL). */
int kkd_mesh_load_serialized(const char *path, int32_t shape_index, const float *to_world, kkd_mesh *out);

/* The .serialized writer, TriMesh::serialize(Stream*) (src/librender/trimesh.cpp:1060-1108), version 4, single
 * precision, followed by the offset dictionary that readOffsetDictionary (trimesh.cpp:296-337) reads back. */
int kkd_mesh_save_serialized(const char *path, const kkd_mesh *meshes, uint32_t n_meshes);

/* The ply plugin, PLYLoader (src/shapes/ply.cpp:96-330, property mapping :326-470): ascii and both binary formats;
 * float32 x/y/z, nx/ny/nz, u|texture_u|s, v|texture_v|t, [diffuse_]red/green/blue as float32 or uint8 (sRGB -> linear
 * when srgb != 0); faces "vertex_indices" | "vertex_index" with a uint8 / uint32 count and int32 / uint32 indices,
 * triangles and quads (a quad becomes (0,1,2) + (3,0,2)); properties of other types are skipped as libply does. */
int kkd_mesh_load_ply(const char *path, const float *to_world, int32_t srgb, kkd_mesh *out);

/* The obj plugin's reader, WavefrontOBJ::WavefrontOBJ + parse + createMesh (src/shapes/obj.cpp:165-349, 373-392, 586-717): "v" /
 * "vn" / "vt" lists, faces "p", "p/t", "p//n", "p/t/n" with 1-based or negative (relative) indices, n-gons as fans, one mesh per
 * geometry that a "g" or "usemtl" line (or the end of the file) closes, corners merged when position, normal and uv are equal
 * after toWorld; mesh names as the plugin forms them (both "g" conventions, "_<index>" for repeats).  A file holds
 * kkd_mesh_count_obj meshes, numbered in the order the plugin creates them with shapeIndex = -1; mesh_index selects one, and
 * `material` (may be NULL) receives the "usemtl" name in force, which WavefrontOBJ::addChild matches BSDF names against
 * (obj.cpp:738-757).  The plugin's 'shapeIndex' = s >= 0 is not simply "mesh s": the meshes closed by a "g" / "usemtl" line are
 * created when their number equals s, but the one the END of the file closes is tested against the number of meshes closed
 * before it minus one (obj.cpp:332), so it is created together with mesh count - 2 and never under its own number.
 * *eof_mesh (may be NULL) says whether the last mesh was closed by the end of the file, so that an adapter can reproduce
 * that; the Python mirror's obj_shape_index_meshes() spells the rule out.  flags: KKD_OBJ_*.  'faceNormals' is KKD_MESH_FACE_NORMALS
 * on the loaded mesh, 'maxSmoothAngle' kkd_mesh_rebuild_topology (obj.cpp:336-343).  Not read: "mtllib" -- the material
 * library would create phong / bitmap-texture BSDFs, which are outside this library's BSDF set; pass the scene's own BSDFs.
 * 'flipNormals' is not offered by any of the loaders. */
#define KKD_OBJ_COLLAPSE         1u   /* 'collapse': ignore "g" / "usemtl" boundaries, one mesh */
#define KKD_OBJ_KEEP_TEXCOORD_V  2u   /* 'flipTexCoords' = false (the plugin's default stores v as 1 - v, obj.cpp:211, 306-307) */
int kkd_mesh_count_obj(const char *path, uint32_t flags, uint32_t *count, uint32_t *eof_mesh);
int kkd_mesh_load_obj(const char *path, uint32_t mesh_index, const float *to_world, uint32_t flags, kkd_mesh *out, char *material, uint32_t material_capacity);

/* The 'maxSmoothAngle' parameter of the mesh plugins (src/shapes/ply.cpp:124-128, serialized.cpp:204-208, obj.cpp:336-342):
 * TriMesh::rebuildTopology (src/librender/trimesh.cpp:472-610) on a loaded mesh, in place -- vertices shared by faces whose
 * normals differ by more than max_angle_deg are split, the others merged (keyed by position, uv and colour); stored normals
 * are dropped, the caller regenerates them as TriMesh::configure does (trimesh.cpp:366-390).  Host only. */
int kkd_mesh_rebuild_topology(kkd_mesh *mesh, float max_angle_deg);

/* Releases the arrays of a loaded mesh (the reference frees them in TriMesh::~TriMesh, src/librender/trimesh.cpp:339-352). */
void kkd_mesh_free(kkd_mesh *mesh);

/* Asynchronous, thread-safe: replaces SamplingIntegrator::cancel (src/librender/integrator.cpp:90-93). */
void kkd_request_cancel(kkd_ctx *ctx);

/* Progress of a render: replaces ProgressReporter::update per finished block (BlockedRenderProcess::processResult,
 * src/librender/renderproc.cpp:142-149) and gives the adapter its hook for RenderQueue::signalWorkEnd / signalRefresh
 * (src/librender/renderqueue.cpp:119-141).  fn is called on the thread inside kkd_render / kkd_render_device, between
 * batches, with the number of work items (camera samples, including the padding of partial tiles) whose results are in
 * the film (completed on the device) and their total;
 * the last call of a finished render reports done == total.  fn == NULL removes the callback.  It must not call into ctx
 * (kkd_request_cancel and kkd_request_flush excepted). */
typedef void (*kkd_progress_fn)(void *user, uint64_t samples_done, uint64_t samples_total);
int kkd_set_progress_callback(kkd_ctx *ctx, kkd_progress_fn fn, void *user);

/* The film while the render runs: what `mitsuba -r <seconds>` does with its FlushThread (src/mitsuba/mitsuba.cpp:105-125 ->
 * RenderQueue::flush, src/librender/renderqueue.cpp:73-78 -> Scene::flush = Film::develop, src/librender/scene.cpp:458-460).
 * kkd_request_flush is asynchronous and thread-safe like kkd_request_cancel (and, like it, may be called from the progress
 * callback); the render thread answers it between two batches: it lets the batches already queued finish, copies the film
 * (width x height x 5 floats, the layout kkd_render returns) into a pinned host buffer of the context and calls fn on the
 * render thread with the work items whose results the copy holds.  The buffer is valid during the call only; develop it
 * with kkd_develop or write it with the kkd_write_* functions.  A request that arrives after the last batch was queued is
 * dropped: the finished film follows anyway.  Costs one drain of the pipeline and one copy per request.  fn == NULL removes
 * the callback.  (kkd_multi_set_flush_callback / kkd_multi_request_flush do the same for a multi-device job.) */
typedef void (*kkd_flush_fn)(void *user, const float *film, uint32_t width, uint32_t height, uint64_t samples_done, uint64_t samples_total);
int kkd_set_flush_callback(kkd_ctx *ctx, kkd_flush_fn fn, void *user);
void kkd_request_flush(kkd_ctx *ctx);

/* ---- several GPUs of one node behind one handle ---------------------------------------------------------------
 * What the reference does with its scheduler -- every worker renders blocks (src/librender/renderproc.cpp:68-86),
 * BlockedRenderProcess::processResult adds each into the film (src/librender/renderproc.cpp:142-149 -> Film::put,
 * src/films/hdrfilm.cpp:391-393); work units come from BlockedImageProcess (src/librender/imageproc.cpp:28-78) -- for the
 * GPUs of one node: one context per listed device, the scene flattened once and replicated, one host thread per device
 * while rendering, no exchange before the frame is done.
 *   KKD_SPLIT_TILE    the image tiles are dealt out to the devices round-robin along the Morton curve of the tile grid;
 *                     each device's tiles travel to the first device (the root) with their filter borders in one peer
 *                     copy and are added into the root's film there;
 *   KKD_SPLIT_SAMPLE  every device renders the whole frame with spp / n_devices samples per pixel (disjoint sample
 *                     indices); the films are summed to the root by one RCCL reduce (librccl is loaded at run time; if it
 *                     is absent, or a device is listed twice, the films are copied to the root and added there).
 * The sample stream is keyed by pixel and sample index, so the film does not depend on the number of devices (up to the
 * order of float additions).  A caller's own tile_index / tile_count / sample_first (a job spread over several nodes or
 * processes) are refined, not replaced. */
#define KKD_SPLIT_TILE   0u
#define KKD_SPLIT_SAMPLE 1u
typedef struct kkd_multi_config {
    uint32_t n_devices;        /* 1..64                                                          */
    uint32_t split;            /* KKD_SPLIT_*                                                    */
    const int32_t *devices;    /* n_devices HIP device ordinals; devices[0] is the root           */
    uint32_t flags;            /* KKD_FLAG_*, as kkd_config                                      */
    uint32_t reserved;
    uint64_t seed;
    uint64_t max_samples_in_flight;   /* per device, 0 = library default                         */
} kkd_multi_config;
typedef struct kkd_multi kkd_multi;

/* One context per listed device: replaces the per-worker state of BlockedRenderProcess / BlockRenderer
 * (src/librender/renderproc.cpp:26-66) and the worker set-up of src/mitsuba/mitsuba.cpp:260-264.  NULL on failure
 * (kkd_last_error(NULL) / kkd_multi_last_error(NULL) has the text). */
kkd_multi *kkd_multi_create(const kkd_multi_config *cfg);
/* kkd_scene_upload for every device: one host-side build (Scene::preprocess -> ShapeKDTree::build,
 * src/librender/skdtree.cpp:68-110), uploads side by side. */
int kkd_multi_scene_upload(kkd_multi *m, const kkd_scene_desc *scene);
/* kkd_render / kkd_render_device over all devices: SamplingIntegrator::render (src/librender/integrator.cpp:95-129) with
 * the film merged as BlockedRenderProcess::processResult does (src/librender/renderproc.cpp:142-149).  out_film: host
 * memory; dev_film: device memory on devices[0].  stats: counters summed over the devices, seconds = host wall clock of the
 * job including the merge; stage_seconds summed over the devices. */
int kkd_multi_render(kkd_multi *m, const kkd_render_params *params, float *out_film, kkd_stats *stats);
/* same (src/librender/integrator.cpp:95-129, src/librender/renderproc.cpp:142-149), merged film left in device memory on devices[0] */
int kkd_multi_render_device(kkd_multi *m, const kkd_render_params *params, void *dev_film, kkd_stats *stats);
/* SamplingIntegrator::cancel (src/librender/integrator.cpp:90-93) for every device; thread-safe. */
void kkd_multi_request_cancel(kkd_multi *m);
/* kkd_set_progress_callback for the whole job (ProgressReporter::update per finished block,
 * src/librender/renderproc.cpp:142-149): fn receives the work items finished on all devices and their total; it is called from the devices' host
 * threads, one call at a time. */
int kkd_multi_set_progress_callback(kkd_multi *m, kkd_progress_fn fn, void *user);
/* how the last render merged its films: "single", "tile-gather", "rccl-reduce" or "copy-add" (measurement / tests; the
 * reference logs its scheduler set-up the same way, src/librender/integrator.cpp:102-109) */
const char *kkd_multi_merge_path(kkd_multi *m);
/* what kkd_multi_create found on this machine, as text: the device list, for every (root, device) pair whether
 * hipDeviceCanAccessPeer allows direct copies in either direction and whether enabling them succeeded, and (sample-split
 * mode) which librccl was loaded and its version -- so that a failing first run on real devices names its cause.  Also
 * printed to stderr by kkd_multi_create when $KKD_MULTI_VERBOSE is set.  The reference logs its worker set-up likewise
 * (src/mitsuba/mitsuba.cpp:260-309). */
const char *kkd_multi_diagnostics(kkd_multi *m);
/* kkd_set_flush_callback / kkd_request_flush for the whole job (`mitsuba -r <seconds>`, src/mitsuba/mitsuba.cpp:105-125 ->
 * RenderQueue::flush, src/librender/renderqueue.cpp:73-78): every device answers the request between two of its batches (or
 * with its finished film, if its share is done) with a copy of ITS film; the films of a tile-sharded or sample-split job add
 * up to the job's film, so the device that completes the round calls fn, on its render thread, with the sum (host memory,
 * valid during the call) and the work items it holds.  A request that arrives while a round is being collected is answered
 * with the finished films at the end of the render.  fn == NULL removes the callback. */
int kkd_multi_set_flush_callback(kkd_multi *m, kkd_flush_fn fn, void *user);
void kkd_multi_request_flush(kkd_multi *m);
uint32_t kkd_multi_device_count(kkd_multi *m);   /* contexts behind the handle (src/librender/integrator.cpp:102: getCoreCount) */
const char *kkd_multi_last_error(kkd_multi *m);  /* as kkd_last_error (src/librender/renderjob.cpp:111-115) */
/* releases every context (the reference tears its workers down in Scheduler::staticShutdown, src/mitsuba/mitsuba.cpp:436) */
void kkd_multi_destroy(kkd_multi *m);

/* Mitsuba convention is Log(EError, ...) -> exception (src/librender/renderjob.cpp:111-115); here the
 * text is kept per context.  Never NULL.  ctx == NULL returns the last error of kkd_create. */
const char *kkd_last_error(kkd_ctx *ctx);

void kkd_destroy(kkd_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* KKDIRECT_H */
