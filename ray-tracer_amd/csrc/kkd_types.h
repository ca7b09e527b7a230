/*
 * kkd_types.h -- layout of the GPU-resident scene of libkkdirect (HBM-resident, read-only during a
 * render) and of the wavefront work buffers.  Shared by the host builders and the gfx950 kernels.
 *
 * Reference types these replace (paths relative to /root/reference/src):
 *   KDNode + m_indices + TriAccel table (librender/skdtree.cpp:68-110)  -> Node4 + LeafTri
 *   BSDF plugin state (bsdfs/*.cpp serialize())                        -> DMaterial
 *   Emitter plugin state + Scene::m_emitterPDF (librender/scene.cpp:376-380) -> DEmitter + emitterCdf
 *   EnvironmentMap tables (emitters/envmap.cpp:260-329)                -> DEnvMap
 *   PerspectiveCameraImpl (sensors/perspective.cpp:126-180)            -> DSensor
 *   ReconstructionFilter table (libcore/rfilter.cpp:37-55)             -> DFilter
 */
#pragma once
#include "kkd_math.h"
#include "../../include/kkdirect.h"

namespace kkd {

/* ---- acceleration structure -------------------------------------------------------------------
 * 4-wide BVH, one 64-byte node = half an L2 line = four 16-byte loads per lane.  What limits incoherent traversal on
 * CDNA4 is the number of divergent 16-byte load instructions a wave issues (every lane touches its own line), not
 * ALU work, so the four child boxes are stored as 8-bit grid coordinates relative to the node's own box:
 *     plane = origin + q * 2^(exp - 127)      (exact power-of-two scale per axis)
 * The builder rounds min planes down and max planes up and verifies the float decode (same fmaf) encloses the true
 * box.  child[i]: 0 = empty slot (the root is never a child); bit31 set = leaf, bits 28..30 = triangle count - 1,
 * bits 0..27 = first LeafTri slot; otherwise index of the child node. */
#ifdef KKD_NODE_FP16
/* Experimental layout (round 3, -DKKD_NODE_FP16): child planes as half floats in units of the node's grid scale, 80 bytes =
 * five 16-byte loads.  v_fma_mix_f32 reads a half operand directly, so a plane costs ONE instruction (t = h * ta + tb) instead
 * of a byte->float conversion and an fma, and 11 bits of mantissa give tighter boxes than the 8-bit grid. */
struct alignas(16) Node4 {
    float ox, oy, oz;            /* grid origin = min corner of the node box                    */
    uint8_t ex, ey, ez, epad;    /* biased exponents of the per-axis grid scale                 */
    uint32_t child[4];
    uint16_t hminx[4], hminy[4], hminz[4], hmaxx[4], hmaxy[4], hmaxz[4];
};
#else
struct alignas(64) Node4 {
    float ox, oy, oz;            /* grid origin = min corner of the node box                    */
    uint8_t ex, ey, ez, epad;    /* biased exponents of the per-axis grid scale                 */
    uint32_t child[4];
    uint8_t qminx[4], qminy[4], qminz[4], qmaxx[4];
    uint8_t qmaxy[4], qmaxz[4]; uint32_t pad[2];
};
#endif
static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kLeafStartMask = 0x0FFFFFFFu;
static constexpr uint32_t kTraversalDone = 0xFFFFFFFFu;   /* never a leaf reference: LeafTri slots stay below kLeafStartMask */
static constexpr int kMaxLeafTris = 4;
/* the per-lane walk addresses node and triangle records by 32-bit byte offsets (kkd_trace.h) */
static constexpr uint64_t kMaxAccelBytes = 0xFFFFFFFFull;

/* The same tree with full-float child boxes, for the packet traversal of primary rays (one wave = the samples of one pixel
 * or of a few neighbouring pixels): the node is fetched ONCE per wave through the scalar cache, so its size costs no
 * per-lane loads.  Planes are stored relative to the node's min corner ("origin"), rounded outwards, so that a slab
 * distance is one fma per plane: t = rel * (1/d) + (origin - o) * (1/d). */
struct alignas(128) WideNode {
    float ox, oy, oz, pad0;
    float rminx[4], rminy[4], rminz[4], rmaxx[4], rmaxy[4], rmaxz[4];
    uint32_t child[4];           /* as Node4::child */
};

/* Triangles in leaf order, 48 bytes: three 16-byte loads.  w components carry ids (bit patterns). */
struct alignas(16) LeafTri {
    float v0x, v0y, v0z; uint32_t prim;   /* global primitive id = first_prim + primIndex          */
    float v1x, v1y, v1z; uint32_t shape;  /* shape index (ShapeKDTree::addShape order)              */
    float v2x, v2y, v2z; uint32_t emitter; /* area emitter of the shape + 1, 0 = none (filled in at upload)  */
};
/* set in the hit id of a BSDF-ray hit (internal hit buffers only) when the surface hit does not emit: the resolve
 * kernel then needs neither the triangle nor the ray */
static constexpr uint32_t kHitNotEmitter = 0x40000000u;

/* Optional second triangle table for KKD_FLAG_PROJECTION_TEST: the reference's TriAccel record (Wald's projection
 * test, skdtree.cpp:77-104 + triaccel.h), same slot order as LeafTri.  k = 3 marks a degenerate triangle. */
struct alignas(16) ProjTri {
    float n_u, n_v, n_d; uint32_t k;
    float a_u, a_v, b_nu, b_nv;
    float c_nu, c_nv; uint32_t prim /* = LeafTri::prim: equal-distance ties go to the lower primitive id */, pad;
};

/* ---- BSDF::EBSDFType bits used on this path (names follow the reference's bsdf.h) ---- */
enum : uint32_t {
    EDiffuseReflection = 0x0004, EGlossyReflection = 0x0010, EDeltaReflection = 0x0040,
    EDeltaTransmission = 0x0080, EFrontSide = 0x1000, EBackSide = 0x2000,
    EDelta = EDeltaReflection | EDeltaTransmission, ETransmission = EDeltaTransmission,
    ESmooth = EDiffuseReflection | 0x0008 /* EDiffuseTransmission */ | EGlossyReflection | 0x0020 /* EGlossyTransmission */
};

struct DMaterial {          /* 128 bytes */
    int32_t type, distr, sampleVisible, nonlinear;
    int32_t nested0, nested1; uint32_t typeFlags; int32_t tableOffset; /* into rtTables: 100 floats T12(cos) */
    float alphaU, alphaV, eta, invEta2;
    float specSamplingWeight, fdr /* 1 - diffuse transmittance of the inside table */, pad0, pad1;
    float refl[3], specRefl[3], specTrans[3], condEta[3], condK[3];
    float pad2;
};

struct DShape {             /* 32 bytes */
    uint32_t firstPrim, nPrims; int32_t material, emitter; uint32_t flags; uint32_t pad[3];
};

struct DEmitter {           /* 64 bytes */
    int32_t type, shape; float weight, invArea;
    float radiance[3]; uint32_t cdfOffset;     /* area: offset of the shape's (nPrims+1)-entry area CDF in areaCdf */
    float position[3]; uint32_t firstPrim;
    uint32_t nPrims, shapeFlags;
    uint32_t guideOffset;      /* area emitters on meshes of >= kAreaGuideMinPrims triangles: 1 + offset of the kAreaGuide + 1 guide
                                  entries of the area CDF in areaGuide; 0 = none */
    uint32_t pad;
};

static constexpr int kEnvMaxLevels = 17;     /* 65535 pixels per side at most (envmap.cpp:161-163) */
struct DEnvMap {
    int32_t w, h; float scale, normalization;
    float toWorld[9], toLocal[9];
    float center[3], radius;
    float pixelSizeX, pixelSizeY;
    const uint32_t *texels;    /* w*h texels, 2 words each: RGBA half (A unused)             */
    const float *cdfRows;      /* h+1                                                        */
    const float *cdfCols;      /* (w+1)*h                                                    */
    const float *rowWeights;   /* h                                                          */
    /* guide tables of the two CDF searches: entry b = lower_bound(cdf, b / G), so the first entry >= a sample in bucket b lies
     * in [guide[b], guide[b + 1]] and the binary search starts on that bracket (same result, 3-6 dependent loads, not 11-12) */
    const uint32_t *guideRows; /* kEnvGuideRows + 1                                          */
    const uint32_t *guideCols; /* h * (kEnvGuideCols + 1)                                    */
    /* guide records: per bucket one 32-byte record holding the bracket of the bucket and every CDF
     * entry the search and the sample-reuse step touch, so a CDF lookup is ONE load into registers
     * instead of a guide load, the loads of the search and the two of the reuse step, each waiting on the one before; brackets that do not fit use the two tables above.  Same
     * index as std::lower_bound either way (envmap.cpp:653-658).  Null unless built with -DKKD_ENV_RECORDS (see below: measured, not kept). */
    const uint32_t *recRows;   /* kEnvGuideRows * kEnvRecRowWords                            */
    const uint32_t *recCols;   /* h * kEnvGuideCols * kEnvRecColWords                        */
    /* MIP pyramid of the filtered lookup for camera rays (envmap.cpp:394-407): level l starts at texel levelOfs[l] of
     * texels and is max(1, ceil(w / 2^l)) x max(1, ceil(h / 2^l)) texels; levelOfs[0] = 0 */
    int32_t nLevels; uint32_t levelOfs[kEnvMaxLevels]; uint16_t levelW[kEnvMaxLevels], levelH[kEnvMaxLevels];
    const float *ewaLut;       /* 64 Gaussian weights of the EWA filter (mipmap.h weight table) */
};
static constexpr uint32_t kEmitterGuide = 1024, kAreaGuide = 256, kAreaGuideMinPrims = 64;
/* (powers of two: sample * G is exact.  Every step of a CDF search is a dependent load in kShade, which runs at 4 waves per
 * SIMD: 256 / 64 buckets gave C4's shade stage 380 ms per step, 1024 / 256: 364, 4096 / 1024: 357; 4 KB per envmap row) */
static constexpr uint32_t kEnvGuideRows = 4096, kEnvGuideCols = 1024;
/* guide records (DEnvMap::recRows/recCols), an experiment of round 3 that is NOT built by default (-DKKD_ENV_RECORDS builds it):
 * word 0 = first | n << 24 (n = 0xFF: no record, use the two tables), then cdf[first - 1 ... last + 1] and 2.0f up to the
 * end: brackets of up to W - 3 entries.  Measured on C4 (profiles/r3_ab_env_records.txt): the shade stage takes 371 ms per
 * frame with 32-byte records read whole into registers, 380 ms with 64-byte ones, 390 ms with 64-byte ones searched in
 * memory -- against 357 ms for the two small tables, which stay in the vector cache and L2 while 67-134 MB of records do
 * not, and whose dependent loads kShade hides behind its other work. */
#ifdef KKD_ENV_RECORDS
#ifndef KKD_ENV_REC_ROWS
#define KKD_ENV_REC_ROWS 8
#endif
#ifndef KKD_ENV_REC_COLS
#define KKD_ENV_REC_COLS 8
#endif
#else
#define KKD_ENV_REC_ROWS 0
#define KKD_ENV_REC_COLS 0
#endif
static constexpr uint32_t kEnvRecRowWords = KKD_ENV_REC_ROWS, kEnvRecColWords = KKD_ENV_REC_COLS;

struct DSensor {
    float toWorld[12];         /* rows of the affine camera-to-world matrix                   */
    float s2c[16];             /* sampleToCamera, full projective 4x4                         */
    float nearClip, farClip, invResX, invResY;
    int32_t cropW, cropH, filmHasAlpha, hqEdges;
    float dxCam[3], dyCam[3];  /* m_dx, m_dy: near-plane offsets of the neighbouring pixels (perspective.cpp:159-163) */
};

struct DFilter { float values[32]; float radius, scaleFactor; int32_t border, pad; };

struct DScene {
    const Node4 *nodes; const LeafTri *tris;
    const WideNode *wideNodes; /* same indices as nodes */
    const ProjTri *projTris;   /* non-null: leaf test with the reference's projection arithmetic */
    const float *positions; const float *normals; const float *texcoords; const uint32_t *indices;
    const DShape *shapes; const DMaterial *materials; const DEmitter *emitters;
    const float *emitterCdf;   /* nEmitters+1                                                 */
    const uint32_t *areaGuide;    /* guide tables of the area CDFs of large emissive meshes (DEmitter::guideOffset) */
    const uint32_t *emitterGuide; /* kEmitterGuide + 1 entries: lower_bound(emitterCdf, b / kEmitterGuide), brackets the search */
    const float *areaCdf;      /* concatenated per-emitter-shape triangle area CDFs            */
    const float *rtTables;     /* rough-transmittance tables, 100 floats each                  */
    uint32_t nShapes, nMaterials, nEmitters; int32_t envEmitter;
    float emitterNormalization; float pad[3];
    DEnvMap env; DSensor sensor; DFilter filter;
};

/* ---- wavefront buffers ---- */
struct alignas(16) DRay { float ox, oy, oz, mint, dx, dy, dz, maxt; };     /* 32 bytes = the C ABI ray  */
struct alignas(16) DHit { float t, u, v; uint32_t id; };                   /* id: LeafTri slot (internal) or prim id (ABI) */
static constexpr uint32_t kNoHit = 0xFFFFFFFFu;

} /* namespace kkd */
