/*
 * kkd_host.h -- host-side flattening of a kkd_scene_desc into the arrays the GPU keeps resident.
 * Replaces Scene::initialize / ShapeKDTree::build / EnvironmentMap::configure / BSDF::configure /
 * ReconstructionFilter::configure of the reference (see kkd_host.cpp for file:line).
 */
#pragma once
#include "kkd_types.h"
#include "kkd_bvh.h"
#include "../../include/kkdirect.h"
#include <string>
#include <vector>

namespace kkd {

/* text behind kkd_last_error(NULL): the last failure of a call that has no context (kkd_create, mesh loaders) */
extern std::string g_globalError;

struct HostScene {
    BVH4 bvh;
    std::vector<uint32_t> shapeOfPrim;
    std::vector<DShape> shapes;
    std::vector<DMaterial> materials;
    std::vector<DEmitter> emitters;
    std::vector<float> emitterCdf, areaCdf, rtTables;
    std::vector<uint32_t> emitterGuide, areaGuide;
    int envEmitter = -1;
    float emitterNormalization = 0.f;
    /* envmap */
    DEnvMap env{};                   /* pointer members filled after upload */
    std::vector<uint16_t> envTexels; /* RGBA half, all MIP levels back to back */
    std::vector<float> envEwaLut;
    std::vector<float> envCdfRows, envCdfCols, envRowWeights;
    std::vector<uint32_t> envGuideRows, envGuideCols;
    std::vector<uint32_t> envRecRows, envRecCols;      /* 64-byte guide records (kkd_types.h) */
    DSensor sensor{};
    DFilter filter{};
    uint64_t nVerts = 0, nTris = 0;
    bool hasNormals = false, hasTexcoords = false;
};

/* Validates the description and fills everything except device pointers.  false + err on failure. */
/* buildBvh = false: everything but the acceleration structure (KKD_FLAG_GPU_BUILD builds it on the device) */
bool flattenScene(const kkd_scene_desc *sd, HostScene &hs, std::string &err, int *errCode, bool buildBvh = true);

/* rough-dielectric transmittance T(cosTheta) for one (distribution, alpha, eta): 100 knots at
 * cosTheta^(1/4) = i/99 plus the diffuse transmittance (what bsdfs/rtrans.h yields after setEta/setAlpha). */
void roughTransmittanceTable(int distr, float alpha, float eta, float *trans100, float *diffTrans);

void configureFilter(int type, float radius, const float *param, DFilter &f);
/* halton sampler: [1024 primes][1024 offsets][Faure permutations as uint16] (libcore/qmc.cpp:23, samplers/faure.cpp:25-142) */
void haltonTables(std::vector<uint32_t> &words);
/* HaltonSampler::setFilmResolution(res, blocked = true) + the end of the dimensions reserved to nArrays 2D arrays (halton.cpp:240-279) */
void haltonSetup(int resX, int resY, uint32_t nArrays, bool faure, const uint32_t *tables, HaltonSetup &h);
/* HammersleySampler::setFilmResolution(res, blocked = true) for a sampler of sampleCount samples (hammersley.cpp:181-197) */
void hammersleySetup(int resX, int resY, uint32_t sampleCount, bool faure, const uint32_t *tables, HaltonSetup &h);

/* sobol sampler: the 1024 x 52 generator matrices (32 bits per column) regenerated from Joe and Kuo's direction numbers; their CRC-32;
 * the tables of one render (matrices + the pixel look-up of resolution 2^m: false if the look-up matrix is singular, which it never
 * is); SobolSampler::setFilmResolution + the end of the dimensions reserved to nArrays 2D arrays (samplers/sobol.cpp:146-179) */
const std::vector<uint32_t> &sobolMatrices();
uint32_t crc32Words(const uint32_t *w, size_t n);
bool sobolTables(uint32_t m, std::vector<uint32_t> &words);
void sobolSetup(int resX, int resY, uint32_t nArrays, const uint32_t *tables, SobolSetup &s);

uint16_t floatToHalf(float f);
float halfToFloat(uint16_t h);

} /* namespace kkd */
