/* kkd_gpu_build.h -- device-side acceleration-structure build (kkd_gpu_build.hip), used with KKD_FLAG_GPU_BUILD. */
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "kkd_types.h"

namespace kkd {

struct GpuBuildResult {
    uint64_t nNodes;            /* 4-wide nodes written                                             */
    int maxStack, maxDepth;     /* traversal stack bound / depth in 4-wide nodes, as BVH4::Info     */
    float cmin[3], cmax[3];     /* bounds of the triangle centroids                                 */
};

/* Builds Node4 / WideNode / LeafTri records for the nTris triangles (dPos, dIdx: device memory) into dNodes / dWide (room for
 * nodeCapacity records; nTris is always enough) and dTris (nTris records).  dShapes: the DShape table on the device (shape and
 * emitter ids of the leaf records).  Synchronises the stream.  hipSuccess, or an error with its text in err. */
hipError_t gpuBuildBvh(hipStream_t st, const float *dPos, const uint32_t *dIdx, uint32_t nTris, const DShape *dShapes, uint32_t nShapes,
                       Node4 *dNodes, WideNode *dWide, LeafTri *dTris, uint32_t nodeCapacity, GpuBuildResult *out, std::string &err);

/* TriAccel records (kkd_bvh.cpp: makeProjTris, the reference's triaccel.h) for nTris leaf records, both on the device; same
 * single-precision operations in the same order as the host version, so the two tables are bit-equal. */
void gpuMakeProjTris(hipStream_t st, const LeafTri *dTris, uint32_t nTris, ProjTri *dProj);

} /* namespace kkd */
