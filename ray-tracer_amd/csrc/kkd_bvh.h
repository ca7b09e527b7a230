/* kkd_bvh.h -- host-side 4-wide BVH builder (see kkd_bvh.cpp). */
#pragma once
#include "kkd_types.h"
#include <string>
#include <vector>

namespace kkd {

struct BVH4 {
    struct Info {
        uint64_t nNodes = 0, nLeafTris = 0;
        int maxDepth = 0;    /* in 4-wide nodes                                               */
        int maxStack = 0;    /* upper bound of the traversal stack occupancy (entries)        */
        float bmin[3] = { 0, 0, 0 }, bmax[3] = { 0, 0, 0 };
    };
    std::vector<Node4> nodes;
    std::vector<WideNode> wide;     /* same tree, float boxes (packet traversal of primary rays) */
    std::vector<LeafTri> tris;
    Info info;
    /* positions: nVerts*3, indices: nTris*3, shapeOfPrim: nTris.  false + err on failure. */
    bool build(const float *positions, const uint32_t *indices, uint64_t nTris, const uint32_t *shapeOfPrim, std::string &err);
};

void makeProjTris(const std::vector<LeafTri> &tris, std::vector<ProjTri> &out);

} /* namespace kkd */
