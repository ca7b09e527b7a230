/*
 * kkd_gpu_build.hip -- acceleration-structure build on the device (KKD_FLAG_GPU_BUILD).
 *
 * Replaces, like the host builder (kkd_bvh.cpp), ShapeKDTree::build of the reference (src/librender/skdtree.cpp:68-110:
 * SAH kd-tree build + TriAccel table, seconds to minutes on the host for 10 M triangles) -- here as a linear BVH that is
 * built where the triangles already are: Morton codes of the triangle centroids, one radix sort, the hierarchy of Karras
 * ("Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012) from the sorted codes, boxes fitted
 * bottom-up, then the same collapse to four children per node (open the child with the largest area) and the same node
 * encoding (kkd_bvh_encode.h) as the host builder, so the traversal kernels read identical record layouts:
 *   kPrimBounds   triangle boxes + bounds of the centroids
 *   kMortonCodes  63-bit codes (21 bits per axis)
 *   rocprim::radix_sort_pairs
 *   kHierarchy    n - 1 inner nodes of the binary radix tree (duplicate codes are told apart by their position)
 *   kFitBoxes     leaves walk up, the second child to arrive at a node merges the boxes
 *   kCollapse     level by level from the root: a 4-wide node per launch item, children allocated with one atomic
 *   kLeafTris     LeafTri records in sorted (= leaf) order
 * A Morton-ordered tree is built in milliseconds but is looser than the host's binned-SAH tree (more node visits per ray),
 * so the host build stays the default for renders that run many samples per pixel; hits do not depend on which one is used
 * (tests/test_gpu_build.py).
 */
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <string>
#include <vector>
#include "kkd_gpu_build.h"
#include "kkd_bvh_encode.h"
#include "kkd_kernels.h"

namespace kkd {

namespace {

struct Box3 { float mn[3], mx[3]; };

/* order-preserving map float -> uint32 for atomicMin / atomicMax */
__device__ __forceinline__ uint32_t orderedBits(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ __forceinline__ float fromOrderedBits(uint32_t b) {
    const uint32_t u = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
    float f; memcpy(&f, &u, 4); return f;
}

/* work[0..2] = min of the centroids (ordered bits), work[3..5] = max, work[6] = non-finite flag */
__global__ void __launch_bounds__(256) kPrimBounds(const float *__restrict__ pos, const uint32_t *__restrict__ idx, uint32_t n,
                                                   float4 *__restrict__ boxMin, float4 *__restrict__ boxMax, uint32_t *__restrict__ work) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    bool finite = true;
    if (p < n) {
        for (int k = 0; k < 3; ++k) {
            const float *v = pos + 3u * (size_t)idx[3u * (size_t)p + k];
            for (int a = 0; a < 3; ++a) { const float x = v[a]; finite = finite && isfinite(x); mn[a] = fminf(mn[a], x); mx[a] = fmaxf(mx[a], x); }
        }
        boxMin[p] = make_float4(mn[0], mn[1], mn[2], 0.f); boxMax[p] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
    float c[3];
    for (int a = 0; a < 3; ++a) c[a] = p < n ? 0.5f * (mn[a] + mx[a]) : 0.f;
    /* wave reduction of the centroid bounds, one atomic pair per wave and axis */
    for (int a = 0; a < 3; ++a) {
        float lo = p < n ? c[a] : INFINITY, hi = p < n ? c[a] : -INFINITY;
        for (int m = 32; m > 0; m >>= 1) { lo = fminf(lo, __shfl_xor(lo, m)); hi = fmaxf(hi, __shfl_xor(hi, m)); }
        if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(work + a, orderedBits(lo)); atomicMax(work + 3 + a, orderedBits(hi)); }
    }
    if (__ballot(p < n && !finite) != 0ull && (threadIdx.x & 63) == 0) atomicOr(work + 6, 1u);
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {      /* bits of v at every third position */
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull; x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256) kMortonCodes(const float4 *__restrict__ boxMin, const float4 *__restrict__ boxMax, uint32_t n, const uint32_t *__restrict__ work,
                                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float4 a = boxMin[p], b = boxMax[p];
    const float c[3] = { 0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z) };
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = fromOrderedBits(work[k]), hi = fromOrderedBits(work[3 + k]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        q[k] = (uint32_t)fminf(fmaxf(t * 2097152.0f, 0.f), 2097151.0f);
    }
    keys[p] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    vals[p] = p;
}

/* length of the common prefix of the codes at sorted positions i and j; equal codes are told apart by their positions */
__device__ __forceinline__ int delta(const uint64_t *__restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

/* child references of the binary tree: bit 31 = leaf (sorted position), else inner node index */
__global__ void __launch_bounds__(256) kHierarchy(const uint64_t *__restrict__ keys, int n, uint32_t *__restrict__ left, uint32_t *__restrict__ right,
                                                  uint32_t *__restrict__ parentOfInner, uint32_t *__restrict__ parentOfLeaf) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) { lmax <<= 1; if (lmax > (1 << 30)) break; }
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t l_ = lo == gamma ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t r_ = hi == gamma + 1 ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = l_; right[i] = r_;
    if (l_ & kLeafBit) parentOfLeaf[gamma] = (uint32_t)i; else parentOfInner[gamma] = (uint32_t)i;
    if (r_ & kLeafBit) parentOfLeaf[gamma + 1] = (uint32_t)i; else parentOfInner[gamma + 1] = (uint32_t)i;
    if (i == 0) parentOfInner[0] = 0xFFFFFFFFu;
}

/* boxes of the inner nodes: every leaf walks towards the root; the first child to reach a node stops, the second one
 * (which then sees both children's boxes: fence + atomic + fence) merges them and goes on.  Every walk ends: at a node
 * where it is first, or at the root. */
__global__ void __launch_bounds__(256) kFitBoxes(const float4 *__restrict__ primMin, const float4 *__restrict__ primMax, const uint32_t *__restrict__ sortedPrim, int n,
                                                 const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                 const uint32_t *__restrict__ parentOfInner, const uint32_t *__restrict__ parentOfLeaf,
                                                 float4 *innerMin, float4 *innerMax, uint32_t *__restrict__ arrived) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= n || n < 2) return;
    uint32_t node = parentOfLeaf[k];
    for (int guard = 0; guard < 4096 && node != 0xFFFFFFFFu; ++guard) {
        __threadfence();
        if (atomicAdd(arrived + node, 1u) == 0u) return;
        __threadfence();
        float4 mn[2], mx[2];
        const uint32_t ch[2] = { left[node], right[node] };
        for (int c = 0; c < 2; ++c) {
            if (ch[c] & kLeafBit) { const uint32_t p = sortedPrim[ch[c] & ~kLeafBit]; mn[c] = primMin[p]; mx[c] = primMax[p]; }
            else {
                /* written by another wave a moment ago: read past this CU's vector cache */
                const float *a = reinterpret_cast<const float *>(innerMin + ch[c]), *b = reinterpret_cast<const float *>(innerMax + ch[c]);
                mn[c] = make_float4(__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(a + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
                mx[c] = make_float4(__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                    __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
            }
        }
        innerMin[node] = make_float4(fminf(mn[0].x, mn[1].x), fminf(mn[0].y, mn[1].y), fminf(mn[0].z, mn[1].z), 0.f);
        innerMax[node] = make_float4(fmaxf(mx[0].x, mx[1].x), fmaxf(mx[0].y, mx[1].y), fmaxf(mx[0].z, mx[1].z), 0.f);
        node = parentOfInner[node];
    }
}

struct CollapseItem { uint32_t node /* binary inner node */, out /* 4-wide node index */; int32_t depth, stackAbove; };

/* stats[0] = 4-wide nodes allocated, stats[1] = max stack bound, stats[2] = max depth, stats[3] = items queued for the next level,
 * stats[4] = overflow flag */
__global__ void __launch_bounds__(256) kCollapse(const CollapseItem *__restrict__ in, uint32_t nIn, CollapseItem *__restrict__ next, uint32_t nextCapacity,
                                                 const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                 const float4 *__restrict__ innerMin, const float4 *__restrict__ innerMax,
                                                 const float4 *__restrict__ primMin, const float4 *__restrict__ primMax, const uint32_t *__restrict__ sortedPrim,
                                                 Node4 *__restrict__ nodes, WideNode *__restrict__ wide, uint32_t nodeCapacity, uint32_t *stats) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nIn) return;
    const CollapseItem it = in[t];
    uint32_t kids[4]; int nk = 2;
    kids[0] = left[it.node]; kids[1] = right[it.node];
    auto area = [&](uint32_t ref) {
        const float4 a = innerMin[ref], b = innerMax[ref];
        const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
        return 2.f * (dx * dy + dy * dz + dz * dx);
    };
    while (nk < 4) {                                   /* open the inner child with the largest area (as Collapser::emit) */
        int best = -1; float bestArea = -1.f;
        for (int k = 0; k < nk; ++k) if (!(kids[k] & kLeafBit)) { const float ar = area(kids[k]); if (ar > bestArea) { bestArea = ar; best = k; } }
        if (best < 0) break;
        const uint32_t c = kids[best];
        kids[best] = left[c]; kids[nk++] = right[c];
    }
    Node4F nd; nd.nk = nk;
    for (int k = 0; k < 4; ++k) { nd.minx[k] = nd.miny[k] = nd.minz[k] = INFINITY; nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = INFINITY; nd.child[k] = 0u; }
    const int stackHere = it.stackAbove + nk - 1;
    atomicMax(stats + 1, (uint32_t)stackHere); atomicMax(stats + 2, (uint32_t)it.depth);
    for (int k = 0; k < nk; ++k) {
        float4 a, b;
        if (kids[k] & kLeafBit) {
            const uint32_t slot = kids[k] & ~kLeafBit, p = sortedPrim[slot];
            a = primMin[p]; b = primMax[p];
            nd.child[k] = kLeafBit | slot;              /* one triangle per leaf: count - 1 = 0 in bits 28..30 */
        } else {
            a = innerMin[kids[k]]; b = innerMax[kids[k]];
            const uint32_t idx = atomicAdd(stats + 0, 1u), q = atomicAdd(stats + 3, 1u);
            if (idx >= nodeCapacity || q >= nextCapacity) { atomicOr(stats + 4, 1u); nd.child[k] = 0u; nd.nk = k; break; }
            nd.child[k] = idx;
            CollapseItem ni; ni.node = kids[k]; ni.out = idx; ni.depth = it.depth + 1; ni.stackAbove = stackHere;
            next[q] = ni;
        }
        nd.minx[k] = a.x; nd.miny[k] = a.y; nd.minz[k] = a.z; nd.maxx[k] = b.x; nd.maxy[k] = b.y; nd.maxz[k] = b.z;
    }
    if (it.out < nodeCapacity) { quantizeNode(nd, nodes[it.out]); widenNode(nd, wide[it.out]); }
}

/* a scene of one triangle: a root with one leaf child (as BVH4::build wraps it) */
__global__ void kSingleLeafRoot(const float4 *__restrict__ primMin, const float4 *__restrict__ primMax, Node4 *__restrict__ nodes, WideNode *__restrict__ wide) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Node4F nd; nd.nk = 1;
    for (int k = 0; k < 4; ++k) { nd.minx[k] = nd.miny[k] = nd.minz[k] = INFINITY; nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = INFINITY; nd.child[k] = 0u; }
    const float4 a = primMin[0], b = primMax[0];
    nd.minx[0] = a.x; nd.miny[0] = a.y; nd.minz[0] = a.z; nd.maxx[0] = b.x; nd.maxy[0] = b.y; nd.maxz[0] = b.z;
    nd.child[0] = kLeafBit | 0u;
    quantizeNode(nd, nodes[0]); widenNode(nd, wide[0]);
}

__global__ void __launch_bounds__(256) kLeafTris(const float *__restrict__ pos, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ sortedPrim, uint32_t n,
                                                 const DShape *__restrict__ shapes, uint32_t nShapes, LeafTri *__restrict__ tris) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = sortedPrim[k];
    uint32_t lo = 0, hi = nShapes;                   /* the shape whose [firstPrim, firstPrim + nPrims) holds p: shapes tile the array in order */
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (shapes[mid].firstPrim <= p) lo = mid; else hi = mid; }
    const float *a = pos + 3u * (size_t)idx[3u * (size_t)p], *b = pos + 3u * (size_t)idx[3u * (size_t)p + 1], *c = pos + 3u * (size_t)idx[3u * (size_t)p + 2];
    LeafTri t;
    t.v0x = a[0]; t.v0y = a[1]; t.v0z = a[2]; t.prim = p;
    t.v1x = b[0]; t.v1y = b[1]; t.v1z = b[2]; t.shape = lo;
    t.v2x = c[0]; t.v2y = c[1]; t.v2z = c[2]; t.emitter = shapes[lo].emitter >= 0 ? (uint32_t)shapes[lo].emitter + 1u : 0u;
    tris[k] = t;
}

/* makeProjTris (kkd_bvh.cpp) on the device: every operation rounded on its own (no contraction), IEEE division */
__global__ void __launch_bounds__(256) kProjTris(const LeafTri *__restrict__ tris, uint32_t n, ProjTri *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const LeafTri t = tris[i];
    ProjTri r; r.n_u = r.n_v = r.n_d = 0.f; r.k = 0u; r.a_u = r.a_v = r.b_nu = r.b_nv = 0.f; r.c_nu = r.c_nv = 0.f; r.prim = t.prim; r.pad = 0u;
    const float A[3] = { t.v0x, t.v0y, t.v0z };
    const float b[3] = { __fsub_rn(t.v2x, t.v0x), __fsub_rn(t.v2y, t.v0y), __fsub_rn(t.v2z, t.v0z) };
    const float c[3] = { __fsub_rn(t.v1x, t.v0x), __fsub_rn(t.v1y, t.v0y), __fsub_rn(t.v1z, t.v0z) };
    const float N[3] = { __fsub_rn(__fmul_rn(c[1], b[2]), __fmul_rn(c[2], b[1])), __fsub_rn(__fmul_rn(c[2], b[0]), __fmul_rn(c[0], b[2])),
                         __fsub_rn(__fmul_rn(c[0], b[1]), __fmul_rn(c[1], b[0])) };
    uint32_t k = 0;
    if (fabsf(N[1]) > fabsf(N[k])) k = 1;
    if (fabsf(N[2]) > fabsf(N[k])) k = 2;
    const uint32_t u = k == 2u ? 0u : k + 1u, v = u == 2u ? 0u : u + 1u;      /* waldModulo: (k + 1) % 3, (k + 2) % 3 */
    auto pick = [](const float *a, uint32_t j) { return j == 0u ? a[0] : (j == 1u ? a[1] : a[2]); };
    const float n_k = pick(N, k), bu = pick(b, u), bv = pick(b, v), cu = pick(c, u), cv = pick(c, v);
    const float denom = __fsub_rn(__fmul_rn(bu, cv), __fmul_rn(bv, cu));
    if (denom == 0.f) { r.k = 3u; out[i] = r; return; }
    r.k = k;
    r.n_u = __fdiv_rn(pick(N, u), n_k); r.n_v = __fdiv_rn(pick(N, v), n_k);
    r.n_d = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(A[0], N[0]), __fmul_rn(A[1], N[1])), __fmul_rn(A[2], N[2])), n_k);
    r.b_nu = __fdiv_rn(bu, denom); r.b_nv = __fdiv_rn(-bv, denom);
    r.a_u = pick(A, u); r.a_v = pick(A, v);
    r.c_nu = __fdiv_rn(cv, denom); r.c_nv = __fdiv_rn(-cu, denom);
    out[i] = r;
}

struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) hipFree(p); }
    template <typename T> hipError_t get(T **out, size_t count) {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
};

} /* namespace */

#define KKD_GB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return e_; } } while (0)

hipError_t gpuBuildBvh(hipStream_t st, const float *dPos, const uint32_t *dIdx, uint32_t nTris, const DShape *dShapes, uint32_t nShapes,
                       Node4 *dNodes, WideNode *dWide, LeafTri *dTris, uint32_t nodeCapacity, GpuBuildResult *out, std::string &err) {
    if (nTris == 0 || nTris > kLeafStartMask || nShapes == 0 || nodeCapacity < 1) { err = "gpuBuildBvh: bad sizes"; return hipErrorInvalidValue; }
    Scratch s;
    const uint32_t n = nTris, blocks = (n + 255) / 256;
    float4 *primMin, *primMax, *innerMin, *innerMax; uint32_t *work, *valsIn, *valsOut, *left, *right, *parentInner, *parentLeaf, *arrived, *stats;
    uint64_t *keysIn, *keysOut; CollapseItem *queueA, *queueB;
    KKD_GB(s.get(&primMin, n)); KKD_GB(s.get(&primMax, n)); KKD_GB(s.get(&innerMin, n)); KKD_GB(s.get(&innerMax, n));
    KKD_GB(s.get(&work, 8)); KKD_GB(s.get(&keysIn, n)); KKD_GB(s.get(&keysOut, n)); KKD_GB(s.get(&valsIn, n)); KKD_GB(s.get(&valsOut, n));
    KKD_GB(s.get(&left, n)); KKD_GB(s.get(&right, n)); KKD_GB(s.get(&parentInner, n)); KKD_GB(s.get(&parentLeaf, n)); KKD_GB(s.get(&arrived, n));
    KKD_GB(s.get(&stats, 8)); KKD_GB(s.get(&queueA, n)); KKD_GB(s.get(&queueB, n));
    const uint32_t workInit[8] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u };
    KKD_GB(hipMemcpyAsync(work, workInit, sizeof(workInit), hipMemcpyHostToDevice, st));
    KKD_GB(hipMemsetAsync(arrived, 0, (size_t)n * sizeof(uint32_t), st));
    hipLaunchKernelGGL(kPrimBounds, dim3(blocks), dim3(256), 0, st, dPos, dIdx, n, primMin, primMax, work);
    hipLaunchKernelGGL(kMortonCodes, dim3(blocks), dim3(256), 0, st, primMin, primMax, n, work, keysIn, valsIn);
    KKD_GB(hipGetLastError());
    size_t tempBytes = 0;
    KKD_GB(rocprim::radix_sort_pairs(nullptr, tempBytes, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0u, 63u, st));
    unsigned char *temp; KKD_GB(s.get(&temp, tempBytes));
    KKD_GB(rocprim::radix_sort_pairs(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0u, 63u, st));
    uint32_t hostWork[8];
    KKD_GB(hipMemcpyAsync(hostWork, work, sizeof(hostWork), hipMemcpyDeviceToHost, st));
    KKD_GB(hipStreamSynchronize(st));
    if (hostWork[6]) { err = "non-finite vertex position"; return hipErrorInvalidValue; }
    uint32_t hostStats[8] = { 1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };      /* node 0 = the root */
    if (n == 1) {
        hipLaunchKernelGGL(kSingleLeafRoot, dim3(1), dim3(64), 0, st, primMin, primMax, dNodes, dWide);
        hostStats[1] = 0; hostStats[2] = 1;
    } else {
        hipLaunchKernelGGL(kHierarchy, dim3((n - 1 + 255) / 256), dim3(256), 0, st, keysOut, (int)n, left, right, parentInner, parentLeaf);
        hipLaunchKernelGGL(kFitBoxes, dim3(blocks), dim3(256), 0, st, primMin, primMax, valsOut, (int)n, left, right, parentInner, parentLeaf, innerMin, innerMax, arrived);
        KKD_GB(hipGetLastError());
        const CollapseItem rootItem = { 0u, 0u, 0, 0 };
        KKD_GB(hipMemcpyAsync(queueA, &rootItem, sizeof(rootItem), hipMemcpyHostToDevice, st));
        KKD_GB(hipMemcpyAsync(stats, hostStats, sizeof(hostStats), hipMemcpyHostToDevice, st));
        uint32_t nIn = 1;
        /* level by level; a binary radix tree over n leaves is at most n - 1 levels deep, 4-wide levels half of that */
        for (uint32_t level = 0; nIn > 0 && level <= n; ++level) {
            hipLaunchKernelGGL(kCollapse, dim3((nIn + 255) / 256), dim3(256), 0, st, queueA, nIn, queueB, n, left, right, innerMin, innerMax, primMin, primMax, valsOut,
                               dNodes, dWide, nodeCapacity, stats);
            KKD_GB(hipGetLastError());
            KKD_GB(hipMemcpyAsync(hostStats, stats, sizeof(hostStats), hipMemcpyDeviceToHost, st));
            KKD_GB(hipStreamSynchronize(st));
            if (hostStats[4]) { err = "gpuBuildBvh: node buffer too small"; return hipErrorOutOfMemory; }
            nIn = hostStats[3];
            KKD_GB(hipMemsetAsync(stats + 3, 0, sizeof(uint32_t), st));
            std::swap(queueA, queueB);
        }
    }
    hipLaunchKernelGGL(kLeafTris, dim3(blocks), dim3(256), 0, st, dPos, dIdx, valsOut, n, dShapes, nShapes, dTris);
    KKD_GB(hipGetLastError());
    KKD_GB(hipStreamSynchronize(st));
    if (out) {
        out->nNodes = hostStats[0]; out->maxStack = (int)hostStats[1]; out->maxDepth = (int)hostStats[2];
        for (int a = 0; a < 3; ++a) { out->cmin[a] = fromOrderedBits(hostWork[a]); out->cmax[a] = fromOrderedBits(hostWork[3 + a]); }
    }
    return hipSuccess;
}

void gpuMakeProjTris(hipStream_t st, const LeafTri *dTris, uint32_t nTris, ProjTri *dProj) {
    if (nTris) hipLaunchKernelGGL(kProjTris, dim3((nTris + 255u) / 256u), dim3(256), 0, st, dTris, nTris, dProj);
}

} /* namespace kkd */
