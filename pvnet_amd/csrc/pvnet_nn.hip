// pvnet_nn.hip -- brute-force nearest neighbour for the reference's symmetric-object metrics (ADD-S, symmetric 2-D
// projection error), hand-written for gfx950.  C ABI: include/pvnet_nn.h.
//
// Replaces lib/utils/extend_utils/src/nearest_neighborhood.cu:48-117 (findNearestPoint{2D,3D}IdxKernel: one thread per
// query walking the whole reference cloud from global memory) and its host launcher (:120-160).
//
// Layout here: a workgroup = 256 queries (one per lane) x one slice of the reference cloud.  The slice streams through
// LDS in tiles of 256 points (one coalesced load per thread, stored as float4), and every lane reads the tile back with
// BROADCAST ds_read_b128 (all lanes the same address: conflict-free, on the LDS pipe), so the inner loop is VALU only:
// 3 subtractions, 3 multiplications, 2 additions in the reference's float32 order (no FMA contraction: the squared
// distance must round as the reference's expression does for the first-index tie-break to agree), one compare, one
// v_min and one select.  Slices of a large cloud run in different workgroups (a 10 k-query problem alone would fill 40
// of 256 CUs) and meet in one packed (distance bits, index) word per query, combined with a 64-bit atomicMin: float32
// bit patterns of non-negative numbers order like the numbers, and on equal distance the smaller index wins -- exactly
// the reference's strict `dist < min_dist` scan from index 0.
#include <hip/hip_runtime.h>

#include <float.h>
#include <stdint.h>

#include "pvnet_nn.h"
#include "pvnet_vote.h"

namespace {

// one unused VGPR granule beyond what a kernel uses: see PVNET_SPARE_VGPRS in vote_common.h
#define PVNET_SPARE_VGPRS_(r) asm volatile("" ::: "v" #r)
#define PVNET_SPARE_VGPRS(r) PVNET_SPARE_VGPRS_(r)
constexpr int NN_T = 256;  // queries per workgroup = reference points per LDS tile
constexpr unsigned long long NN_NONE = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;  // (FLT_MAX, no index)

template <int DIM>
__global__ __launch_bounds__(NN_T) void nn_search_kernel(const float* __restrict__ ref, const float* __restrict__ que,
                                                         unsigned long long* __restrict__ best, int32_t* __restrict__ idxs,
                                                         int pn1, int pn2, int slice, int exclude_self) {
#pragma clang fp contract(off)
    PVNET_SPARE_VGPRS(63);
    __shared__ float4 s_ref[NN_T];
    const int bi = blockIdx.z;
    const int q = blockIdx.x * NN_T + threadIdx.x;
    const int r0 = blockIdx.y * slice;
    const int r1 = r0 + slice < pn1 ? r0 + slice : pn1;
    const float* rb = ref + (size_t)bi * pn1 * DIM;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (q < pn2) {
        const float* qp = que + ((size_t)bi * pn2 + q) * DIM;
        qx = qp[0];
        qy = qp[1];
        if (DIM == 3) qz = qp[2];
    }
    float min_dist = FLT_MAX;
    int min_idx = -1;
    for (int t0 = r0; t0 < r1; t0 += NN_T) {
        const int n = r1 - t0 < NN_T ? r1 - t0 : NN_T;
        __syncthreads();  // the previous tile has been consumed
        if ((int)threadIdx.x < n) {
            const float* p = rb + (size_t)(t0 + threadIdx.x) * DIM;
            s_ref[threadIdx.x] = make_float4(p[0], p[1], DIM == 3 ? p[2] : 0.f, 0.f);
        }
        __syncthreads();
        const int skip = exclude_self ? q - t0 : -1;  // the tile slot holding this query's own index, if any
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            const float4 r = s_ref[j];
            const float dx = r.x - qx, dy = r.y - qy;
            float dist = dx * dx + dy * dy;
            if (DIM == 3) {
                const float dz = r.z - qz;
                dist = dist + dz * dz;
            }
            const bool lt = dist < min_dist && j != skip;  // strict: the first index wins ties (kernel.cu:75-79)
            min_idx = lt ? t0 + j : min_idx;
            min_dist = lt ? dist : min_dist;
        }
    }
    if (q >= pn2) return;
    if (gridDim.y == 1) {  // one slice: the answer is final (an empty scan keeps the reference's initial index 0)
        idxs[(size_t)bi * pn2 + q] = min_idx < 0 ? 0 : min_idx;
    } else if (min_idx >= 0) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(min_dist) << 32) | (uint32_t)min_idx;
        atomicMin(best + (size_t)bi * pn2 + q, key);
    }
}

__global__ __launch_bounds__(256) void nn_init_kernel(unsigned long long* __restrict__ best, size_t n) {
    PVNET_SPARE_VGPRS(23);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) best[i] = NN_NONE;
}

__global__ __launch_bounds__(256) void nn_final_kernel(const unsigned long long* __restrict__ best,
                                                       int32_t* __restrict__ idxs, size_t n) {
    PVNET_SPARE_VGPRS(23);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const unsigned long long k = best[i];
        idxs[i] = k == NN_NONE ? 0 : (int32_t)(uint32_t)(k & 0xFFFFFFFFull);
    }
}

}  // namespace

extern "C" {

size_t pvnet_nearest_workspace_bytes(int b, int pn2) {
    if (b <= 0 || pn2 <= 0) return 0;
    return ((size_t)b * pn2 * sizeof(unsigned long long) + 255) / 256 * 256;
}

int pvnet_nearest_point_idx(const float* ref_pts, const float* que_pts, int32_t* idxs, int b, int pn1, int pn2, int dim,
                            int exclude_self, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ref_pts || !que_pts || !idxs || b <= 0 || pn1 < 0 || pn2 <= 0 || (dim != 2 && dim != 3)) return PVNET_E_BADARG;
    if (b > 65535) return PVNET_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int qblocks = (pn2 + NN_T - 1) / NN_T;
    const int tiles = (pn1 + NN_T - 1) / NN_T;
    // enough workgroups to fill the chip (~4 per CU), but never slices shorter than one tile
    long long want = (1024 + (long long)qblocks * b - 1) / ((long long)qblocks * b);
    int nslices = (int)(want < 1 ? 1 : (want > tiles ? tiles : want));
    if (nslices < 1) nslices = 1;
    if (nslices > 65535) nslices = 65535;
    const int slice = tiles ? ((tiles + nslices - 1) / nslices) * NN_T : NN_T;
    nslices = pn1 > 0 ? (pn1 + slice - 1) / slice : 1;
    unsigned long long* best = static_cast<unsigned long long*>(workspace);
    const size_t n = (size_t)b * pn2;
    if (nslices > 1) {
        if (!workspace || workspace_bytes < pvnet_nearest_workspace_bytes(b, pn2)) return PVNET_E_WORKSPACE;
        if ((reinterpret_cast<uintptr_t>(workspace) & 7u) != 0) return PVNET_E_BADARG;
        hipLaunchKernelGGL(nn_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, best, n);
    }
    const dim3 grid(qblocks, nslices, b);
    if (dim == 3)
        hipLaunchKernelGGL(nn_search_kernel<3>, grid, dim3(NN_T), 0, s, ref_pts, que_pts, best, idxs, pn1, pn2, slice,
                           exclude_self);
    else
        hipLaunchKernelGGL(nn_search_kernel<2>, grid, dim3(NN_T), 0, s, ref_pts, que_pts, best, idxs, pn1, pn2, slice,
                           exclude_self);
    if (nslices > 1)
        hipLaunchKernelGGL(nn_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, best, idxs, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// The reference's launcher, same symbol, same host-pointer contract (nearest_neighborhood.cu:120-160).  Unlike the
// original's gpuErrchk it never exit()s: on a runtime error the indices are left untouched.
void findNearestPointIdxLauncher(float* ref_pts, float* que_pts, int* idxs, int b, int pn1, int pn2, int dim,
                                 int exclude_self) {
    if (!ref_pts || !que_pts || !idxs || b <= 0 || pn1 < 0 || pn2 <= 0) return;
    float *d_ref = nullptr, *d_que = nullptr;
    int32_t* d_idx = nullptr;
    void* d_ws = nullptr;
    const size_t ws = pvnet_nearest_workspace_bytes(b, pn2);
    bool ok = hipMalloc(&d_ref, sizeof(float) * (size_t)b * (pn1 > 0 ? pn1 : 1) * dim) == hipSuccess &&
              hipMalloc(&d_que, sizeof(float) * (size_t)b * pn2 * dim) == hipSuccess &&
              hipMalloc(&d_idx, sizeof(int32_t) * (size_t)b * pn2) == hipSuccess && hipMalloc(&d_ws, ws) == hipSuccess;
    ok = ok && (pn1 == 0 || hipMemcpy(d_ref, ref_pts, sizeof(float) * (size_t)b * pn1 * dim, hipMemcpyHostToDevice) == hipSuccess);
    ok = ok && hipMemcpy(d_que, que_pts, sizeof(float) * (size_t)b * pn2 * dim, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && pvnet_nearest_point_idx(d_ref, d_que, d_idx, b, pn1, pn2, dim, exclude_self, d_ws, ws, nullptr) == 0;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (ok) (void)hipMemcpy(idxs, d_idx, sizeof(int32_t) * (size_t)b * pn2, hipMemcpyDeviceToHost);
    (void)hipFree(d_ref);
    (void)hipFree(d_que);
    (void)hipFree(d_idx);
    (void)hipFree(d_ws);
}

}  // extern "C"
