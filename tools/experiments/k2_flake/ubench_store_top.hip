// Micro-test (development aid): global_store_dwordx4 whose 128-bit data are the LAST four VGPRs of the wave's allocation,
// issued right before the wave ends, while other workgroups start on the CU.  TOP = 1: data in v[20:23] of 24 allocated;
// TOP = 0: data in v[8:11].  Every 16-byte record is checked on the host.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_store_top.hip -o tools/ubench_store_top.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int TOP>
__global__ __launch_bounds__(256) void k(uint4* out, int spin) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    uint4* p = out + gid;
    // a little LDS + barrier + block-dependent delay, as in the compaction kernel
    __shared__ unsigned s[64];
    if (threadIdx.x < 64) s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned acc = s[(gid * 7u) & 63u];
    for (int i = 0; i < (int)((blockIdx.x * 2654435761u >> 22) % (unsigned)spin); ++i) acc = acc * 1664525u + s[acc & 63u];
    if (acc == 0x12345u) return;
    if (TOP)
        asm volatile("v_mov_b32 v20, %[g]\n\tv_xor_b32 v21, 0x5bd1e995, %[g]\n\tv_not_b32 v22, %[g]\n\tv_add_u32 v23, 77, %[g]\n\t"
                     "global_store_dwordx4 %[p], v[20:23], off" : : [g] "v"(gid), [p] "v"(p) : "v20", "v21", "v22", "v23", "memory");
    else
        asm volatile("v_mov_b32 v8, %[g]\n\tv_xor_b32 v9, 0x5bd1e995, %[g]\n\tv_not_b32 v10, %[g]\n\tv_add_u32 v11, 77, %[g]\n\t"
                     "global_store_dwordx4 %[p], v[8:11], off" : : [g] "v"(gid), [p] "v"(p) : "v8", "v9", "v10", "v11", "v23", "memory");
}

int main() {
    const int blocks = 2048, n = blocks * 256;
    uint4* d;
    (void)hipMalloc(&d, (size_t)n * 16);
    std::vector<uint4> h(n);
    for (int top = 1; top >= 0; --top)
        for (int spin : {1, 64, 2048}) {
            long bad = 0;
            for (int rep = 0; rep < 40; ++rep) {
                (void)hipMemset(d, 0xFF, (size_t)n * 16);
                if (top) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, spin);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, spin);
                (void)hipMemcpy(h.data(), d, (size_t)n * 16, hipMemcpyDeviceToHost);
                for (int i = 0; i < n; ++i) {
                    const unsigned g = (unsigned)i;
                    if (h[i].x != g || h[i].y != (g ^ 0x5bd1e995u) || h[i].z != ~g || h[i].w != g + 77u) ++bad;
                }
            }
            printf("store data in %s, spin %4d: %ld wrong records of %ld\n", top ? "v[20:23] (top of 24)" : "v[8:11]            ", spin, bad, 40L * n);
        }
    return 0;
}
