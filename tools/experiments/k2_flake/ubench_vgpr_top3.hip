// Stress test 3 (development aid): the launch pattern of the compaction kernel -- a 14 x 3 x 9 grid of 256-thread workgroups
// of which 9 in 14 exit at once (alloc / free churn on every CU), the rest use LDS + a barrier, hold patterns in v8..v23 of
// exactly 24 allocated VGPRs for a few microseconds and check them.  SPARE adds one unused granule.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_vgpr_top3.hip -o tools/ubench_vgpr_top3.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define SET(r) "v_add_u32 v" #r ", %[seed], " #r "\n\t"
#define CHK(r) "v_sub_u32 v2, v" #r ", %[seed]\n\tv_cmp_ne_u32 vcc, " #r ", v2\n\tv_addc_co_u32 %[bad], vcc, 0, %[bad], vcc\n\t"
#define R8_23(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23)
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"

template <bool SPARE>
__global__ __launch_bounds__(256) void k(unsigned* bad_total, unsigned* per_z, int naps) {
    __shared__ unsigned s[64];
    if (blockIdx.x < 5 || blockIdx.x > 9) return;  // "empty segments"
    if (threadIdx.x < 64) s[threadIdx.x] = threadIdx.x * 3u;
    __syncthreads();
    const unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + blockIdx.z * 977u + s[threadIdx.x & 63u];
    unsigned bad = 0;
    if ((threadIdx.x >> 6) == 3 && (blockIdx.x & 1)) return;  // a wave without work leaves right after the barrier
    asm volatile(R8_23(SET) : : [seed] "v"(seed) : CLOB);
    if (SPARE) asm volatile("" ::: "v31");
    for (int i = 0; i < naps; ++i) asm volatile("s_sleep 4" ::: "memory");
    asm volatile(R8_23(CHK) : [bad] "+v"(bad) : [seed] "v"(seed) : "v2", "vcc", CLOB);
    if (bad) { atomicAdd(bad_total, bad); atomicAdd(per_z + blockIdx.z, 1u); }
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 64);
    for (int spare = 0; spare < 2; ++spare)
        for (int naps : {2, 20, 200}) {
            (void)hipMemset(d, 0, 64);
            for (int r = 0; r < 300; ++r) {
                if (spare) hipLaunchKernelGGL(k<true>, dim3(14, 3, 9), dim3(256), 0, 0, d, d + 1, naps);
                else hipLaunchKernelGGL(k<false>, dim3(14, 3, 9), dim3(256), 0, 0, d, d + 1, naps);
            }
            unsigned h[16];
            (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("%s naps %3d: %u registers lost their contents; lanes per blockIdx.z:", spare ? "spare granule " : "24 of 24 used ", naps, h[0]);
            for (int z = 0; z < 9; ++z) printf(" %u", h[1 + z]);
            printf("\n");
        }
    return 0;
}
