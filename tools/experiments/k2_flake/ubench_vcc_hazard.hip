// Micro-test (development aid): "VALU writes VCC -> VALU reads VCC" on gfx950.  LLVM pads this pair with 2 wait states
// (s_nop 1).  The compaction kernel's rank search (v_lshrrev_b64 / v_and / v_bcnt / v_cmp / s_nop 1 / v_cndmask ...) returned
// wrong selections on nearly idle CUs; this test replays that instruction sequence with N wait states between the compare
// and the select, one wave per workgroup on an otherwise idle chip and on a full one, and checks every result.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_vcc_hazard.hip -o tools/ubench_vcc_hazard.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define STEP(NOPS)                                                                                              \
    asm volatile("v_bcnt_u32_b32 %[c], %[tl], 0\n\t"                                                          \
                 "v_cmp_lt_i32 vcc, %[r], %[c]\n\t" NOPS                                                      \
                 "v_cndmask_b32_e64 %[sel], %[c], 0, vcc\n\t"                                                 \
                 "v_sub_u32 %[r2], %[r], %[sel]\n\t"                                                          \
                 "v_cndmask_b32_e64 %[add], 2, 0, vcc"                                                         \
                 : [c] "=&v"(c), [sel] "=&v"(sel), [r2] "=&v"(r2), [add] "=&v"(add)                            \
                 : [tl] "v"(tl), [r] "v"(r)                                                                    \
                 : "vcc")

template <int N>
__global__ __launch_bounds__(64) void k(unsigned* bad, int reps) {
    unsigned nbad = 0;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < reps; ++i) {
        x = x * 1664525u + 1013904223u;
        const unsigned long long wd = ((unsigned long long)x << 32) | (x * 2246822519u);
        const unsigned sh = (x >> 7) & 62u;
        const int r = (int)((x >> 13) & 3u);
        const unsigned tl = (unsigned)((wd >> sh) & 3ull);  // v_lshrrev_b64 + v_and, as in the kernel
        unsigned add;
        int c, sel, r2;
        if (N == 0) STEP("");
        else if (N == 1) STEP("s_nop 0\n\t");
        else if (N == 2) STEP("s_nop 1\n\t");
        else STEP("s_nop 7\n\t");
        const int ce = __popc(tl);
        const bool lt = r < ce;
        const int r2e = r - (lt ? 0 : ce);
        const unsigned adde = lt ? 0u : 2u;
        if (c != ce || r2 != r2e || add != adde) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned* bad;
    hipMalloc(&bad, 4);
    const int grids[3] = {8, 256, 16384};
    for (int g = 0; g < 3; ++g)
        for (int n = 0; n < 4; ++n) {
            hipMemset(bad, 0, 4);
            const int reps = g == 2 ? 20000 : 400000;
            if (n == 0) hipLaunchKernelGGL(k<0>, dim3(grids[g]), dim3(64), 0, 0, bad, reps);
            if (n == 1) hipLaunchKernelGGL(k<1>, dim3(grids[g]), dim3(64), 0, 0, bad, reps);
            if (n == 2) hipLaunchKernelGGL(k<2>, dim3(grids[g]), dim3(64), 0, 0, bad, reps);
            if (n == 3) hipLaunchKernelGGL(k<3>, dim3(grids[g]), dim3(64), 0, 0, bad, reps);
            unsigned h = 0;
            hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            printf("grid %5d x 1 wave, %s wait states between v_cmp and v_cndmask: %u wrong of %llu\n", grids[g],
                   n == 0 ? "0" : n == 1 ? "1" : n == 2 ? "2 (LLVM's pad)" : "8", h, (unsigned long long)grids[g] * 64 * reps);
        }
    return 0;
}
