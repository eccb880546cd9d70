// Micro-test 5 (development aid): write-after-write on the HIGH half of a v_lshrrev_b64 result.
// hipcc gives the dead high half of a 64-bit shift to the next value it computes (compact_kernel<false,1>:
// `v_lshrrev_b64 v[16:17], v20, v[14:15] ; v_sub ; v_and ; v_add3_u32 v17, ...`).  If the shift's high half lands AFTER the
// later 32-bit write, the later value is lost.  N independent VALU instructions between the two writes.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_shift64_waw.hip -o tools/ubench_shift64_waw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CASE(FILL)                                                                                                   \
    asm volatile("v_lshrrev_b64 v[40:41], %[sh], %[wd]\n\t" FILL "v_add3_u32 v41, %[a], %[b], %[c]\n\t"               \
                 "s_nop 7\n\t"                                                                                       \
                 "v_mov_b32 %[out], v41\n\t"                                                                         \
                 "v_mov_b32 %[lo], v40"                                                                              \
                 : [out] "=&v"(out), [lo] "=&v"(lo), [f1] "=&v"(f1), [f2] "=&v"(f2)                                  \
                 : [sh] "v"(sh), [wd] "v"(wd), [a] "v"(a), [b] "v"(b), [c] "v"(c)                                    \
                 : "v40", "v41")

template <int N>
__global__ __launch_bounds__(64) void k(unsigned* bad, unsigned* bad_lo, int reps) {
    unsigned nbad = 0, nbadlo = 0;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < reps; ++i) {
        x = x * 1664525u + 1013904223u;
        const unsigned long long wd = ((unsigned long long)(x * 2246822519u) << 32) | (x ^ 0x9e3779b9u);
        const unsigned sh = (x >> 7) & 31u;
        const unsigned a = x >> 3, b = x * 7u, c = 0x100u;
        unsigned out, lo, f1, f2;
        if (N == 0) CASE("");
        else if (N == 1) CASE("v_sub_u32 %[f1], %[a], %[b]\n\t");
        else if (N == 2) CASE("v_sub_u32 %[f1], %[a], %[b]\n\tv_and_b32 %[f2], 0xffff, %[a]\n\t");
        else if (N == 3) CASE("v_sub_u32 %[f1], %[a], %[b]\n\tv_and_b32 %[f2], 0xffff, %[a]\n\tv_sub_u32 %[f1], %[f1], %[b]\n\t");
        else CASE("s_nop 7\n\t");
        (void)f1; (void)f2;
        if (out != a + b + c) ++nbad;
        if (lo != (unsigned)(wd >> sh)) ++nbadlo;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nbadlo) atomicAdd(bad_lo, nbadlo);
}

int main() {
    unsigned* bad;
    (void)hipMalloc(&bad, 8);
    const int grids[3] = {8, 300, 8192};
    for (int g = 0; g < 3; ++g)
        for (int n = 0; n < 5; ++n) {
            (void)hipMemset(bad, 0, 8);
            const int reps = g == 2 ? 20000 : 400000;
            if (n == 0) hipLaunchKernelGGL(k<0>, dim3(grids[g]), dim3(64), 0, 0, bad, bad + 1, reps);
            if (n == 1) hipLaunchKernelGGL(k<1>, dim3(grids[g]), dim3(64), 0, 0, bad, bad + 1, reps);
            if (n == 2) hipLaunchKernelGGL(k<2>, dim3(grids[g]), dim3(64), 0, 0, bad, bad + 1, reps);
            if (n == 3) hipLaunchKernelGGL(k<3>, dim3(grids[g]), dim3(64), 0, 0, bad, bad + 1, reps);
            if (n == 4) hipLaunchKernelGGL(k<4>, dim3(grids[g]), dim3(64), 0, 0, bad, bad + 1, reps);
            unsigned h[2] = {0, 0};
            (void)hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
            printf("grid %5d, %s VALU between v_lshrrev_b64 v[40:41] and the 32-bit write of v41: later value lost %u times, low half wrong %u times, of %llu\n",
                   grids[g], n == 4 ? "s_nop 7 instead of" : (n == 0 ? "0" : n == 1 ? "1" : n == 2 ? "2" : "3"), h[0], h[1], (unsigned long long)grids[g] * 64 * reps);
        }
    return 0;
}
