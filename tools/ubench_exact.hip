// Micro-benchmark + instruction-semantics check (development aid, not product) for the EXACT scoring epilogue:
//   t   = clamp(dt - |cr|)            one v_fma_mixlo/hi_f16 per test: a saturating ramp, stored as f16 (two tests per VGPR)
//   x   = top bytes of four t's       one v_perm_b32 per four tests: 0x00 (no vote), 0x3C (vote), anything else = "near the threshold"
//   S1 += sum x,  S2 += sum x^2       two v_dot4_u32_u8 per four tests
// All x in {0, 0x3C}  <=>  0x3C * S1 == S2  (x (0x3C - x) > 0 for every other byte; no wrap: S2 <= 256 * 0x3C^2).
// = 1.75 VALU operations per test against the 1.5 of the plain clamp + add3 epilogue (vote8i, copied here as the baseline).
// Also measures the matrix pipe's accumulation error against exact arithmetic (the constant the rounding band needs).
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_exact.hip -o tools/ubench_exact.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__host__ __device__ inline u16 bf16_rn(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (u16)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__host__ __device__ inline float bf16_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ---------------------------------------------------------------------------------------------------------------
// 1. instruction semantics: every lane gets 8 (d, c) pairs, returns w0..w3, x0, x1, S1, S2
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void vote8x(unsigned& s1, unsigned& s2, unsigned sel, unsigned ones, float d0, float c0, float d1,
                                       float c1, float d2, float c2, float d3, float c3, float d4, float c4, float d5,
                                       float c5, float d6, float c6, float d7, float c7) {
    unsigned w0, w1, w2, w3;
    asm volatile(
        "v_fma_mixlo_f16 %2, %8, 1.0, -|%9| clamp\n"
        "v_fma_mixlo_f16 %3, %12, 1.0, -|%13| clamp\n"
        "v_fma_mixlo_f16 %4, %16, 1.0, -|%17| clamp\n"
        "v_fma_mixlo_f16 %5, %20, 1.0, -|%21| clamp\n"
        "v_fma_mixhi_f16 %2, %10, 1.0, -|%11| clamp\n"
        "v_fma_mixhi_f16 %3, %14, 1.0, -|%15| clamp\n"
        "v_fma_mixhi_f16 %4, %18, 1.0, -|%19| clamp\n"
        "v_fma_mixhi_f16 %5, %22, 1.0, -|%23| clamp\n"
        "v_perm_b32 %2, %3, %2, %6\n"
        "v_perm_b32 %4, %5, %4, %6\n"
        "v_dot4_u32_u8 %0, %2, %7, %0\n"
        "v_dot4_u32_u8 %1, %2, %2, %1\n"
        "v_dot4_u32_u8 %0, %4, %7, %0\n"
        "v_dot4_u32_u8 %1, %4, %4, %1\n"
        : "+v"(s1), "+v"(s2), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
        : "v"(sel), "v"(ones), "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4),
          "v"(d5), "v"(c5), "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}
// the same with 16-bit moments (no byte packing): 2.0 operations per test
__device__ __forceinline__ void vote8y(unsigned& s1, unsigned& s2, unsigned ones, float d0, float c0, float d1, float c1,
                                       float d2, float c2, float d3, float c3, float d4, float c4, float d5, float c5,
                                       float d6, float c6, float d7, float c7) {
    unsigned w0, w1, w2, w3;
    asm volatile(
        "v_fma_mixlo_f16 %2, %7, 1.0, -|%8| clamp\n"
        "v_fma_mixlo_f16 %3, %11, 1.0, -|%12| clamp\n"
        "v_fma_mixlo_f16 %4, %15, 1.0, -|%16| clamp\n"
        "v_fma_mixlo_f16 %5, %19, 1.0, -|%20| clamp\n"
        "v_fma_mixhi_f16 %2, %9, 1.0, -|%10| clamp\n"
        "v_fma_mixhi_f16 %3, %13, 1.0, -|%14| clamp\n"
        "v_fma_mixhi_f16 %4, %17, 1.0, -|%18| clamp\n"
        "v_fma_mixhi_f16 %5, %21, 1.0, -|%22| clamp\n"
        "v_dot2_u32_u16 %0, %2, %6, %0\n"
        "v_dot2_u32_u16 %1, %2, %2, %1\n"
        "v_dot2_u32_u16 %0, %3, %6, %0\n"
        "v_dot2_u32_u16 %1, %3, %3, %1\n"
        "v_dot2_u32_u16 %0, %4, %6, %0\n"
        "v_dot2_u32_u16 %1, %4, %4, %1\n"
        "v_dot2_u32_u16 %0, %5, %6, %0\n"
        "v_dot2_u32_u16 %1, %5, %5, %1\n"
        : "+v"(s1), "+v"(s2), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
        : "v"(ones), "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4), "v"(d5),
          "v"(c5), "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}
__device__ __forceinline__ void vote8i(unsigned& acc, float d0, float c0, float d1, float c1, float d2, float c2, float d3,
                                       float c3, float d4, float c4, float d5, float c5, float d6, float c6, float d7,
                                       float c7) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_sub_f32_e64 %1, %5, |%6| clamp\n"
        "v_sub_f32_e64 %2, %7, |%8| clamp\n"
        "v_sub_f32_e64 %3, %9, |%10| clamp\n"
        "v_sub_f32_e64 %4, %11, |%12| clamp\n"
        "v_add3_u32 %0, %1, %2, %0\n"
        "v_sub_f32_e64 %1, %13, |%14| clamp\n"
        "v_sub_f32_e64 %2, %15, |%16| clamp\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_sub_f32_e64 %3, %17, |%18| clamp\n"
        "v_sub_f32_e64 %4, %19, |%20| clamp\n"
        "v_add3_u32 %0, %1, %2, %0\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4), "v"(d5), "v"(c5),
          "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}


// ---- a/b formulation (product: vote8ab): a = dt - cr, b = dt + cr; vote = clamp(min(a, b)); band = min over |a|, |b|
#define AB_ARGS float a0, float b0, float a1, float b1, float a2, float b2, float a3, float b3, float a4, float b4, float a5, \
                float b5, float a6, float b6, float a7, float b7
#define AB_IN "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), \
              "v"(a6), "v"(b6), "v"(a7), "v"(b7)
// EPI 3: the product's epilogue, one min3 chain
__device__ __forceinline__ void ab_min3_1chain(unsigned& acc, float& dm, AB_ARGS) {
    float t0, t1, t2;
    asm volatile(
        "v_min_f32_e64 %2, %5, %6 clamp\n"
        "v_min_f32_e64 %3, %7, %8 clamp\n"
        "v_min3_f32 %1, %1, |%5|, |%6|\n"
        "v_min_f32_e64 %4, %9, %10 clamp\n"
        "v_min3_f32 %1, %1, |%7|, |%8|\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        "v_min_f32_e64 %2, %11, %12 clamp\n"
        "v_min3_f32 %1, %1, |%9|, |%10|\n"
        "v_min_f32_e64 %3, %13, %14 clamp\n"
        "v_min3_f32 %1, %1, |%11|, |%12|\n"
        "v_add3_u32 %0, %4, %2, %0\n"
        "v_min_f32_e64 %4, %15, %16 clamp\n"
        "v_min3_f32 %1, %1, |%13|, |%14|\n"
        "v_min_f32_e64 %2, %17, %18 clamp\n"
        "v_min3_f32 %1, %1, |%15|, |%16|\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_min_f32_e64 %3, %19, %20 clamp\n"
        "v_min3_f32 %1, %1, |%17|, |%18|\n"
        "v_min3_f32 %1, %1, |%19|, |%20|\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        : "+v"(acc), "+v"(dm), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : AB_IN);
}
// EPI 4: four independent min3 chains
__device__ __forceinline__ void ab_min3_4chain(unsigned& acc, float& d0, float& d1, float& d2, float& d3, AB_ARGS) {
    float t0, t1, t2;
    asm volatile(
        "v_min_f32_e64 %5, %8, %9 clamp\n"
        "v_min_f32_e64 %6, %10, %11 clamp\n"
        "v_min3_f32 %1, %1, |%8|, |%9|\n"
        "v_min_f32_e64 %7, %12, %13 clamp\n"
        "v_min3_f32 %2, %2, |%10|, |%11|\n"
        "v_add3_u32 %0, %5, %6, %0\n"
        "v_min_f32_e64 %5, %14, %15 clamp\n"
        "v_min3_f32 %3, %3, |%12|, |%13|\n"
        "v_min_f32_e64 %6, %16, %17 clamp\n"
        "v_min3_f32 %4, %4, |%14|, |%15|\n"
        "v_add3_u32 %0, %7, %5, %0\n"
        "v_min_f32_e64 %7, %18, %19 clamp\n"
        "v_min3_f32 %1, %1, |%16|, |%17|\n"
        "v_min_f32_e64 %5, %20, %21 clamp\n"
        "v_min3_f32 %2, %2, |%18|, |%19|\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        "v_min_f32_e64 %6, %22, %23 clamp\n"
        "v_min3_f32 %3, %3, |%20|, |%21|\n"
        "v_min3_f32 %4, %4, |%22|, |%23|\n"
        "v_add3_u32 %0, %5, %6, %0\n"
        : "+v"(acc), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : AB_IN);
}
// EPI 5: two-input minima instead of min3 (two chains): 3.5 operations per test
__device__ __forceinline__ void ab_min2(unsigned& acc, float& d0, float& d1, AB_ARGS) {
    float t0, t1, t2;
    asm volatile(
        "v_min_f32_e64 %3, %6, %7 clamp\n"
        "v_min_f32_e64 %4, %8, %9 clamp\n"
        "v_min_f32_e64 %1, %1, |%6|\n"
        "v_min_f32_e64 %2, %2, |%7|\n"
        "v_min_f32_e64 %1, %1, |%8|\n"
        "v_min_f32_e64 %2, %2, |%9|\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_min_f32_e64 %3, %10, %11 clamp\n"
        "v_min_f32_e64 %4, %12, %13 clamp\n"
        "v_min_f32_e64 %1, %1, |%10|\n"
        "v_min_f32_e64 %2, %2, |%11|\n"
        "v_min_f32_e64 %1, %1, |%12|\n"
        "v_min_f32_e64 %2, %2, |%13|\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_min_f32_e64 %3, %14, %15 clamp\n"
        "v_min_f32_e64 %4, %16, %17 clamp\n"
        "v_min_f32_e64 %1, %1, |%14|\n"
        "v_min_f32_e64 %2, %2, |%15|\n"
        "v_min_f32_e64 %1, %1, |%16|\n"
        "v_min_f32_e64 %2, %2, |%17|\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_min_f32_e64 %3, %18, %19 clamp\n"
        "v_min_f32_e64 %4, %20, %21 clamp\n"
        "v_min_f32_e64 %1, %1, |%18|\n"
        "v_min_f32_e64 %2, %2, |%19|\n"
        "v_min_f32_e64 %1, %1, |%20|\n"
        "v_min_f32_e64 %2, %2, |%21|\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        : "+v"(acc), "+v"(d0), "+v"(d1), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : AB_IN);
}
// EPI 6: u = clamp(min(|a|, |b|)) (1.0 unless a test is near the band), two of them per v_add3_u32: 3 operations per test
__device__ __forceinline__ void ab_uclamp(unsigned& acc, unsigned& uacc, AB_ARGS) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_min_f32_e64 %2, %6, %7 clamp\n"
        "v_min_f32_e64 %3, %8, %9 clamp\n"
        "v_min_f32_e64 %4, |%6|, |%7| clamp\n"
        "v_min_f32_e64 %5, |%8|, |%9| clamp\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        "v_min_f32_e64 %2, %10, %11 clamp\n"
        "v_min_f32_e64 %3, %12, %13 clamp\n"
        "v_add3_u32 %1, %4, %5, %1\n"
        "v_min_f32_e64 %4, |%10|, |%11| clamp\n"
        "v_min_f32_e64 %5, |%12|, |%13| clamp\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        "v_min_f32_e64 %2, %14, %15 clamp\n"
        "v_min_f32_e64 %3, %16, %17 clamp\n"
        "v_add3_u32 %1, %4, %5, %1\n"
        "v_min_f32_e64 %4, |%14|, |%15| clamp\n"
        "v_min_f32_e64 %5, |%16|, |%17| clamp\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        "v_min_f32_e64 %2, %18, %19 clamp\n"
        "v_min_f32_e64 %3, %20, %21 clamp\n"
        "v_add3_u32 %1, %4, %5, %1\n"
        "v_min_f32_e64 %4, |%18|, |%19| clamp\n"
        "v_min_f32_e64 %5, |%20|, |%21| clamp\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        "v_add3_u32 %1, %4, %5, %1\n"
        : "+v"(acc), "+v"(uacc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : AB_IN);
}
// EPI 7: the vote alone in the a/b form (no band): 1.5 operations per test, as EPI 0
__device__ __forceinline__ void ab_vote_only(unsigned& acc, AB_ARGS) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_min_f32_e64 %1, %5, %6 clamp\n"
        "v_min_f32_e64 %2, %7, %8 clamp\n"
        "v_min_f32_e64 %3, %9, %10 clamp\n"
        "v_min_f32_e64 %4, %11, %12 clamp\n"
        "v_add3_u32 %0, %1, %2, %0\n"
        "v_min_f32_e64 %1, %13, %14 clamp\n"
        "v_min_f32_e64 %2, %15, %16 clamp\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        "v_min_f32_e64 %3, %17, %18 clamp\n"
        "v_min_f32_e64 %4, %19, %20 clamp\n"
        "v_add3_u32 %0, %1, %2, %0\n"
        "v_add3_u32 %0, %3, %4, %0\n"
        : "+v"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : AB_IN);
}
// EPI 8: min3 WITHOUT the |.| modifiers (is it the modifiers or the instruction?)
__device__ __forceinline__ void ab_min3_noabs(unsigned& acc, float& d0, float& d1, float& d2, float& d3, AB_ARGS) {
    float t0, t1, t2;
    asm volatile(
        "v_min_f32_e64 %5, %8, %9 clamp\n"
        "v_min_f32_e64 %6, %10, %11 clamp\n"
        "v_min3_f32 %1, %1, %8, %9\n"
        "v_min_f32_e64 %7, %12, %13 clamp\n"
        "v_min3_f32 %2, %2, %10, %11\n"
        "v_add3_u32 %0, %5, %6, %0\n"
        "v_min_f32_e64 %5, %14, %15 clamp\n"
        "v_min3_f32 %3, %3, %12, %13\n"
        "v_min_f32_e64 %6, %16, %17 clamp\n"
        "v_min3_f32 %4, %4, %14, %15\n"
        "v_add3_u32 %0, %7, %5, %0\n"
        "v_min_f32_e64 %7, %18, %19 clamp\n"
        "v_min3_f32 %1, %1, %16, %17\n"
        "v_min_f32_e64 %5, %20, %21 clamp\n"
        "v_min3_f32 %2, %2, %18, %19\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        "v_min_f32_e64 %6, %22, %23 clamp\n"
        "v_min3_f32 %3, %3, %20, %21\n"
        "v_min3_f32 %4, %4, %22, %23\n"
        "v_add3_u32 %0, %5, %6, %0\n"
        : "+v"(acc), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : AB_IN);
}

// EPI 9: x = min(a, b, 1) (1.0 = vote, <= -1 = clean non-vote, in between = inside the band); band = min |x| (two x per
// v_min3); votes = v_cvt_pknorm_u16_f32 packs two clamp(x) as 0xFFFF / 0 halves, two such words per v_add3_u32: 2.25 op/test
__device__ __forceinline__ void ab_x(unsigned& acc, float& dm, AB_ARGS) {
    float x0, x1, x2, x3;
    unsigned w0, w1;
    asm volatile(
        "v_min3_f32 %2, %8, %9, 1.0\n"
        "v_min3_f32 %3, %10, %11, 1.0\n"
        "v_min3_f32 %4, %12, %13, 1.0\n"
        "v_min3_f32 %5, %14, %15, 1.0\n"
        "v_min3_f32 %1, %1, |%2|, |%3|\n"
        "v_cvt_pknorm_u16_f32 %6, %2, %3\n"
        "v_min3_f32 %2, %16, %17, 1.0\n"
        "v_min3_f32 %3, %18, %19, 1.0\n"
        "v_min3_f32 %1, %1, |%4|, |%5|\n"
        "v_cvt_pknorm_u16_f32 %7, %4, %5\n"
        "v_min3_f32 %4, %20, %21, 1.0\n"
        "v_min3_f32 %5, %22, %23, 1.0\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        "v_min3_f32 %1, %1, |%2|, |%3|\n"
        "v_cvt_pknorm_u16_f32 %6, %2, %3\n"
        "v_min3_f32 %1, %1, |%4|, |%5|\n"
        "v_cvt_pknorm_u16_f32 %7, %4, %5\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        : "+v"(acc), "+v"(dm), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(w0), "=&v"(w1)
        : AB_IN);
}
__global__ void k_semantics_x(const float* __restrict__ a, const float* __restrict__ b, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float av[8], bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { av[q] = a[i * 8 + q]; bv[q] = b[i * 8 + q]; }
    unsigned acc = 0;
    float dm = 3e38f;
    ab_x(acc, dm, av[0], bv[0], av[1], bv[1], av[2], bv[2], av[3], bv[3], av[4], bv[4], av[5], bv[5], av[6], bv[6], av[7], bv[7]);
    out[i * 2] = acc;
    out[i * 2 + 1] = __builtin_bit_cast(unsigned, dm);
}
// EPI 10: y = clamp(min(a, b, 1)) in [0, 1] (the band shifted to (0, 1) by the operands: y = 0 a clean non-vote, y = 1 a clean vote);
// two y per v_cvt_pk_fp8_f32 (four tests per dword: 0x00 / C = fp8(1.0) / something in between), votes S += bytes and band test
// T += |byte - C/2| with one v_sad_u8 each per dword: a cell is clean iff T == tests * C/2 (every other byte is closer to C/2),
// and then votes = S / C.  2.0 operations per test.
__device__ __forceinline__ void ab_fp8(unsigned& S, unsigned& T, unsigned mid, AB_ARGS) {
    float y0, y1, y2, y3;
    unsigned w0, w1;
    asm volatile(
        "v_min3_f32 %2, %8, %9, 1.0 clamp\n"
        "v_min3_f32 %3, %10, %11, 1.0 clamp\n"
        "v_min3_f32 %4, %12, %13, 1.0 clamp\n"
        "v_min3_f32 %5, %14, %15, 1.0 clamp\n"
        "v_cvt_pk_fp8_f32 %6, %2, %3\n"
        "v_cvt_pk_fp8_f32 %6, %4, %5 op_sel:[0,0,1]\n"
        "v_min3_f32 %2, %16, %17, 1.0 clamp\n"
        "v_min3_f32 %3, %18, %19, 1.0 clamp\n"
        "v_min3_f32 %4, %20, %21, 1.0 clamp\n"
        "v_min3_f32 %5, %22, %23, 1.0 clamp\n"
        "v_sad_u8 %0, %6, 0, %0\n"
        "v_cvt_pk_fp8_f32 %7, %2, %3\n"
        "v_sad_u8 %1, %6, %24, %1\n"
        "v_cvt_pk_fp8_f32 %7, %4, %5 op_sel:[0,0,1]\n"
        "v_sad_u8 %0, %7, 0, %0\n"
        "v_sad_u8 %1, %7, %24, %1\n"
        : "+v"(S), "+v"(T), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(w0), "=&v"(w1)
        : AB_IN, "s"(mid));
}
// EPI 11: the same with v_cvt_pk_bf8_f32 (e5m2)
__device__ __forceinline__ void ab_bf8(unsigned& S, unsigned& T, unsigned mid, AB_ARGS) {
    float y0, y1, y2, y3;
    unsigned w0, w1;
    asm volatile(
        "v_min3_f32 %2, %8, %9, 1.0 clamp\n"
        "v_min3_f32 %3, %10, %11, 1.0 clamp\n"
        "v_min3_f32 %4, %12, %13, 1.0 clamp\n"
        "v_min3_f32 %5, %14, %15, 1.0 clamp\n"
        "v_cvt_pk_bf8_f32 %6, %2, %3\n"
        "v_cvt_pk_bf8_f32 %6, %4, %5 op_sel:[0,0,1]\n"
        "v_min3_f32 %2, %16, %17, 1.0 clamp\n"
        "v_min3_f32 %3, %18, %19, 1.0 clamp\n"
        "v_min3_f32 %4, %20, %21, 1.0 clamp\n"
        "v_min3_f32 %5, %22, %23, 1.0 clamp\n"
        "v_sad_u8 %0, %6, 0, %0\n"
        "v_cvt_pk_bf8_f32 %7, %2, %3\n"
        "v_sad_u8 %1, %6, %24, %1\n"
        "v_cvt_pk_bf8_f32 %7, %4, %5 op_sel:[0,0,1]\n"
        "v_sad_u8 %0, %7, 0, %0\n"
        "v_sad_u8 %1, %7, %24, %1\n"
        : "+v"(S), "+v"(T), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(w0), "=&v"(w1)
        : AB_IN, "s"(mid));
}
// EPI 12: the conversions alone (1.5 operations: is v_cvt_pk_fp8_f32 full rate?)     EPI 13: min3-clamp + the SADs on the raw
// float bits (1.5 operations: is v_sad_u8 full rate?)
__device__ __forceinline__ void ab_fp8_only(unsigned& S, AB_ARGS) {
    float y0, y1, y2, y3;
    unsigned w0, w1;
    asm volatile(
        "v_min3_f32 %1, %7, %8, 1.0 clamp\n"
        "v_min3_f32 %2, %9, %10, 1.0 clamp\n"
        "v_min3_f32 %3, %11, %12, 1.0 clamp\n"
        "v_min3_f32 %4, %13, %14, 1.0 clamp\n"
        "v_cvt_pk_fp8_f32 %5, %1, %2\n"
        "v_cvt_pk_fp8_f32 %5, %3, %4 op_sel:[0,0,1]\n"
        "v_min3_f32 %1, %15, %16, 1.0 clamp\n"
        "v_min3_f32 %2, %17, %18, 1.0 clamp\n"
        "v_min3_f32 %3, %19, %20, 1.0 clamp\n"
        "v_min3_f32 %4, %21, %22, 1.0 clamp\n"
        "v_cvt_pk_fp8_f32 %6, %1, %2\n"
        "v_cvt_pk_fp8_f32 %6, %3, %4 op_sel:[0,0,1]\n"
        "v_xor_b32 %0, %0, %5\n"
        "v_xor_b32 %0, %0, %6\n"
        : "+v"(S), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(w0), "=&v"(w1)
        : AB_IN);
}
__device__ __forceinline__ void ab_sad_only(unsigned& S, unsigned& T, unsigned mid, AB_ARGS) {
    float y0, y1, y2, y3;
    asm volatile(
        "v_min3_f32 %2, %6, %7, 1.0 clamp\n"
        "v_min3_f32 %3, %8, %9, 1.0 clamp\n"
        "v_min3_f32 %4, %10, %11, 1.0 clamp\n"
        "v_min3_f32 %5, %12, %13, 1.0 clamp\n"
        "v_sad_u8 %0, %2, 0, %0\n"
        "v_sad_u8 %1, %3, %22, %1\n"
        "v_min3_f32 %2, %14, %15, 1.0 clamp\n"
        "v_min3_f32 %3, %16, %17, 1.0 clamp\n"
        "v_sad_u8 %0, %4, 0, %0\n"
        "v_sad_u8 %1, %5, %22, %1\n"
        "v_min3_f32 %4, %18, %19, 1.0 clamp\n"
        "v_min3_f32 %5, %20, %21, 1.0 clamp\n"
        : "+v"(S), "+v"(T), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
        : AB_IN, "s"(mid));
}
__global__ void k_semantics_fp8(const float* __restrict__ a, const float* __restrict__ b, unsigned* __restrict__ out, int bf8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float av[8], bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { av[q] = a[i * 8 + q]; bv[q] = b[i * 8 + q]; }
    unsigned S = 0, T = 0;
    const unsigned one = bf8 ? (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(1.0f, 1.0f, 0, false) & 0xFFu
                             : (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(1.0f, 1.0f, 0, false) & 0xFFu;
    const unsigned mid = __builtin_amdgcn_readfirstlane((one >> 1) * 0x01010101u);
    if (bf8) ab_bf8(S, T, mid, av[0], bv[0], av[1], bv[1], av[2], bv[2], av[3], bv[3], av[4], bv[4], av[5], bv[5], av[6], bv[6], av[7], bv[7]);
    else ab_fp8(S, T, mid, av[0], bv[0], av[1], bv[1], av[2], bv[2], av[3], bv[3], av[4], bv[4], av[5], bv[5], av[6], bv[6], av[7], bv[7]);
    out[i * 4] = S;
    out[i * 4 + 1] = T;
    out[i * 4 + 2] = one;
    // the codes of the first four tests, for the edge report
    unsigned w = 0;
    float y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = fminf(fmaxf(fminf(fminf(av[q], bv[q]), 1.0f), 0.f), 1.f);
    if (bf8) { w = __builtin_amdgcn_cvt_pk_bf8_f32(y[0], y[1], w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(y[2], y[3], w, true); }
    else { w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true); }
    out[i * 4 + 3] = w;
}
#define AB_HALF(o) d[o + 0], cr[o + 0], d[o + 1], cr[o + 1], d[o + 2], cr[o + 2], d[o + 3], cr[o + 3], d[o + 4], cr[o + 4], \
                   d[o + 5], cr[o + 5], d[o + 6], cr[o + 6], d[o + 7], cr[o + 7]

// the same eight tests with the second operand taken r registers further (semantically meaningless: does the VGPR BANK relation
// of the two MFMA results a v_min3 reads matter?  d and cr are 16-register blocks, so d[q] and cr[q] share a bank when the blocks
// are a multiple of four registers apart)
#define AB_ROT(o, r) d[o + 0], cr[(o + 0 + r) & 15], d[o + 1], cr[(o + 1 + r) & 15], d[o + 2], cr[(o + 2 + r) & 15], d[o + 3], cr[(o + 3 + r) & 15], \
                     d[o + 4], cr[(o + 4 + r) & 15], d[o + 5], cr[(o + 5 + r) & 15], d[o + 6], cr[(o + 6 + r) & 15], d[o + 7], cr[(o + 7 + r) & 15]

__global__ void k_semantics(const float* __restrict__ d, const float* __restrict__ c, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float dv[8], cv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { dv[q] = d[i * 8 + q]; cv[q] = c[i * 8 + q]; }
    unsigned s1 = 0, s2 = 0, t1 = 0, t2 = 0;
    vote8x(s1, s2, 0x07050301u, 0x01010101u, dv[0], cv[0], dv[1], cv[1], dv[2], cv[2], dv[3], cv[3], dv[4], cv[4], dv[5], cv[5],
           dv[6], cv[6], dv[7], cv[7]);
    vote8y(t1, t2, 0x00010001u, dv[0], cv[0], dv[1], cv[1], dv[2], cv[2], dv[3], cv[3], dv[4], cv[4], dv[5], cv[5], dv[6], cv[6],
           dv[7], cv[7]);
    out[i * 4 + 0] = s1;
    out[i * 4 + 1] = s2;
    out[i * 4 + 2] = t1;
    out[i * 4 + 3] = t2;
}

// ---------------------------------------------------------------------------------------------------------------
// 2. matrix-pipe accumulation error: D = A (32 x 16) * B (16 x 32), bf16 operands, against float64
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_mfma_err(const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const size_t blk = blockIdx.x;
    bf16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = __builtin_bit_cast(__bf16, A[blk * 512 + col * 16 + half * 8 + k]);  // row = col (A rows indexed by lane & 31)
        b[k] = __builtin_bit_cast(__bf16, B[blk * 512 + col * 16 + half * 8 + k]);  // column = col
    }
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = (q >> 2) * 8 + half * 4 + (q & 3);
        D[blk * 1024 + row * 32 + col] = r[q];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. throughput beside the MFMAs (the pipeline of the product kernel: PIPE 4 of tools/ubench_mfma.hip)
// ---------------------------------------------------------------------------------------------------------------
constexpr int TILE_BYTES = 2 * 32 * 16 * 2;
template <int MH, int EPI>  // EPI 0: clamp + add3 (1.5 op)   1: mix + perm + dot4 moments (1.75)   2: mix + dot2 moments (2.0)
__global__ __launch_bounds__(256) void k_pipe(const u16* __restrict__ Bsrc, const u16* __restrict__ Asrc,
                                              unsigned* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    for (int i = threadIdx.x; i < ntiles * TILE_BYTES / 2; i += 256) lds[i] = Asrc[i];
    bf16x8 B[MH];
    unsigned s1[MH], s2[MH];
    float f0[MH], f1 = 3e38f, f2 = 3e38f, f3 = 3e38f;
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) B[t][k] = __builtin_bit_cast(__bf16, Bsrc[j * 16 + half * 8 + k]);
        s1[t] = 0u;
        s2[t] = 0u;
        f0[t] = 3e38f;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lbase = reinterpret_cast<const char*>(lds) + (lane & 31) * 32 + half * 16;
    const unsigned sel = 0x07050301u, ones8 = 0x01010101u, ones16 = 0x00010001u;
    const unsigned mid8 = __builtin_amdgcn_readfirstlane(0x1C1C1C1Cu + (unsigned)(reps >> 30));
    for (int r = 0; r < reps; ++r) {
        bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase), Ad = *reinterpret_cast<const bf16x8*>(lbase + 1024);
        f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[0], zero, 0, 0, 0);
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[0], zero, 0, 0, 0);
        for (int tile = 0; tile < ntiles; ++tile) {
            const int nt = tile + 1 < ntiles ? tile + 1 : tile;
            const bf16x8 Ncr = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES);
            const bf16x8 Nd = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES + 1024);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const f32x16 cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (EPI == 0) vote8i(s1[t], d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                else if (EPI == 1) vote8x(s1[t], s2[t], sel, ones8, d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                else if (EPI == 2) vote8y(s1[t], s2[t], ones16, d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                else if (EPI == 3) ab_min3_1chain(s1[t], f0[t], AB_HALF(0));
                else if (EPI == 4) ab_min3_4chain(s1[t], f0[t], f1, f2, f3, AB_HALF(0));
                else if (EPI == 5) ab_min2(s1[t], f0[t], f1, AB_HALF(0));
                else if (EPI == 6) ab_uclamp(s1[t], s2[t], AB_HALF(0));
                else if (EPI == 7) ab_vote_only(s1[t], AB_HALF(0));
                else if (EPI == 8) ab_min3_noabs(s1[t], f0[t], f1, f2, f3, AB_HALF(0));
                else if (EPI == 14) ab_x(s1[t], f0[t], AB_ROT(0, 1));
                else if (EPI == 15) ab_x(s1[t], f0[t], AB_ROT(0, 2));
                else if (EPI == 10) ab_fp8(s1[t], s2[t], mid8, AB_HALF(0));
                else if (EPI == 11) ab_bf8(s1[t], s2[t], mid8, AB_HALF(0));
                else if (EPI == 12) ab_fp8_only(s1[t], AB_HALF(0));
                else if (EPI == 13) ab_sad_only(s1[t], s2[t], mid8, AB_HALF(0));
                else ab_x(s1[t], f0[t], AB_HALF(0));
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (EPI == 0) vote8i(s1[t], d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                else if (EPI == 1) vote8x(s1[t], s2[t], sel, ones8, d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                else if (EPI == 2) vote8y(s1[t], s2[t], ones16, d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                else if (EPI == 3) ab_min3_1chain(s1[t], f0[t], AB_HALF(8));
                else if (EPI == 4) ab_min3_4chain(s1[t], f0[t], f1, f2, f3, AB_HALF(8));
                else if (EPI == 5) ab_min2(s1[t], f0[t], f1, AB_HALF(8));
                else if (EPI == 6) ab_uclamp(s1[t], s2[t], AB_HALF(8));
                else if (EPI == 7) ab_vote_only(s1[t], AB_HALF(8));
                else if (EPI == 8) ab_min3_noabs(s1[t], f0[t], f1, f2, f3, AB_HALF(8));
                else if (EPI == 14) ab_x(s1[t], f0[t], AB_ROT(8, 1));
                else if (EPI == 15) ab_x(s1[t], f0[t], AB_ROT(8, 2));
                else if (EPI == 10) ab_fp8(s1[t], s2[t], mid8, AB_HALF(8));
                else if (EPI == 11) ab_bf8(s1[t], s2[t], mid8, AB_HALF(8));
                else if (EPI == 12) ab_fp8_only(s1[t], AB_HALF(8));
                else if (EPI == 13) ab_sad_only(s1[t], s2[t], mid8, AB_HALF(8));
                else ab_x(s1[t], f0[t], AB_HALF(8));
                __builtin_amdgcn_sched_barrier(0);
                cr = cr2;
                d = d2;
            }
            Acr = Ncr;
            Ad = Nd;
        }
    }
#pragma unroll
    for (int t = 0; t < MH; ++t)
        counts[(((size_t)blockIdx.x * 4 + wave) * MH + t) * 64 + lane] =
            s1[t] ^ s2[t] ^ __builtin_bit_cast(unsigned, f0[t]) ^ __builtin_bit_cast(unsigned, f1 + f2 + f3);
}

// PIPE 2: two accumulator pairs as shipped, but BOTH MFMAs of the next step issued before the votes of this one (instead of one
// before each half); PIPE 3: both issued between the two halves
template <int MH, int ORDER>
__global__ __launch_bounds__(256) void k_pipe2(const u16* __restrict__ Bsrc, const u16* __restrict__ Asrc,
                                               unsigned* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    for (int i = threadIdx.x; i < ntiles * TILE_BYTES / 2; i += 256) lds[i] = Asrc[i];
    bf16x8 B[MH];
    unsigned s1[MH];
    float f0[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) B[t][k] = __builtin_bit_cast(__bf16, Bsrc[j * 16 + half * 8 + k]);
        s1[t] = 0u;
        f0[t] = 3e38f;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lbase = reinterpret_cast<const char*>(lds) + (lane & 31) * 32 + half * 16;
    for (int r = 0; r < reps; ++r) {
        bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase), Ad = *reinterpret_cast<const bf16x8*>(lbase + 1024);
        f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[0], zero, 0, 0, 0);
        f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[0], zero, 0, 0, 0);
        for (int tile = 0; tile < ntiles; ++tile) {
            const int nt = tile + 1 < ntiles ? tile + 1 : tile;
            const bf16x8 Ncr = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES);
            const bf16x8 Nd = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES + 1024);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                f32x16 cr2, d2;
                if (ORDER == 2) {
                    cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ab_x(s1[t], f0[t], AB_HALF(0));
                __builtin_amdgcn_sched_barrier(0);
                if (ORDER == 3) {
                    cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ab_x(s1[t], f0[t], AB_HALF(8));
                __builtin_amdgcn_sched_barrier(0);
                cr = cr2;
                d = d2;
            }
            Acr = Ncr;
            Ad = Nd;
        }
    }
#pragma unroll
    for (int t = 0; t < MH; ++t)
        counts[(((size_t)blockIdx.x * 4 + wave) * MH + t) * 64 + lane] = s1[t] ^ __builtin_bit_cast(unsigned, f0[t]);
}

// PIPE 1: the same work with ONE accumulator pair -- the two MFMAs of a step are issued and their 32 results consumed by the
// same wave right away (the wave waits for the matrix pipe; the SIMD's other waves fill the gap).  32 VGPRs fewer.
template <int MH, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_pipe1(
    const u16* __restrict__ Bsrc, const u16* __restrict__ Asrc, unsigned* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    for (int i = threadIdx.x; i < ntiles * TILE_BYTES / 2; i += 256) lds[i] = Asrc[i];
    bf16x8 B[MH];
    unsigned s1[MH];
    float f0[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < 8; ++k) B[t][k] = __builtin_bit_cast(__bf16, Bsrc[j * 16 + half * 8 + k]);
        s1[t] = 0u;
        f0[t] = 3e38f;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lbase = reinterpret_cast<const char*>(lds) + (lane & 31) * 32 + half * 16;
    for (int r = 0; r < reps; ++r) {
        for (int tile = 0; tile < ntiles; ++tile) {
            const bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase + tile * TILE_BYTES);
            const bf16x8 Ad = *reinterpret_cast<const bf16x8*>(lbase + tile * TILE_BYTES + 1024);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[t], zero, 0, 0, 0);
                const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[t], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 11");  // MFMA result -> inline-asm VALU read: the wait states are ours to insert
                __builtin_amdgcn_sched_barrier(0);
                ab_x(s1[t], f0[t], AB_HALF(0));
                ab_x(s1[t], f0[t], AB_HALF(8));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MH; ++t)
        counts[(((size_t)blockIdx.x * 4 + wave) * MH + t) * 64 + lane] = s1[t] ^ __builtin_bit_cast(unsigned, f0[t]);
}

static float h2f(unsigned short h) { _Float16 x = __builtin_bit_cast(_Float16, h); return (float)x; }

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    // ---- 1. semantics
    {
        const int n = 64 * 64;
        std::vector<float> d(n * 8), c(n * 8);
        for (int i = 0; i < n * 8; ++i) {
            const int kind = rand() % 8;
            float dv, cv;
            if (kind == 0) { dv = rnd() * 1000.f; cv = dv + 5.f + rnd() * 100.f; cv = rand() & 1 ? cv : -cv; }          // far below: 0
            else if (kind == 1) { cv = (rnd() - .5f) * 1000.f; dv = fabsf(cv) + 1.f + rnd() * 50.f; }                  // far above: 1
            else if (kind == 2) { cv = (rnd() - .5f) * 100.f; dv = fabsf(cv) + rnd(); }                                // in the ramp
            else if (kind == 3) { cv = (rnd() - .5f) * 4.f; dv = fabsf(cv) + 0.5f; }                                   // exactly 0.5 (often)
            else if (kind == 4) { cv = 0.f; dv = ldexpf(1.f, -(rand() % 30)); }                                         // powers of two
            else if (kind == 5) { cv = NAN; dv = 1.f; }
            else if (kind == 6) { cv = 3.f; dv = INFINITY; }
            else { cv = (rnd() - .5f) * 2.f; dv = fabsf(cv) + 1.f - ldexpf(1.f, -(rand() % 26)); }                      // just below 1
            d[i] = dv; c[i] = cv;
        }
        float *dd, *dc; unsigned* dout;
        hipMalloc(&dd, n * 32); hipMalloc(&dc, n * 32); hipMalloc(&dout, n * 16);
        hipMemcpy(dd, d.data(), n * 32, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), n * 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_semantics, dim3(n / 64), dim3(64), 0, 0, dd, dc, dout);
        std::vector<unsigned> out(n * 4);
        hipMemcpy(out.data(), dout, n * 16, hipMemcpyDeviceToHost);
        long bad8 = 0, bad16 = 0, nfrac = 0, nvote = 0, detect_miss = 0;
        for (int i = 0; i < n; ++i) {
            unsigned e1 = 0, e2 = 0, f1 = 0, f2 = 0;
            bool anyfrac = false;
            for (int q = 0; q < 8; ++q) {
                float t = d[i * 8 + q] - fabsf(c[i * 8 + q]);
                t = t != t ? 0.f : (t < 0.f ? 0.f : (t > 1.f ? 1.f : t));
                const _Float16 hh = (_Float16)t;  // round to nearest even
                const unsigned short p = __builtin_bit_cast(unsigned short, hh);
                const unsigned b = p >> 8;
                e1 += b; e2 += b * b; f1 += p; f2 += (unsigned)p * p;
                if (b != 0 && b != 0x3C) anyfrac = true;
                nvote += p == 0x3C00;
            }
            nfrac += anyfrac;
            if (out[i * 4] != e1 || out[i * 4 + 1] != e2) { if (++bad8 <= 5) printf("  lane %d: u8 moments %u %u, expected %u %u\n", i, out[i*4], out[i*4+1], e1, e2); }
            if (out[i * 4 + 2] != f1 || out[i * 4 + 3] != f2) { if (++bad16 <= 5) printf("  lane %d: u16 moments %u %u, expected %u %u\n", i, out[i*4+2], out[i*4+3], f1, f2); }
            if (anyfrac && 0x3Cu * out[i * 4] == out[i * 4 + 1]) ++detect_miss;
        }
        printf("semantics: %d lanes x 8 tests: u8 moments wrong in %ld lanes, u16 moments wrong in %ld lanes; %ld lanes hold a fractional byte, %ld of them undetected by 0x3C*S1 != S2; %ld votes\n",
               n, bad8, bad16, nfrac, detect_miss, nvote);
        (void)h2f;
    }

    {   // ---- 1b. semantics of the x = min3(a, b, 1) epilogue: votes decoded from the packed-norm sum, band from min |x|
        const int n = 64 * 64;
        std::vector<float> a(n * 8), b(n * 8);
        for (int i = 0; i < n * 8; ++i) {
            const int kind = rand() % 6;
            float av, bv;
            if (kind == 0) { av = 1.f + rnd() * 1e4f; bv = 1.f + rnd() * 1e4f; }                     // clean vote
            else if (kind == 1) { av = -1.f - rnd() * 1e4f; bv = (rnd() - .5f) * 1e4f; }              // clean non-vote
            else if (kind == 2) { av = 1.f; bv = 1.f + rnd(); }                                        // exactly at the edge: vote, clean
            else if (kind == 3) { av = -1.f; bv = 5.f; }                                               // exactly -1: clean non-vote
            else if (kind == 4) { av = (rnd() - .5f) * 1.99f; bv = 3.f + rnd(); }                       // inside the band
            else { av = -4.f; bv = -4.f; }                                                            // padding row
            if (rand() & 1) { float t = av; av = bv; bv = t; }
            a[i] = av; b[i] = bv;
        }
        if (true) {  // one cell with a hypothesis-free of band tests only, to make sure clean cells exist
            for (int q = 0; q < 8 * 64; ++q) { a[q] = q & 1 ? 2.f : -3.f; b[q] = 7.f; }
        }
        float *da, *db; unsigned* dout;
        hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dout, n * 8);
        hipMemcpy(da, a.data(), n * 32, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), n * 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_semantics_x, dim3(n / 64), dim3(64), 0, 0, da, db, dout);
        std::vector<unsigned> out(n * 2);
        hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
        long clean = 0, wrong_votes = 0, wrong_band = 0;
        for (int i = 0; i < n; ++i) {
            int v_lo = 0, v_hi = 0;
            float dm = 3e38f;
            for (int q = 0; q < 8; ++q) {
                const float x = fminf(fminf(a[i * 8 + q], b[i * 8 + q]), 1.f);
                dm = fminf(dm, fabsf(x));
                if (x >= 1.f) { if (q & 1) ++v_hi; else ++v_lo; }
            }
            const float got_dm = __builtin_bit_cast(float, out[i * 2 + 1]);
            if ((got_dm >= 1.f) != (dm >= 1.f)) ++wrong_band;
            if (dm >= 1.f) {  // clean cell: decode  acc = 0xFFFF v_lo + 65536 * 0xFFFF v_hi  (mod 2^32)
                ++clean;
                const unsigned acc = out[i * 2];
                const unsigned lo = (0u - acc) & 0xFFFFu;                 // = v_lo
                const unsigned hi = (lo - ((acc + lo) >> 16)) & 0xFFFFu;  // acc + v_lo = 65536 (v_lo - v_hi)  (mod 2^32)
                if ((int)lo != v_lo || (int)hi != v_hi) { if (++wrong_votes <= 5) printf("  lane %d: acc %08x decodes to (%u, %u), expected (%d, %d)\n", i, acc, lo, hi, v_lo, v_hi); }
            }
        }
        printf("x-epilogue semantics: %d cells of 8 tests, %ld clean; wrong band flags %ld, wrong vote decodes in clean cells %ld\n", n, clean, wrong_band, wrong_votes);
    }
    // ---- 1c. the fp8 / bf8 + SAD epilogue
    for (int bf8 = 0; bf8 < 2; ++bf8) {
        const int n = 64 * 256;
        std::vector<float> a(n * 8), b(n * 8);
        for (int i = 0; i < n; ++i) {
            const int cellkind = rand() % 4;  // 0: all clean, else: some tests inside (0, 1)
            for (int q = 0; q < 8; ++q) {
                float lo;
                const int kind = cellkind == 0 ? rand() % 2 : rand() % 8;
                if (kind == 0) lo = -rnd() * 50.f;                       // clean non-vote (y = 0)
                else if (kind == 1) lo = 1.f + rnd() * 50.f;             // clean vote (y = 1)
                else if (kind == 2) lo = rnd();                          // anywhere in (0, 1)
                else if (kind == 3) lo = ldexpf(1.f, -(rand() % 20));    // small powers of two
                else if (kind == 4) lo = 1.f - ldexpf(1.f, -(rand() % 20));  // just below 1
                else if (kind == 5) lo = 0.05f + 0.9f * rnd();           // the band proper
                else if (kind == 6) lo = rand() & 1 ? 0.05f : 0.95f;     // its two ends
                else lo = 0.f;
                const float hi = lo + rnd() * 3.f;
                if (rand() & 1) { a[i * 8 + q] = lo; b[i * 8 + q] = hi; } else { a[i * 8 + q] = hi; b[i * 8 + q] = lo; }
            }
        }
        float *da, *db; unsigned* dout;
        hipMalloc(&da, n * 32); hipMalloc(&db, n * 32); hipMalloc(&dout, n * 16);
        hipMemcpy(da, a.data(), n * 32, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), n * 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_semantics_fp8, dim3(n / 64), dim3(64), 0, 0, da, db, dout, bf8);
        std::vector<unsigned> out(n * 4);
        hipMemcpy(out.data(), dout, n * 16, hipMemcpyDeviceToHost);
        const unsigned one = out[2];
        long clean_true = 0, flagged = 0, missed_band = 0, wrong_votes = 0, edge_to_0 = 0, edge_to_1 = 0, nonmono = 0;
        float max_to_0 = 0.f, min_to_1 = 1.f;
        for (int i = 0; i < n; ++i) {
            int votes = 0; bool all_clean = true, has_band = false;
            for (int q = 0; q < 8; ++q) {
                const float m = fminf(a[i * 8 + q], b[i * 8 + q]);
                const float y = m < 0.f ? 0.f : (m > 1.f ? 1.f : m);
                if (y == 1.f) ++votes;
                if (y != 0.f && y != 1.f) all_clean = false;
                if (y >= 0.05f && y <= 0.95f) has_band = true;
                if (q < 4) {
                    const unsigned code = (out[i * 4 + 3] >> (8 * q)) & 0xFFu;
                    if (code > one) ++nonmono;
                    if (y > 0.f && y < 1.f && code == 0) { ++edge_to_0; if (y > max_to_0) max_to_0 = y; }
                    if (y > 0.f && y < 1.f && code == one) { ++edge_to_1; if (y < min_to_1) min_to_1 = y; }
                }
            }
            const unsigned S = out[i * 4], T = out[i * 4 + 1];
            const bool looks_clean = T == 8 * (one >> 1);
            if (all_clean) ++clean_true;
            if (!looks_clean) ++flagged;
            if (has_band && looks_clean) ++missed_band;
            if (all_clean && (!looks_clean || S != one * (unsigned)votes)) ++wrong_votes;
        }
        printf("%s + sad semantics: code(1.0) = 0x%02x; %d cells of 8 tests, %ld truly clean, %ld flagged; cells with a test in [0.05, 0.95] NOT flagged: %ld; "
               "clean cells with wrong flag / votes: %ld; codes above code(1.0): %ld; y in (0,1) coded 0: %ld (largest %.3g), coded 1.0: %ld (smallest %.6f)\n",
               bf8 ? "bf8" : "fp8", one, n, clean_true, flagged, missed_band, wrong_votes, nonmono, edge_to_0, max_to_0, edge_to_1, min_to_1);
    }
    // ---- 2. MFMA accumulation error
    {
        const int nb = 4096;
        std::vector<u16> A(nb * 512), B(nb * 512);
        for (size_t i = 0; i < A.size(); ++i) {
            // products of very different magnitudes and signs, as the bf16x3 parts of one fp32 product have
            const float ma = ldexpf(rnd() + 1.f, -(rand() % 18)) * (rand() & 1 ? 1.f : -1.f);
            const float mb = ldexpf(rnd() + 1.f, (rand() % 12)) * (rand() & 1 ? 1.f : -1.f);
            A[i] = bf16_rn(ma);
            B[i] = bf16_rn(mb);
        }
        u16 *dA, *dB; float* dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, (size_t)nb * 1024 * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma_err, dim3(nb), dim3(64), 0, 0, dA, dB, dD);
        std::vector<float> D((size_t)nb * 1024);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double worst_abs = 0, worst_res = 0;  // error / (2^-24 * sum |terms|)   and   error / (2^-24 * |result|)
        long wrong_layout = 0;
        for (int blk = 0; blk < nb; ++blk)
            for (int r = 0; r < 32; ++r)
                for (int cc = 0; cc < 32; ++cc) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < 16; ++k) {
                        const double p = (double)bf16_f(A[(size_t)blk * 512 + r * 16 + k]) * bf16_f(B[(size_t)blk * 512 + cc * 16 + k]);
                        s += p; sa += fabs(p);
                    }
                    const double e = fabs((double)D[(size_t)blk * 1024 + r * 32 + cc] - s);
                    if (e > 1e-3 * sa) ++wrong_layout;
                    const double u = ldexp(1.0, -24);
                    if (e / (u * sa) > worst_abs) worst_abs = e / (u * sa);
                    if (fabs(s) > 0 && e / (u * fabs(s)) > worst_res) worst_res = e / (u * fabs(s));
                }
        printf("mfma accumulation (32x32x16 bf16, %d tiles): max |error| = %.3f * 2^-24 * sum|terms|   (%.1f * 2^-24 * |result| worst, cancellation included); layout mismatches %ld\n",
               nb, worst_abs, worst_res, wrong_layout);
    }
    // ---- 3. throughput
    {
        constexpr int MH = 8;
        const int ntiles = 8;
        std::vector<u16> Bs(4 * MH * 32 * 16), As(ntiles * TILE_BYTES / 2);
        for (auto& x : Bs) x = bf16_rn((rnd() - .5f) * 64.f);
        for (auto& x : As) x = bf16_rn((rnd() - .5f) * 4.f);
        u16 *dB, *dA; unsigned* dc;
        hipMalloc(&dB, Bs.size() * 2); hipMalloc(&dA, As.size() * 2); hipMalloc(&dc, (size_t)cus * 8 * 4 * MH * 64 * 4);
        hipMemcpy(dB, Bs.data(), Bs.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dA, As.data(), As.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int wpc = 3; wpc <= 3; ++wpc)
            for (int epi = 0; epi < 16; ++epi) {
                const dim3 g(cus * wpc), b(256);
                const int reps = 64;
                auto launch = [&] {
#define LAUNCH(E) case E: hipLaunchKernelGGL((k_pipe<MH, E>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps); break;
                    switch (epi) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15) }
#undef LAUNCH
                };
                launch();
                hipDeviceSynchronize();
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    for (int i = 0; i < 5; ++i) launch();
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms / 5 < best ? ms / 5 : best;
                }
                const double tests = (double)g.x * 4 * MH * 32 * ntiles * 32 * reps;
                printf("workgroups/CU %d  MH=%d  %s: %8.3f ms  %7.2f T tests/s\n", wpc, MH,
                       epi == 0 ? "clamp + add3              (1.5  op/test)" : epi == 1 ? "mix + perm + dot4 moments (1.75 op/test)" :
                       epi == 2 ? "mix + dot2 moments        (2.0  op/test)" : epi == 3 ? "a/b: min-clamp, min3 x1 chain, add3 (2.5)" :
                       epi == 4 ? "a/b: min-clamp, min3 x4 chains, add3 (2.5)" : epi == 5 ? "a/b: min-clamp, 2 x min |.|, add3  (3.5)" :
                       epi == 6 ? "a/b: min-clamp, min|.|-clamp, 2 add3 (3.0)" : epi == 7 ? "a/b: min-clamp + add3 only        (1.5)" :
                       epi == 8 ? "a/b: min3 without |.|, 4 chains    (2.5)" :
                       epi == 10 ? "a/b: y = clamp min3, cvt_pk_fp8, 2 x sad_u8 (2.0)" :
                       epi == 11 ? "a/b: y = clamp min3, cvt_pk_bf8, 2 x sad_u8 (2.0)" :
                       epi == 12 ? "a/b: y = clamp min3, cvt_pk_fp8 only       (1.5+)" :
                       epi == 13 ? "a/b: y = clamp min3, sad_u8 on raw bits    (1.5)" :
                       epi == 14 ? "x-epilogue, second operand one register on (bank +1)" :
                       epi == 15 ? "x-epilogue, second operand two registers on (bank +2)" :
                                  "a/b: x = min3(a, b, 1), min3 |x|, cvt_pknorm, add3 (2.25)",
                       best, tests / best / 1e9);
            }
    }
    // ---- 4. one accumulator pair instead of two (PIPE 1), 3 and 4 workgroups per CU
    {
        constexpr int MH = 8;
        const int ntiles = 8;
        std::vector<u16> Bs(4 * MH * 32 * 16), As(ntiles * TILE_BYTES / 2);
        for (auto& x : Bs) x = bf16_rn((rnd() - .5f) * 64.f);
        for (auto& x : As) x = bf16_rn((rnd() - .5f) * 4.f);
        u16 *dB, *dA; unsigned* dc;
        hipMalloc(&dB, Bs.size() * 2); hipMalloc(&dA, As.size() * 2); hipMalloc(&dc, (size_t)cus * 8 * 4 * MH * 64 * 4);
        hipMemcpy(dB, Bs.data(), Bs.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dA, As.data(), As.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int var = 0; var < 5; ++var) {
            const int wpc = var == 2 ? 4 : 3;
            const dim3 g(cus * wpc), b(256);
            const int reps = 64;
            auto launch = [&] {
                if (var == 0) hipLaunchKernelGGL((k_pipe<MH, 9>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps);
                else if (var == 1) hipLaunchKernelGGL((k_pipe1<MH, 3>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps);
                else if (var == 2) hipLaunchKernelGGL((k_pipe1<MH, 4>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps);
                else if (var == 3) hipLaunchKernelGGL((k_pipe2<MH, 2>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps);
                else hipLaunchKernelGGL((k_pipe2<MH, 3>), g, b, ntiles * TILE_BYTES, 0, dB, dA, dc, ntiles, reps);
            };
            int occ = 0;
            if (var == 0) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pipe<MH, 9>, 256, ntiles * TILE_BYTES);
            else if (var == 1) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pipe1<MH, 3>, 256, ntiles * TILE_BYTES);
            else if (var == 2) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pipe1<MH, 4>, 256, ntiles * TILE_BYTES);
            else if (var == 3) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pipe2<MH, 2>, 256, ntiles * TILE_BYTES);
            else hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pipe2<MH, 3>, 256, ntiles * TILE_BYTES);
            launch();
            hipDeviceSynchronize();
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < 5; ++i) launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms / 5 < best ? ms / 5 : best;
            }
            const double tests = (double)g.x * 4 * MH * 32 * ntiles * 32 * reps;
            printf("%s, %d workgroups/CU launched (occupancy %d): %8.3f ms  %7.2f T tests/s\n",
                   var == 0 ? "two accumulator pairs (shipped pipeline), x-epilogue" : var <= 2 ? "ONE accumulator pair, x-epilogue" :
                   var == 3 ? "two pairs, BOTH next MFMAs before this step's votes" : "two pairs, both next MFMAs between the halves", wpc, occ, best,
                   tests / best / 1e9);
        }
    }
    return 0;
}
