// Micro-benchmark (development aid, not product): issue rates of the fp32 VALU instructions the scoring kernel is
// built from, on gfx950.  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define ITER 4096

// 8 independent accumulators per lane, plain v_fma_f32
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// same with an SGPR operand (constant bus)
__global__ __launch_bounds__(256) void k_fma_s(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(
            "v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
            "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f,
            x6 = x0 + 6.f, x7 = x0 + 7.f;
    float2v va = {a, a}, vb = {b, b};
    for (int i = 0; i < ITER; ++i) {
        asm volatile(
            "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
            "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
            "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
// the scoring loop's 9-op pair body, 4 hypotheses per lane, scalar record operands
__global__ __launch_bounds__(256) void k_pair9(float* out, float cx, float cy, float mx, float my) {
    float hx[4], hy[4];
    int cnt[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) { hx[j] = threadIdx.x + j; hy[j] = threadIdx.x * 0.5f - j; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dx, dy, dot, l2, q;
            asm volatile(
                "v_subrev_f32 %1, %6, %8\n v_subrev_f32 %2, %7, %9\n v_mul_f32 %3, %10, %1\n v_fmac_f32 %3, %11, %2\n"
                "v_mul_f32 %4, %1, %1\n v_fmac_f32 %4, %2, %2\n v_mul_f32 %5, %3, |%3|\n"
                "v_cmp_gt_f32 vcc, %5, %4\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
                : "+v"(cnt[j]), "=&v"(dx), "=&v"(dy), "=&v"(dot), "=&v"(l2), "=&v"(q)
                : "s"(cx), "s"(cy), "v"(hx[j]), "v"(hy[j]), "s"(mx), "s"(my) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}
// packed variant: 2 hypotheses per packed op for the 6 arithmetic ops, then unpacked mul|abs|, cmp, addc
__global__ __launch_bounds__(256) void k_pair_pk(float* out, float cx, float cy, float mx, float my) {
    float2v hx[2], hy[2];
    int cnt[4] = {0, 0, 0, 0};
    for (int j = 0; j < 2; ++j) { hx[j] = (float2v){threadIdx.x + 2.f * j, threadIdx.x + 2.f * j + 1}; hy[j] = hx[j] * 0.5f; }
    float2v c_x = {cx, cx}, c_y = {cy, cy}, m_x = {mx, mx}, m_y = {my, my};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float2v dx, dy, dot, l2;
            asm volatile(
                "v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_mul_f32 %2, %0, %8\n v_pk_fma_f32 %2, %1, %9, %2\n"
                "v_pk_mul_f32 %3, %0, %0\n v_pk_fma_f32 %3, %1, %1, %3\n"
                : "=&v"(dx), "=&v"(dy), "=&v"(dot), "=&v"(l2)
                : "v"(hx[j]), "v"(hy[j]), "v"(c_x), "v"(c_y), "v"(m_x), "v"(m_y));
            float q0 = dot.x * __builtin_fabsf(dot.x), q1 = dot.y * __builtin_fabsf(dot.y);
            float l0 = l2.x, l1 = l2.y;
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt[2 * j]) : "v"(q0), "v"(l0) : "vcc");
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt[2 * j + 1]) : "v"(q1), "v"(l1) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

template <typename F>
float time_kernel(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clock_khz = 0;
    hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, clock attr %d kHz\n", cus, clock_khz);
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 32);
    for (int wpc = 1; wpc <= 8; wpc *= 2) {  // workgroups (4 waves) per CU => waves per SIMD
        dim3 g(cus * wpc), b(256);
        double lanes = (double)g.x * 256;
        float t;
        t = time_kernel([&] { hipLaunchKernelGGL(k_fma, g, b, 0, 0, out, 1.0001f, 0.5f); }, 5);
        printf("waves/SIMD %d  v_fma_f32      : %8.3f ms  %7.1f TFLOP/s  %6.1f Tinstr-lanes/s\n", wpc, t, lanes * ITER * 8 * 2 / t / 1e9, lanes * ITER * 8 / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_fma_s, g, b, 0, 0, out, 1.0001f, 0.5f); }, 5);
        printf("waves/SIMD %d  v_fmac_f32(sgpr): %8.3f ms  %7.1f TFLOP/s\n", wpc, t, lanes * ITER * 8 * 2 / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_pkfma, g, b, 0, 0, out, 1.0001f, 0.5f); }, 5);
        printf("waves/SIMD %d  v_pk_fma_f32   : %8.3f ms  %7.1f TFLOP/s\n", wpc, t, lanes * ITER * 8 * 4 / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_pair9, g, b, 0, 0, out, 3.f, 4.f, 0.6f, 0.8f); }, 5);
        printf("waves/SIMD %d  pair9 (9 ops)  : %8.3f ms  %7.2f Tpairs/s  %6.1f Tinstr-lanes/s\n", wpc, t, lanes * ITER * 4 / t / 1e9, lanes * ITER * 36 / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_pair_pk, g, b, 0, 0, out, 3.f, 4.f, 0.6f, 0.8f); }, 5);
        printf("waves/SIMD %d  pair_pk (12/2) : %8.3f ms  %7.2f Tpairs/s\n", wpc, t, lanes * ITER * 4 / t / 1e9);
    }
    return 0;
}
