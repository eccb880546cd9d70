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


// ---- the two candidate scoring loops, both with VGPR-only VALU operands and LDS-broadcast "other side" ----
// (a) lane owns 8 hypotheses, pixel records broadcast from LDS: 6 VALU per test (the shipped loop)
__global__ __launch_bounds__(256) void k_clamp6(float* out, int npix, int reps) {
    __shared__ float4 s_a[1024];
    __shared__ float2 s_b[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) {
        s_a[i] = make_float4(0.5f + i, -0.25f * i, 3.f - i, 0.125f * i);
        s_b[i] = make_float2(0.75f - i, 10.f + i);
    }
    __syncthreads();
    float hx[8], hy[8], cnt[8];
    for (int j = 0; j < 8; ++j) { hx[j] = threadIdx.x + j; hy[j] = threadIdx.x * 0.5f - j; cnt[j] = 0.f; }
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < npix; i += 4) {
            float4 qa[4]; float2 qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { qa[u] = s_a[i + u]; qb[u] = s_b[i + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float cr = fmaf(hx[j], qa[u].x, fmaf(hy[j], qa[u].y, qa[u].z));
                    const float t = qb[u].y - fabsf(cr);
                    cnt[j] += __builtin_amdgcn_fmed3f(fmaf(hx[j], qa[u].w, fmaf(hy[j], qb[u].x, t)), 0.f, 1.f);
                }
        }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += cnt[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (b) lane owns PPL pixels (6 VGPRs each), hypotheses broadcast from LDS: 4 fma + 1 v_cmp (|cr| as a source
//     modifier, wave mask to an SGPR pair) per test on the VALU, s_bcnt1 + s_add on the scalar unit
template <int PPL>
__global__ __launch_bounds__(256) void k_ballot5(float* out, int nhyp, int reps, float sd) {
    __shared__ float2 s_h[1024];
    __shared__ int s_cnt[4][1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_h[i] = make_float2(0.5f + i, 3.f - 0.25f * i);
    __syncthreads();
    float A[PPL], B[PPL], C[PPL], D[PPL], E[PPL], F[PPL];
    for (int u = 0; u < PPL; ++u) {
        A[u] = sd * (threadIdx.x + u); B[u] = sd * (threadIdx.x * 0.5f - u); C[u] = sd * (1.f + u + threadIdx.x);
        D[u] = sd * (0.25f * threadIdx.x - u); E[u] = sd * (7.f - u - threadIdx.x); F[u] = sd * (threadIdx.x - 100.f - u);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = 0; r < reps; ++r)
        for (int h = 0; h < nhyp; h += 2) {
            const float4 hv = *reinterpret_cast<const float4*>(&s_h[h]);  // two hypotheses per broadcast read
            int c0 = 0, c1 = 0;
#pragma unroll
            for (int u = 0; u < PPL; ++u) {
                const float cr0 = fmaf(hv.x, A[u], fmaf(hv.y, B[u], C[u]));
                const float d0 = fmaf(hv.x, D[u], fmaf(hv.y, E[u], F[u]));
                c0 += __builtin_popcountll(__builtin_amdgcn_fcmpf(d0, fabsf(cr0), 2 /* FCMP_OGT */));
                const float cr1 = fmaf(hv.z, A[u], fmaf(hv.w, B[u], C[u]));
                const float d1 = fmaf(hv.z, D[u], fmaf(hv.w, E[u], F[u]));
                c1 += __builtin_popcountll(__builtin_amdgcn_fcmpf(d1, fabsf(cr1), 2));
            }
            if (lane == 0) *reinterpret_cast<int2*>(&s_cnt[wave][h]) = make_int2(c0, c1);
        }
    __syncthreads();
    int s = 0;
    for (int i = lane; i < nhyp; i += 64) s += s_cnt[wave][i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}


// (c) issue cost of v_cmp_*_e64 writing an SGPR pair: 4 fma + 1 compare whose mask is discarded (no SALU at all)
template <int MODE>  // 0: v_cmp_e64 -> SGPR pair, 1: v_cmp_e32 -> vcc, 2: no compare (4 fma + 1 v_max)
__global__ __launch_bounds__(256) void k_cmpcost(float* out, int nhyp, int reps, float sd) {
    __shared__ float2 s_h[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_h[i] = make_float2(0.5f + i, 3.f - 0.25f * i);
    __syncthreads();
    float A[8], B[8], C[8], D[8], E[8], F[8];
    for (int u = 0; u < 8; ++u) {
        A[u] = sd * (threadIdx.x + u); B[u] = sd * (threadIdx.x * 0.5f - u); C[u] = sd * (1.f + u + threadIdx.x);
        D[u] = sd * (0.25f * threadIdx.x - u); E[u] = sd * (7.f - u - threadIdx.x); F[u] = sd * (threadIdx.x - 100.f - u);
    }
    float sink = 0.f;
    for (int r = 0; r < reps; ++r)
        for (int h = 0; h < nhyp; ++h) {
            const float2 hv = s_h[h];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float cr0 = fmaf(hv.x, A[u], fmaf(hv.y, B[u], C[u]));
                const float d0 = fmaf(hv.x, D[u], fmaf(hv.y, E[u], F[u]));
                if (MODE == 0) { unsigned long long m; asm volatile("v_cmp_gt_f32_e64 %0, %1, |%2|" : "=s"(m) : "v"(d0), "v"(cr0)); }
                else if (MODE == 1) { asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(d0), "v"(cr0) : "vcc"); }
                else { float t; asm volatile("v_max_f32 %0, %1, |%2|" : "=v"(t) : "v"(d0), "v"(cr0)); }
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = sink;
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
    // candidate scoring loops: same number of pair tests each (256 threads x 8 x 1024 x reps per workgroup)
    for (int wpc = 2; wpc <= 8; wpc *= 2) {
        dim3 g(cus * wpc), b(256);
        const int reps = 8;
        const double tests = (double)g.x * 256 * 8 * 1024 * reps;
        float t;
        t = time_kernel([&] { hipLaunchKernelGGL(k_clamp6, g, b, 0, 0, out, 1024, reps); }, 5);
        printf("waves/SIMD %d  clamp6 lane-owns-hyps (8/lane)   : %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_ballot5<8>, g, b, 0, 0, out, 1024, reps, 1.37f); }, 5);
        printf("waves/SIMD %d  ballot5 lane-owns-pixels (8/lane): %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_ballot5<4>, g, b, 0, 0, out, 1024, 2 * reps, 1.37f); }, 5);
        printf("waves/SIMD %d  ballot5 lane-owns-pixels (4/lane): %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_ballot5<16>, g, b, 0, 0, out, 1024, reps / 2, 1.37f); }, 5);
        printf("waves/SIMD %d  ballot5 lane-owns-pixels (16/lane): %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_cmpcost<0>, g, b, 0, 0, out, 1024, reps, 1.37f); }, 5);
        printf("waves/SIMD %d  4 fma + v_cmp_e64->sgpr (discarded): %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_cmpcost<1>, g, b, 0, 0, out, 1024, reps, 1.37f); }, 5);
        printf("waves/SIMD %d  4 fma + v_cmp_e32->vcc  (discarded): %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(k_cmpcost<2>, g, b, 0, 0, out, 1024, reps, 1.37f); }, 5);
        printf("waves/SIMD %d  4 fma + v_max (5 plain VALU)       : %8.3f ms  %7.2f Tpairs/s\n", wpc, t, tests / t / 1e9);
    }
    return 0;
}
