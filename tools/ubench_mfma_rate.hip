// Micro-benchmark (development aid): issue rate of MFMA shapes on gfx950 -- is a legacy K=8 instruction cheaper per
// 32x32 output tile than the K=16 one?   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_rate.hip -o tools/ubench_mfma_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int n) {
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
    }
    const float s = (float)threadIdx.x * 1e-3f;
    f16x4 a4 = {(_Float16)s, (_Float16)1.f, (_Float16)2.f, (_Float16)3.f}, b4 = a4;
    f16x8 a8 = {(_Float16)s, 1, 2, 3, 4, 5, 6, 7}, b8 = a8;
    bf16x8 c8 = {(__bf16)s, 1, 2, 3, 4, 5, 6, 7}, d8 = c8;
    s16x4 e4 = {(short)threadIdx.x, 1, 2, 3}, g4 = e4;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c8, d8, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
            if (KIND == 3) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(e4, g4, acc[i], 0, 0, 0);
            if (KIND == 4) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4[i], 0, 0, 0);
            if (KIND == 5) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc4[i], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 16; ++j) r += acc[i][j];
        for (int j = 0; j < 4; ++j) r += acc4[i][j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4096 * 4);
    const char* names[6] = {"v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16 (legacy)",
                            "v_mfma_f32_32x32x8_bf16_1k (legacy)", "v_mfma_f32_16x16x16_f16 (legacy)", "v_mfma_f32_16x16x32_f16"};
    const int outs[6] = {1024, 1024, 1024, 1024, 256, 256};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int kind = 0; kind < 6; ++kind) {
            const int n = 20000, grid = 256 * wps;
            auto launch = [&] {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, n); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, n); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, n); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, n); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, n); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, d, n); break;
                }
            };
            launch();
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfma_per_simd = (double)n * 4 * wps;  // each wave issues n*4; wps waves per SIMD
            const double ns_each = ms * 1e6 / mfma_per_simd;
            printf("%d waves/SIMD  %-40s %7.2f ns per MFMA per SIMD  -> %6.1f output elements per ns per SIMD (%5.1f cycles at 2.0 GHz)\n",
                   wps, names[kind], ns_each, outs[kind] / ns_each, ns_each * 2.0);
        }
    return 0;
}
