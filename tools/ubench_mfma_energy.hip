// Micro-benchmark (development aid, not product): does the RATE of v_mfma_f32_32x32x16_{bf16,f16} on a power-capped
// MI355X depend on the operand DATA?  A chip-filling grid issues independent MFMAs on register operands for ~50 ms per
// pattern; the package sits at its power cap, so a pattern that switches fewer multiplier inputs runs at a higher clock.
//   pattern 0: random bits in all 16 K slots        1: K slots 8..15 zero (lanes 32..63 hold them)      2: all zero
//   pattern 3: bf16x3 split operands as the scoring kernel builds them (A: a0 a1 a0 a2 a0 a1 | ... ; B: x0 x0 x1 ...)
//   pattern 4: small-mantissa values (top 3 mantissa bits only)
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_energy.hip -o tools/ubench_mfma_energy.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define MF(acc, a, b)                                                                                                   \
    acc = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), zero, 0, 0, 0) \
              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), zero, 0, 0, 0)

// ORDER 0: A alternates on every MFMA, B changes every second one (the scoring kernel's order: cr, dt per tile pair)
// ORDER 1: A stationary over four different B, then the other A over the same four B
// ORDER 2: the same two operands on every MFMA (no operand switching at all)
template <bool F16, int ORDER>
__global__ __launch_bounds__(256) void k_mfma(const uint4* __restrict__ ops, float* __restrict__ sink, int reps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    uint4 a0 = ops[(t * 8 + 0) & 0xFFFF], a1 = ops[(t * 8 + 1) & 0xFFFF];
    uint4 b0 = ops[(t * 8 + 2) & 0xFFFF], b1 = ops[(t * 8 + 3) & 0xFFFF];
    uint4 b2 = ops[(t * 8 + 4) & 0xFFFF], b3 = ops[(t * 8 + 5) & 0xFFFF];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    const f32x16 zero = {};
    for (int r = 0; r < reps; ++r) {
        if (ORDER == 0) {
            MF(c0, a0, b0); MF(c1, a1, b0); MF(c2, a0, b1); MF(c3, a1, b1);
            MF(c4, a0, b2); MF(c5, a1, b2); MF(c6, a0, b3); MF(c7, a1, b3);
        } else if (ORDER == 1) {
            MF(c0, a0, b0); MF(c2, a0, b1); MF(c4, a0, b2); MF(c6, a0, b3);
            MF(c1, a1, b0); MF(c3, a1, b1); MF(c5, a1, b2); MF(c7, a1, b3);
        } else {
            MF(c0, a0, b0); MF(c1, a0, b0); MF(c2, a0, b0); MF(c3, a0, b0);
            MF(c4, a0, b0); MF(c5, a0, b0); MF(c6, a0, b0); MF(c7, a0, b0);
        }
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0.x) : "v"((unsigned)(r & 0)));  // 1 VALU per 8 MFMA: not hoistable
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i] + c6[i] + c7[i];
    if (s == 12345.678f) sink[t] = s;
}

template <bool F16>
static void launch(int order, int blocks, const uint4* d, float* sink, int reps) {
    if (order == 0) hipLaunchKernelGGL((k_mfma<F16, 0>), dim3(blocks), dim3(256), 0, 0, d, sink, reps);
    if (order == 1) hipLaunchKernelGGL((k_mfma<F16, 1>), dim3(blocks), dim3(256), 0, 0, d, sink, reps);
    if (order == 2) hipLaunchKernelGGL((k_mfma<F16, 2>), dim3(blocks), dim3(256), 0, 0, d, sink, reps);
}

static u16 bf16_rn(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
static float bf16_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

int main() {
    const int NOPS = 65536;
    std::vector<uint4> h(NOPS);
    uint4* d;
    float* sink;
    hipMalloc(&d, NOPS * sizeof(uint4));
    hipMalloc(&sink, 1 << 22);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int f16 = 0; f16 < 2; ++f16)
        for (int pat = 0; pat < 5; ++pat) {
            srand(1);
            for (int i = 0; i < NOPS; ++i) {
                u16 k[8];
                const int t = i / 8, lane = t & 63, upper = lane >> 5;
                for (int j = 0; j < 8; ++j) {
                    // a finite random value of the format: sign + exponent near 1 + random mantissa
                    u16 v = f16 ? (u16)(((rand() & 1) << 15) | ((12 + rand() % 6) << 10) | (rand() & 0x3FF))
                                : (u16)(((rand() & 1) << 15) | ((120 + rand() % 12) << 7) | (rand() & 0x7F));
                    if (pat == 1 && upper) v = 0;
                    if (pat == 2) v = 0;
                    if (pat == 4) v &= f16 ? 0xFF80 : 0xFFF0;
                    k[j] = v;
                }
                if (pat == 3 && !f16) {  // the scoring kernel's operand shapes (random fp32 values, bf16x3 parts)
                    float x = (float)rand() / RAND_MAX * 2000.f - 1000.f, y = (float)rand() / RAND_MAX * 2000.f - 1000.f;
                    u16 p[2][3];
                    float vals[2] = {x, y};
                    for (int c = 0; c < 2; ++c) {
                        float r = vals[c];
                        for (int q = 0; q < 3; ++q) { p[c][q] = bf16_rn(r); r -= bf16_f(p[c][q]); }
                    }
                    if (!upper) { k[0] = p[0][0]; k[1] = p[0][1]; k[2] = p[0][0]; k[3] = p[0][2]; k[4] = p[0][0]; k[5] = p[0][1]; k[6] = p[1][0]; k[7] = p[1][1]; }
                    else { k[0] = p[1][0]; k[1] = p[1][2]; k[2] = p[1][0]; k[3] = p[1][1]; k[4] = p[0][0]; k[5] = p[0][1]; k[6] = p[0][2]; k[7] = 0; }
                }
                h[i] = make_uint4(k[0] | (k[1] << 16), k[2] | (k[3] << 16), k[4] | (k[5] << 16), k[6] | (k[7] << 16));
            }
            hipMemcpy(d, h.data(), NOPS * sizeof(uint4), hipMemcpyHostToDevice);
            const int reps = 20000;  // ~25 ms per launch
            for (int order = 0; order < 3; ++order)
                for (int w = 0; w < 2; ++w) {  // first pass warms clocks / power state, second is reported
                    hipEventRecord(e0);
                    if (f16) launch<true>(order, blocks, d, sink, reps);
                    else launch<false>(order, blocks, d, sink, reps);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double flop = (double)blocks * 4 /*waves*/ * reps * 8 /*mfma*/ * 32768.0;
                    if (w) printf("%s pattern %d order %d: %8.2f ms  %7.1f TFLOP/s\n", f16 ? "f16 " : "bf16", pat, order, ms, flop / ms / 1e9);
                }
        }
    return 0;
}
