// Micro-benchmark + layout check (development aid, not product): the bf16x3 matrix-pipe form of the vote.
//   cr_ij = hx_j*a_i + hy_j*b_i + c_i      d_ij = hx_j*e_i + hy_j*f_i + g_i      vote_ij = d_ij > |cr_ij|
// Every fp32 operand is split into three bf16 parts (round-to-nearest); a product keeps the six part pairs whose
// weights are >= 2^-16 relative, so a 3-term fp32 dot product becomes ONE v_mfma_f32_32x32x16_bf16 (K = 6+6+3+1):
//   A row i   : [a0 a1 a0 a2 a0 a1 | b0 b1 b0 b2 b0 b1 | c0 c1 c2 0]      (pixel side, from LDS)
//   B column j: [x0 x0 x1 x0 x2 x1 | y0 y0 y1 y0 y2 y1 | 1  1  1  0]      (hypothesis side, in registers)
// Lane l of the result holds column j = l&31 (its hypothesis) and 16 rows: 2 VALU ops per test (v_cmp + v_addc).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__host__ __device__ inline u16 bf16_rn(float x) {  // round-to-nearest-even to bf16, returned as raw bits
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (u16)(u >> 16);  // inf / nan as they are
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__host__ __device__ inline float bf16_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__host__ __device__ inline void split3(float x, u16& p0, u16& p1, u16& p2) {
    p0 = bf16_rn(x);
    const float r1 = x - bf16_f(p0);
    p1 = bf16_rn(r1);
    const float r2 = r1 - bf16_f(p1);
    p2 = bf16_rn(r2);
}

constexpr int TILE_BYTES = 2 * 32 * 16 * 2;  // A_cr | A_d, 32 rows x 16 bf16 each

// pixc: [npix][6] = a b c e f g.  One workgroup stages `ntiles` tiles; thread = pixel.
__device__ inline void stage_pixels(const float* pixc, int npix, u16* lds) {
    for (int i = threadIdx.x; i < npix; i += blockDim.x) {
        u16 p[6][3];
#pragma unroll
        for (int c = 0; c < 6; ++c) split3(pixc[i * 6 + c], p[c][0], p[c][1], p[c][2]);
        u16* t = lds + (i >> 5) * (TILE_BYTES / 2) + (i & 31) * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // m = 0: (a b c)   m = 1: (e f g)
            u16* r = t + m * 32 * 16;
            const u16(*q)[3] = p + 3 * m;
            r[0] = q[0][0]; r[1] = q[0][1]; r[2] = q[0][0]; r[3] = q[0][2]; r[4] = q[0][0]; r[5] = q[0][1];
            r[6] = q[1][0]; r[7] = q[1][1]; r[8] = q[1][0]; r[9] = q[1][2]; r[10] = q[1][0]; r[11] = q[1][1];
            r[12] = q[2][0]; r[13] = q[2][1]; r[14] = q[2][2]; r[15] = 0;
        }
    }
}

__device__ inline bf16x8 make_b(float hx, float hy, int half) {
    u16 x[3], y[3];
    split3(hx, x[0], x[1], x[2]);
    split3(hy, y[0], y[1], y[2]);
    const u16 one = 0x3f80;
    u16 k[8];
    if (half == 0) { k[0] = x[0]; k[1] = x[0]; k[2] = x[1]; k[3] = x[0]; k[4] = x[2]; k[5] = x[1]; k[6] = y[0]; k[7] = y[0]; }
    else { k[0] = y[1]; k[1] = y[0]; k[2] = y[2]; k[3] = y[1]; k[4] = one; k[5] = one; k[6] = one; k[7] = 0; }
    bf16x8 b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(__bf16, k[i]);
    return b;
}

template <int MH>
__global__ __launch_bounds__(256) void k_mfma_vote(const float* __restrict__ hyp, const float* __restrict__ pixc,
                                                   int* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    stage_pixels(pixc + (size_t)blockIdx.y * ntiles * 32 * 6, ntiles * 32, lds);
    bf16x8 B[MH];
    int cnt[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
        B[t] = make_b(hyp[j * 2], hyp[j * 2 + 1], half);
        cnt[t] = 0;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (int tile = 0; tile < ntiles; ++tile) {
            const char* tp = reinterpret_cast<const char*>(lds) + tile * TILE_BYTES + (lane & 31) * 32 + half * 16;
            const bf16x8 Acr = *reinterpret_cast<const bf16x8*>(tp);
            const bf16x8 Ad = *reinterpret_cast<const bf16x8*>(tp + 32 * 32);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[t], zero, 0, 0, 0);
                const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[t], zero, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) cnt[t] += d[q] > __builtin_fabsf(cr[q]) ? 1 : 0;
            }
        }
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int c = cnt[t] + __shfl_xor(cnt[t], 32, 64);  // the two half-waves hold different rows of the column
        if (half == 0) counts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 * MH * 32 + (wave * MH + t) * 32 + lane] = c;
    }
}


// ---- hand-scheduled epilogue: 8 tests per block, producer->consumer distance >= 3 instructions (no s_nop needed
//      for the "VALU writes SGPR -> VALU reads it" hazard of gfx940+), 2 VALU per test
__device__ __forceinline__ void vote8(int& cnt, float d0, float c0, float d1, float c1, float d2, float c2, float d3,
                                      float c3, float d4, float c4, float d5, float c5, float d6, float c6, float d7,
                                      float c7) {
    unsigned long long m0, m1, m2, m3;
    int x, y;
    asm volatile(
        "v_cmp_gt_f32_e64 %1, %7, |%8|\n"
        "v_cmp_gt_f32_e64 %2, %9, |%10|\n"
        "v_cmp_gt_f32_e64 %3, %11, |%12|\n"
        "v_cmp_gt_f32_e64 %4, %13, |%14|\n"
        "v_cndmask_b32_e64 %5, 0, 1, %1\n"
        "v_cmp_gt_f32_e64 %1, %15, |%16|\n"
        "v_addc_co_u32_e64 %0, %2, %0, %5, %2\n"
        "v_cndmask_b32_e64 %6, 0, 1, %3\n"
        "v_cmp_gt_f32_e64 %2, %17, |%18|\n"
        "v_addc_co_u32_e64 %0, %4, %0, %6, %4\n"
        "v_cmp_gt_f32_e64 %3, %19, |%20|\n"
        "v_cmp_gt_f32_e64 %4, %21, |%22|\n"
        "v_cndmask_b32_e64 %5, 0, 1, %1\n"
        "v_cndmask_b32_e64 %6, 0, 1, %3\n"
        "v_addc_co_u32_e64 %0, %2, %0, %5, %2\n"
        "v_addc_co_u32_e64 %0, %4, %0, %6, %4\n"
        : "+v"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&v"(x), "=&v"(y)
        : "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4), "v"(d5), "v"(c5),
          "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}
// plain-VALU epilogue: s = clamp(d - |c|) (0/1 when the operands carry the 2^90 record scaling), cnt += s
__device__ __forceinline__ void vote8f(float& cnt, float d0, float c0, float d1, float c1, float d2, float c2, float d3,
                                       float c3, float d4, float c4, float d5, float c5, float d6, float c6, float d7,
                                       float c7) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_sub_f32_e64 %1, %5, |%6| clamp\n"
        "v_sub_f32_e64 %2, %7, |%8| clamp\n"
        "v_sub_f32_e64 %3, %9, |%10| clamp\n"
        "v_sub_f32_e64 %4, %11, |%12| clamp\n"
        "v_add_f32_e32 %0, %0, %1\n"
        "v_sub_f32_e64 %1, %13, |%14| clamp\n"
        "v_add_f32_e32 %0, %0, %2\n"
        "v_sub_f32_e64 %2, %15, |%16| clamp\n"
        "v_add_f32_e32 %0, %0, %3\n"
        "v_sub_f32_e64 %3, %17, |%18| clamp\n"
        "v_add_f32_e32 %0, %0, %4\n"
        "v_sub_f32_e64 %4, %19, |%20| clamp\n"
        "v_add_f32_e32 %0, %0, %1\n"
        "v_add_f32_e32 %0, %0, %2\n"
        "v_add_f32_e32 %0, %0, %3\n"
        "v_add_f32_e32 %0, %0, %4\n"
        : "+v"(cnt), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4), "v"(d5), "v"(c5),
          "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}
// 1.5-op epilogue: t = clamp(d - |c|) is the bit pattern 0x3F800000 or 0; v_add3_u32 sums two of them per
// instruction in a 32-bit integer that wraps: acc = n * 0x3F800000 mod 2^32 = ((127 n) mod 512) << 23, and 127 is
// invertible mod 512 (127 * 383 = 95 * 512 + 1), so n = (((acc >> 23) * 383) & 511) as long as n < 512
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
__device__ __forceinline__ void vote16(int& cnt, const f32x16& d, const f32x16& c) {
    vote8(cnt, d[0], c[0], d[1], c[1], d[2], c[2], d[3], c[3], d[4], c[4], d[5], c[5], d[6], c[6], d[7], c[7]);
    vote8(cnt, d[8], c[8], d[9], c[9], d[10], c[10], d[11], c[11], d[12], c[12], d[13], c[13], d[14], c[14], d[15], c[15]);
}

template <int MH, int PIPE>
__global__ __launch_bounds__(256) void k_mfma_vote2(const float* __restrict__ hyp, const float* __restrict__ pixc,
                                                    int* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    stage_pixels(pixc + (size_t)blockIdx.y * ntiles * 32 * 6, ntiles * 32, lds);
    bf16x8 B[MH];
    int cnt[MH];
    float fcnt[MH];
    unsigned ucnt[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
        B[t] = make_b(hyp[j * 2], hyp[j * 2 + 1], half);
        cnt[t] = 0;
        fcnt[t] = 0.f;
        ucnt[t] = 0u;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lbase = reinterpret_cast<const char*>(lds) + (lane & 31) * 32 + half * 16;
    for (int r = 0; r < reps; ++r) {
        if (PIPE) {  // flat software pipeline over (pixel tile, hypothesis tile): the MFMAs of step i+1 are issued, then
                     // the votes of step i are counted while they run
            bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase), Ad = *reinterpret_cast<const bf16x8*>(lbase + 1024);
            f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[0], zero, 0, 0, 0);
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[0], zero, 0, 0, 0);
            for (int tile = 0; tile < ntiles; ++tile) {
                const int nt = tile + 1 < ntiles ? tile + 1 : tile;  // (the last prefetch is a harmless repeat)
                const bf16x8 Ncr = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES);
                const bf16x8 Nd = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES + 1024);
#pragma unroll
                for (int t = 0; t < MH; ++t) {
                    f32x16 cr2, d2;
                    if (PIPE == 4) {  // as PIPE == 2 with the 1.5-op wrapped-integer epilogue
                        cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8i(ucnt[t], d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                        __builtin_amdgcn_sched_barrier(0);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8i(ucnt[t], d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (PIPE == 3) {  // as PIPE == 2 with the plain-VALU float epilogue
                        cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8f(fcnt[t], d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                        __builtin_amdgcn_sched_barrier(0);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8f(fcnt[t], d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (PIPE == 2) {  // half of the votes in the shadow of each MFMA
                        cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8(cnt[t], d[0], cr[0], d[1], cr[1], d[2], cr[2], d[3], cr[3], d[4], cr[4], d[5], cr[5], d[6], cr[6], d[7], cr[7]);
                        __builtin_amdgcn_sched_barrier(0);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], zero, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        vote8(cnt[t], d[8], cr[8], d[9], cr[9], d[10], cr[10], d[11], cr[11], d[12], cr[12], d[13], cr[13], d[14], cr[14], d[15], cr[15]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                    if (t + 1 < MH) {
                        cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[t + 1], zero, 0, 0, 0);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[t + 1], zero, 0, 0, 0);
                    } else {
                        cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ncr, B[0], zero, 0, 0, 0);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Nd, B[0], zero, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    vote16(cnt[t], d, cr);
                    __builtin_amdgcn_sched_barrier(0);
                    }
                    cr = cr2;
                    d = d2;
                }
                Acr = Ncr;
                Ad = Nd;
            }
        } else {
            for (int tile = 0; tile < ntiles; ++tile) {
                const bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase + tile * TILE_BYTES);
                const bf16x8 Ad = *reinterpret_cast<const bf16x8*>(lbase + tile * TILE_BYTES + 1024);
#pragma unroll
                for (int t = 0; t < MH; ++t) {
                    const f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[t], zero, 0, 0, 0);
                    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[t], zero, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 15\n s_nop 3");  // XDL write -> VALU read wait states (inline asm is opaque to hipcc)
                    vote16(cnt[t], d, cr);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int ci = PIPE == 4 ? (int)(((ucnt[t] >> 23) * 383u) & 511u) : PIPE == 3 ? (int)fcnt[t] : cnt[t];
        const int c = ci + __shfl_xor(ci, 32, 64);
        if (half == 0) counts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 * MH * 32 + (wave * MH + t) * 32 + lane] = c;
    }
}

// ---- overlap diagnostics: ROLE 0 = the two MFMAs per step only (accumulating, results kept live), ROLE 1 = the 1.5-op
// epilogue only (on fixed registers), ROLE 2 = 512-thread workgroup, waves 0-3 MFMA-only and waves 4-7 VALU-only (two
// waves per SIMD, one of each kind), ROLE 3 = both in every wave (same as PIPE 4, accumulating MFMAs)
template <int MH, int ROLE>
__global__ __launch_bounds__(ROLE == 2 ? 512 : 256) void k_diag(const float* __restrict__ hyp, const float* __restrict__ pixc,
                                                                 int* __restrict__ counts, int ntiles, int reps) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, half = lane >> 5;
    const int kind = ROLE == 2 ? (threadIdx.x >> 8) : ROLE;  // 0 mfma, 1 valu, 3 both
    stage_pixels(pixc + (size_t)blockIdx.y * ntiles * 32 * 6, ntiles * 32, lds);
    bf16x8 B[MH];
    unsigned ucnt[MH];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int j = (wave * MH + t) * 32 + (lane & 31);
        B[t] = make_b(hyp[j * 2], hyp[j * 2 + 1], half);
        ucnt[t] = 0u;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* lbase = reinterpret_cast<const char*>(lds) + (lane & 31) * 32 + half * 16;
    bf16x8 Acr = *reinterpret_cast<const bf16x8*>(lbase), Ad = *reinterpret_cast<const bf16x8*>(lbase + 1024);
    f32x16 cr[2], d[2];
    cr[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[0], zero, 0, 0, 0);
    d[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, B[0], zero, 0, 0, 0);
    cr[1] = cr[0]; d[1] = d[0];
    for (int r = 0; r < reps; ++r) {
        for (int tile = 0; tile < ntiles; ++tile) {
            const int nt = tile + 1 < ntiles ? tile + 1 : tile;
            const bf16x8 Ncr = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES);
            const bf16x8 Nd = *reinterpret_cast<const bf16x8*>(lbase + nt * TILE_BYTES + 1024);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                f32x16& c0 = cr[t & 1];
                f32x16& d0 = d[t & 1];
                if (kind != 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], kind == 3 ? zero : c0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                f32x16& c1 = cr[(t + 1) & 1];
                f32x16& d1 = d[(t + 1) & 1];
                if (kind != 0) vote8i(ucnt[t], d1[0], c1[0], d1[1], c1[1], d1[2], c1[2], d1[3], c1[3], d1[4], c1[4], d1[5], c1[5], d1[6], c1[6], d1[7], c1[7]);
                __builtin_amdgcn_sched_barrier(0);
                if (kind != 1) d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ad : Nd, B[(t + 1) % MH], kind == 3 ? zero : d0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kind != 0) vote8i(ucnt[t], d1[8], c1[8], d1[9], c1[9], d1[10], c1[10], d1[11], c1[11], d1[12], c1[12], d1[13], c1[13], d1[14], c1[14], d1[15], c1[15]);
                __builtin_amdgcn_sched_barrier(0);
            }
            Acr = Ncr;
            Ad = Nd;
        }
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) keep += cr[0][i] + cr[1][i] + d[0][i] + d[1][i];
#pragma unroll
    for (int t = 0; t < MH; ++t) {
        const int ci = (int)(((ucnt[t] >> 23) * 383u) & 511u) + (keep == 12345.f);
        if (half == 0 && kind != 0 || keep == 54321.f) counts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 * MH * 32 + (wave * MH + t) * 32 + (lane & 31)] = ci;
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    constexpr int MH = 4;
    const int nh = 4 * MH * 32;  // hypotheses per workgroup
    const int ntiles = 8, npix = ntiles * 32;
    // ---- correctness on one workgroup, against exact arithmetic
    std::vector<float> hyp(nh * 2), pix(npix * 6);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    const float tau = 0.1425f;
    for (int j = 0; j < nh; ++j) { hyp[2 * j] = (rnd() - 0.5f) * 60.f; hyp[2 * j + 1] = (rnd() - 0.5f) * 60.f; }
    for (int i = 0; i < npix; ++i) {
        const float cx = (rnd() - 0.5f) * 80.f, cy = (rnd() - 0.5f) * 80.f;
        const float th = atan2f(-cy, -cx) + (rnd() - 0.5f) * 0.4f;  // roughly towards the origin
        const float ux = cosf(th), uy = sinf(th);
        float* p = &pix[i * 6];
        p[0] = uy; p[1] = -ux; p[2] = -(cx * uy - cy * ux);                 // cr = (h - c) x u
        p[3] = tau * ux; p[4] = tau * uy; p[5] = -tau * (cx * ux + cy * uy);  // d*tau = tau (h - c).u
    }
    float *dh, *dp; int* dc;
    const int max_wgs = cus * 4;
    hipMalloc(&dh, hyp.size() * 4); hipMalloc(&dp, pix.size() * 4); hipMalloc(&dc, sizeof(int) * max_wgs * nh);
    hipMemcpy(dh, hyp.data(), hyp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, pix.data(), pix.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma_vote<MH>, dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
    std::vector<int> got(nh);
    hipMemcpy(got.data(), dc, nh * 4, hipMemcpyDeviceToHost);
    long tot = 0, bad = 0, flips = 0;
    for (int j = 0; j < nh; ++j) {
        int want = 0;
        for (int i = 0; i < npix; ++i) {
            const float* p = &pix[i * 6];
            const double cr = (double)hyp[2 * j] * p[0] + (double)hyp[2 * j + 1] * p[1] + p[2];
            const double d = (double)hyp[2 * j] * p[3] + (double)hyp[2 * j + 1] * p[4] + p[5];
            want += d > fabs(cr);
        }
        tot += want;
        if (want != got[j]) { ++bad; flips += labs((long)want - got[j]); }
    }
    printf("layout check: %d hypotheses x %d pixels, %ld votes expected, %ld hypotheses differ (sum |diff| %ld)\n", nh, npix,
           tot, bad, flips);
    {   // clamp-vote variants need the 2^90 record scaling: check them on scaled pixel constants
        std::vector<float> pix2(pix);
        for (auto& x : pix2) x *= 0x1p90f;
        hipMemcpy(dp, pix2.data(), pix2.size() * 4, hipMemcpyHostToDevice);
        for (int variant = 3; variant <= 4; ++variant) {
            if (variant == 3) hipLaunchKernelGGL((k_mfma_vote2<MH, 3>), dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
            else hipLaunchKernelGGL((k_mfma_vote2<MH, 4>), dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
            std::vector<int> g2(nh);
            hipMemcpy(g2.data(), dc, nh * 4, hipMemcpyDeviceToHost);
            int nb = 0;
            for (int j = 0; j < nh; ++j) nb += g2[j] != got[j];
            printf("clamp-vote variant %d (2^90-scaled records): %d hypotheses differ from the compare kernel\n", variant, nb);
        }
        hipMemcpy(dp, pix.data(), pix.size() * 4, hipMemcpyHostToDevice);
    }
    for (int variant = 0; variant < 3; ++variant) {
        if (variant == 0) hipLaunchKernelGGL((k_mfma_vote2<MH, 0>), dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
        else if (variant == 1) hipLaunchKernelGGL((k_mfma_vote2<MH, 1>), dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
        else hipLaunchKernelGGL((k_mfma_vote2<MH, 2>), dim3(1, 1), dim3(256), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, 1);
        std::vector<int> g2(nh);
        hipMemcpy(g2.data(), dc, nh * 4, hipMemcpyDeviceToHost);
        int nb = 0;
        for (int j = 0; j < nh; ++j) nb += g2[j] != got[j];
        printf("asm epilogue variant %d: %d hypotheses differ from the compiler-scheduled kernel\n", variant, nb);
    }
    // ---- throughput
    for (int wpc = 1; wpc <= 4; ++wpc) {
        dim3 g(cus * wpc, 1), b(256);
        const int reps = 64;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mfma_vote<MH>, g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_mfma_vote<MH>, g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double tests = (double)g.x * nh * npix * reps;
        printf("waves/SIMD %d  bf16x3 mfma vote (MH=%d): %8.3f ms  %7.2f Tpairs/s\n", wpc, MH, ms, tests / ms / 1e9);
        for (int variant = 0; variant < 5; ++variant) {
            auto launch = [&] {
                if (variant == 0) hipLaunchKernelGGL((k_mfma_vote2<MH, 0>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else if (variant == 1) hipLaunchKernelGGL((k_mfma_vote2<MH, 1>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else if (variant == 2) hipLaunchKernelGGL((k_mfma_vote2<MH, 2>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else if (variant == 3) hipLaunchKernelGGL((k_mfma_vote2<MH, 3>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else hipLaunchKernelGGL((k_mfma_vote2<MH, 4>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("waves/SIMD %d    asm epilogue%s: %8.3f ms  %7.2f Tpairs/s\n", wpc, variant == 4 ? " clamp + add3 (1.5 op), interleaved" : variant == 3 ? " float clamp, interleaved" : variant == 2 ? " + interleaved" : variant ? " + pipelined  " : "              ", ms, tests / ms / 1e9);
        }
        for (int role = 0; role < 4; ++role) {
            auto launch = [&] {
                if (role == 0) hipLaunchKernelGGL((k_diag<MH, 0>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else if (role == 1) hipLaunchKernelGGL((k_diag<MH, 1>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else if (role == 2) hipLaunchKernelGGL((k_diag<MH, 2>), g, dim3(512), ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
                else hipLaunchKernelGGL((k_diag<MH, 3>), g, b, ntiles * TILE_BYTES, 0, dh, dp, dc, ntiles, reps);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("waves/SIMD %d    diag %s: %8.3f ms  %7.2f Tpairs/s-equivalent\n", role == 2 ? 2 * wpc : wpc,
                   role == 0 ? "MFMA only            " : role == 1 ? "VALU (1.5 op) only    " : role == 2 ? "MFMA waves + VALU waves (512 thr)" : "both in every wave    ", ms, tests / ms / 1e9);
        }
    }
    return 0;
}
