// k4_score_mfma.hip -- K4, approximate mode (PVNET_F_APPROX): matrix-pipe scoring without the rounding band
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

// ------------------------------------------------------------------------------------------------------------
// K4 (fast mode): matrix-pipe scoring.  A work item = (image, key-point, pixel group of wg_s chunks, slice of
// 4 * MH * 32 hypotheses).  The workgroup turns the group's records into bf16x3 A rows in LDS (32-pixel tiles,
// A_cr | A_dt); each of its 4 waves keeps the B columns of MH * 32 hypotheses in registers (written by K3) and, per
// tile, issues 2 MFMAs per 32 hypotheses: cr and dt of 32 x 32 (pixel, hypothesis) pairs land in the lane that
// owns the hypothesis (column = lane & 31, 16 rows per lane), so the vote is cnt += clamp(dt - |cr|) (vote8):
// 2 VALU operations per test instead of 6, the other four run on the matrix pipe at bf16 rate.
// ------------------------------------------------------------------------------------------------------------

// Eight votes of the lane's hypothesis, 1.5 plain VALU operations per test.  The staged rows carry M = 2^k * direction with
// max(|Mx|, |My|) in [2^60, 2^61) (vote_scale), so any non-zero margin is >= 1 in magnitude and the clamp output modifier turns t = clamp(dt - |cr|) into exactly
// 1.0f or 0.0f (NaN -> 0), i.e. the bit pattern 0x3F800000 or 0: no compare, no SGPR mask, no carry chain.  v_add3_u32
// then sums TWO of them per instruction into a 32-bit integer that is allowed to wrap:
//     acc = n * 0x3F800000 mod 2^32 = ((127 n) mod 512) << 23,
// and 127 is invertible mod 512 (127 * 383 = 95 * 512 + 1), so n = ((acc >> 23) * 383) & 511 for any n < 512
// (votes_of()).  A lane accumulates 16 tests per pixel tile and at most 16 tiles per work item (256 < 512).
// Measured beside the MFMAs (tools/ubench_mfma.hip): 18.3 T tests/s against 16.5 T for v_sub clamp + v_add_f32 and
// 15.7 T for v_cmp + v_cndmask + v_addc.  Hand-placed: every difference is consumed >= 3 instructions after it was
// produced.
__device__ __forceinline__ void vote8(unsigned& acc, float d0, float c0, float d1, float c1, float d2, float c2, float d3,
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
__device__ __forceinline__ int votes_of(unsigned acc) { return (int)(((acc >> 23) * 383u) & 511u); }

// TIMED (profiling entry pvnet_vote_v3_stage_repeat only): every workgroup stores the constant-rate device clock at its
// first and last instruction into its own slot of the (idle during this stage) `pix` buffer -- two plain 8-byte
// stores per workgroup; max end - min start over the slots is the kernel's duration as a kernel trace reports it,
// measured live and free of launch gaps.
template <int MH, bool TIMED>
__global__ __launch_bounds__(256) void score_mfma_kernel(VoteParams P) {
    if (MH == 8) PVNET_SPARE_VGPRS(159); else if (MH == 4) PVNET_SPARE_VGPRS(143); else PVNET_SPARE_VGPRS(111);
    unsigned long long* __restrict__ stamps = reinterpret_cast<unsigned long long*>(P.pix);
    if (TIMED && threadIdx.x == 0) stamps[2 * blockIdx.x] = (unsigned long long)wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_t = reinterpret_cast<uint4*>(smem);
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int npx = P.wg_s * P.chunk, ntiles = npx >> 5;
    const int32_t* __restrict__ ctrl = P.ctrl;
    const int total = ctrl[P.b * CTRL_STRIDE];
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* lbase = s_t + half * 32 + col;  // this lane's 16 bytes of every A row block (tile layout: TILE_U4)

    const ItemRange ir = my_items(P, total);
    for (int item = ir.first; item < ir.end; item += ir.step) {
        const int4 desc = P.items[item];  // (image, key-point, chunk group, hypothesis slice), planned by K3
        const int bi = desc.x, k = item_kp(desc.y), cg = desc.z, hq = desc.w;
        const int tn = ctrl[bi * CTRL_STRIDE + C_TN];
        const float ox = (float)ctrl[bi * CTRL_STRIDE + C_OX], oy = (float)ctrl[bi * CTRL_STRIDE + C_OY];
        const size_t bk = (size_t)bi * P.vn + k;
        const int tpad = (tn + PAD - 1) / PAD * PAD;
        const int h0 = hq * 4 * MH * 32 + wave * MH * 32;  // this wave's first hypothesis

        bf16x8 B[MH];
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            const uint4 raw = P.hypb[(bk * P.hn_pad + h0 + t * 32 + col) * 2 + half];
            B[t] = __builtin_bit_cast(bf16x8, raw);
        }
        lds_barrier();  // the previous item's tiles have been consumed
        for (int i = threadIdx.x; i < npx; i += 256) {  // thread = pixel: expand its record once per workgroup
            const int p = cg * npx + i;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < tpad) q = P.rec[bk * P.cap + p];
            float4 a;
            float2 b;
            make_pixrec(q, P.tau, ox, oy, a, b);  // a = (My, -Mx, -Ec, Tx), b = (Ty, -Ed); zero record -> zero rows
            uint4* t = s_t + (i >> 5) * TILE_U4 + (i & 31);
            a_row(a.x, a.y, a.z, t[0], t[32]);
            a_row(a.w, b.x, b.y, t[64], t[96]);
        }
        lds_barrier();

        unsigned cnt[MH];  // wrapped vote accumulators (vote8): 16 * ntiles <= 256 votes each
#pragma unroll
        for (int t = 0; t < MH; ++t) cnt[t] = 0u;
        // Flat software pipeline over (pixel tile, hypothesis tile) steps: the two MFMAs of step i+1 are issued
        // around the votes of step i (half of them behind each), on ping-pong accumulators.
        bf16x8 Acr = __builtin_bit_cast(bf16x8, lbase[0]), Adt = __builtin_bit_cast(bf16x8, lbase[64]);
        f32x16 cr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Acr, B[0], zero, 0, 0, 0);
        f32x16 dt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Adt, B[0], zero, 0, 0, 0);
        // the image's last pixel group is usually partial: tiles beyond the padded pixel count hold only zero rows
        const int left = (tpad - cg * npx + 31) >> 5;
        const int nti = left < ntiles ? left : ntiles;
        for (int tile = 0; tile < nti; ++tile) {
            const int nt = tile + 1 < nti ? tile + 1 : tile;  // (after the last tile: a harmless repeat)
            const bf16x8 Ncr = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4]);
            const bf16x8 Ndt = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4 + 64]);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const f32x16 cr2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Acr : Ncr, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                vote8(cnt[t], dt[0], cr[0], dt[1], cr[1], dt[2], cr[2], dt[3], cr[3], dt[4], cr[4], dt[5], cr[5], dt[6],
                      cr[6], dt[7], cr[7]);
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 dt2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Adt : Ndt, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                vote8(cnt[t], dt[8], cr[8], dt[9], cr[9], dt[10], cr[10], dt[11], cr[11], dt[12], cr[12], dt[13], cr[13],
                      dt[14], cr[14], dt[15], cr[15]);
                __builtin_amdgcn_sched_barrier(0);
                cr = cr2;
                dt = dt2;
            }
            Acr = Ncr;
            Adt = Ndt;
        }
        // the group's counts: atomic adds into counts[] (default), or one uint16 row per chunk GROUP for K5 to sum
        uint16_t* po = P.partial + (bk * P.max_chunks + cg) * P.hn_pad + h0;
        // (round 5, found by the knob fuzz: the pair form decodes the SUM of two lanes' wrapped accumulators -- up to 32 votes per
        //  pixel tile, i.e. exactly 512 = 0 mod 512 when all 512 pixels of a 16-tile item vote (PVNET_SCORE_CHUNK=256): only below 16 tiles)
        if (MH >= 2 && P.atomic_counts && ntiles < 16) {  // tiles in pairs: lanes 0..31 finish tile t, lanes 32..63 tile t + 1
#pragma unroll
            for (int t = 0; t + 1 < MH; t += 2) {
                const int c = votes_of(half_wave_sum2(cnt[t], cnt[t + 1]));
                if (c > 0) atomicAdd(P.counts + bk * P.hn_pad + h0 + t * 32 + lane, c);
            }
        } else {
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const int ci = votes_of(cnt[t]);
                const int c = half_wave_sum(ci);  // the half-waves hold different rows of the column
                if (P.atomic_counts) {
                    if (half == 0 && c > 0) atomicAdd(P.counts + bk * P.hn_pad + h0 + t * 32 + col, c);
                } else if (half == 0) {
                    po[t * 32 + col] = (uint16_t)c;
                }
            }
        }
    }
    if (TIMED) {
        lds_barrier();
        if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
    }
}


}  // namespace

int launch_score_mfma(const VoteParams& P, dim3 g, size_t lds, hipStream_t s, bool timed) {
    const int mh = P.wg_g * P.hpl / 2;  // hypotheses per item = wg_g * 64 * hpl = 4 waves * mh * 32
    const dim3 t(256);
#ifdef PVNET_DEV
    if (timed) {  // same code + two clock stamps per workgroup (pvnet_vote_v3_stage_repeat)
        switch (mh) {
            case 1: hipLaunchKernelGGL((score_mfma_kernel<1, true>), g, t, lds, s, P); break;
            case 2: hipLaunchKernelGGL((score_mfma_kernel<2, true>), g, t, lds, s, P); break;
            case 4: hipLaunchKernelGGL((score_mfma_kernel<4, true>), g, t, lds, s, P); break;
            case 8: hipLaunchKernelGGL((score_mfma_kernel<8, true>), g, t, lds, s, P); break;
            default: return PVNET_E_UNSUPPORTED;
        }
        return 0;
    }
#else
    if (timed && mh != 8) return PVNET_E_UNSUPPORTED;   // (release: the clock-stamping form exists at 8 tiles per wave only)
    if (timed) {
        hipLaunchKernelGGL((score_mfma_kernel<8, true>), g, t, lds, s, P);
        return 0;
    }
#endif
    switch (mh) {
        case 1: hipLaunchKernelGGL((score_mfma_kernel<1, false>), g, t, lds, s, P); break;
        case 2: hipLaunchKernelGGL((score_mfma_kernel<2, false>), g, t, lds, s, P); break;
        case 4: hipLaunchKernelGGL((score_mfma_kernel<4, false>), g, t, lds, s, P); break;
        case 8: hipLaunchKernelGGL((score_mfma_kernel<8, false>), g, t, lds, s, P); break;
        default: return PVNET_E_UNSUPPORTED;
    }
    return 0;
}

}  // namespace pvd
