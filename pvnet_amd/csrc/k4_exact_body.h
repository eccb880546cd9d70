// k4_exact_body.h -- the exact mode's vote epilogue and the body of its dense scoring kernel; included by k4_score_exact.hip (the dense
// kernels) and k4_score_cull.hip (the merged dense + disc-culling launch).
#pragma once
#include "vote_common.h"

namespace pvd {
namespace {

// ------------------------------------------------------------------------------------------------------------
// K4 (exact mode, the default): the matrix-pipe scoring of score_mfma_kernel with the rounding-band epilogue
// described above b_col_exact(): counts EQUAL to the reference kernel's, 2.5 VALU operations per test.
// ------------------------------------------------------------------------------------------------------------
// Eight tests of the lane's hypothesis in 18 VALU operations (2.25 per test; cells of a whole work item, FOLD = 0):
//   x = d - |c|                   v_sub_f32 with a source modifier (d = dt', c = cr' of the two MFMAs): >= 1 = a vote outside the
//                                 band, <= -1 = a non-vote outside the band, anything in between = a test inside the band
//   dm = min(dm, |x|, |x'|)       v_min3_f32, two tests per instruction: the cell is clean iff dm >= 1
//   w = pknorm_u16(x, x')         v_cvt_pknorm_u16_f32, two tests per instruction: clamp(x) * 65535 -> 0xFFFF for a vote, 0 else
//   acc += w + w'                 v_add3_u32, four tests per instruction (wraps; votes_of_norm() decodes)
// (parameter names a_i / b_i: the dt' / cr' values -- round 3 passed a' = dt' - cr', b' = dt' + cr' and took min3(a', b', 1))
__device__ __forceinline__ void vote8x(unsigned& acc, float& dm, float a0, float b0, float a1, float b1, float a2, float b2,
                                       float a3, float b3, float a4, float b4, float a5, float b5, float a6, float b6,
                                       float a7, float b7) {
    float x0, x1, x2, x3;
    unsigned w0, w1;
    asm volatile(
        "v_sub_f32_e64 %2, %8, |%9|\n"
        "v_sub_f32_e64 %3, %10, |%11|\n"
        "v_sub_f32_e64 %4, %12, |%13|\n"
        "v_sub_f32_e64 %5, %14, |%15|\n"
        "v_min3_f32 %1, %1, |%2|, |%3|\n"
        "v_cvt_pknorm_u16_f32 %6, %2, %3\n"
        "v_sub_f32_e64 %2, %16, |%17|\n"
        "v_sub_f32_e64 %3, %18, |%19|\n"
        "v_min3_f32 %1, %1, |%4|, |%5|\n"
        "v_cvt_pknorm_u16_f32 %7, %4, %5\n"
        "v_sub_f32_e64 %4, %20, |%21|\n"
        "v_sub_f32_e64 %5, %22, |%23|\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        "v_min3_f32 %1, %1, |%2|, |%3|\n"
        "v_cvt_pknorm_u16_f32 %6, %2, %3\n"
        "v_min3_f32 %1, %1, |%4|, |%5|\n"
        "v_cvt_pknorm_u16_f32 %7, %4, %5\n"
        "v_add3_u32 %0, %6, %7, %0\n"
        : "+v"(acc), "+v"(dm), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(w0), "=&v"(w1)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5),
          "v"(a6), "v"(b6), "v"(a7), "v"(b7));
}
// votes in a vote8x accumulator: acc = 0xFFFF v_lo + 65536 * 0xFFFF v_hi (mod 2^32) for v_lo / v_hi votes in the low / high
// halves (even / odd tests), i.e. acc = 65536 (v_lo - v_hi) - v_lo: both counts < 65536 are recovered exactly
__device__ __forceinline__ int votes_of_norm(unsigned acc) {
    const unsigned lo = (0u - acc) & 0xFFFFu;
    const unsigned hi = (lo - ((acc + lo) >> 16)) & 0xFFFFu;
    return (int)(lo + hi);
}
// FOLD = 1 (one cell per step), round 4: the step's 40 operations in the order the two issue ports like (tools/ubench_issue.py
// "epi_sub order 2", profiles/r04_ubench_issue.txt: 152 cycles per step against 168 for the order "MFMA, subtractions, the
// rest" and 187 for round 3's min3 form).  A SIMD issues the fast class (v_sub_f32 here) through either of two ports, the slow
// class (min3, pknorm, add3, cmp, cndmask) through one, and an MFMA keeps the other busy for 32 cycles -- so behind every MFMA
// come the ten slow operations of the PREVIOUS eight tests (their x wait in eight registers), and only then the eight
// subtractions of the next eight tests:
//     MFMA dt(next) | slow + close (tests 8..15 of the previous step) | x = d - |c| (tests 0..7 of this step)
//     MFMA cr(next) | slow, open   (tests 0..7 of this step)         | x = d - |c| (tests 8..15 of this step)
// vote_subs: eight differences.  vote_slow_open: the first eight tests OPEN the cell -- acc and dm are produced, not updated.
// vote_slow_close: the second eight CLOSE it: the cell's votes join cnt only if no |x| fell below 1, and the verdict is
// shifted into flg (v_addc_co_u32 flg = 2 flg + bad: after the item's tiles bit (nti - 1 - tile) belongs to `tile`).
__device__ __forceinline__ void vote_subs(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5, float& x6,
                                          float& x7, float d0, float c0, float d1, float c1, float d2, float c2, float d3,
                                          float c3, float d4, float c4, float d5, float c5, float d6, float c6, float d7,
                                          float c7) {
    asm volatile(
        "v_sub_f32_e64 %0, %8, |%9|\n"
        "v_sub_f32_e64 %1, %10, |%11|\n"
        "v_sub_f32_e64 %2, %12, |%13|\n"
        "v_sub_f32_e64 %3, %14, |%15|\n"
        "v_sub_f32_e64 %4, %16, |%17|\n"
        "v_sub_f32_e64 %5, %18, |%19|\n"
        "v_sub_f32_e64 %6, %20, |%21|\n"
        "v_sub_f32_e64 %7, %22, |%23|\n"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
        : "v"(d0), "v"(c0), "v"(d1), "v"(c1), "v"(d2), "v"(c2), "v"(d3), "v"(c3), "v"(d4), "v"(c4), "v"(d5), "v"(c5),
          "v"(d6), "v"(c6), "v"(d7), "v"(c7));
}
__device__ __forceinline__ void vote_slow_open(unsigned& acc, float& dm, float x0, float x1, float x2, float x3, float x4,
                                               float x5, float x6, float x7) {
    unsigned w0, w1;
    asm volatile(
        "v_min_f32_e64 %1, |%4|, |%5|\n"
        "v_cvt_pknorm_u16_f32 %2, %4, %5\n"
        "v_min3_f32 %1, %1, |%6|, |%7|\n"
        "v_cvt_pknorm_u16_f32 %3, %6, %7\n"
        "v_add_u32_e32 %0, %2, %3\n"
        "v_min3_f32 %1, %1, |%8|, |%9|\n"
        "v_cvt_pknorm_u16_f32 %2, %8, %9\n"
        "v_min3_f32 %1, %1, |%10|, |%11|\n"
        "v_cvt_pknorm_u16_f32 %3, %10, %11\n"
        "v_add3_u32 %0, %2, %3, %0\n"
        : "=&v"(acc), "=&v"(dm), "=&v"(w0), "=&v"(w1)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
}
__device__ __forceinline__ void vote_slow_close(unsigned& cnt, unsigned& flg, unsigned& acc, float& dm, float x0, float x1,
                                                float x2, float x3, float x4, float x5, float x6, float x7) {
    unsigned w0, w1;
    asm volatile(
        "v_min3_f32 %3, %3, |%6|, |%7|\n"
        "v_cvt_pknorm_u16_f32 %4, %6, %7\n"
        "v_min3_f32 %3, %3, |%8|, |%9|\n"
        "v_cvt_pknorm_u16_f32 %5, %8, %9\n"
        "v_add3_u32 %2, %4, %5, %2\n"
        "v_min3_f32 %3, %3, |%10|, |%11|\n"
        "v_cvt_pknorm_u16_f32 %4, %10, %11\n"
        "v_min3_f32 %3, %3, |%12|, |%13|\n"
        "v_cvt_pknorm_u16_f32 %5, %12, %13\n"
        "v_cmp_nle_f32_e32 vcc, 1.0, %3\n"       // bad = !(dm >= 1)   (NaN cannot occur: |x| of finite x)
        "v_add3_u32 %2, %4, %5, %2\n"
        "s_nop 0\n"
        "v_cndmask_b32_e64 %2, %2, 0, vcc\n"     // the cell's votes, or nothing
        "v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n"   // flg = 2 flg + bad
        "v_add_u32_e32 %0, %0, %2\n"
        : "+v"(cnt), "+v"(flg), "+v"(acc), "+v"(dm), "=&v"(w0), "=&v"(w1)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7)
        : "vcc");
}

// FOLD = 0: one cell per (lane, hypothesis tile) and work item -- the minimum runs over all the item's pixel tiles
//           (cheapest epilogue; right when flagged cells are very rare: loose thresholds, band_fold());
// FOLD = 1: one cell per (lane, hypothesis tile, PIXEL tile) -- the test is made after every step (four more VALU operations,
//           vote8x_open / vote8x_close), and a flagged cell costs 16 literal tests instead of 16 * tiles.
// NACC = 2: two accumulator pairs -- the MFMAs of step i + 1 are issued around the votes of step i (the flat pipeline of the
//           approximate kernel);
// NACC = 1: one pair -- a wave issues the step's two MFMAs and consumes their results right away, the SIMD's other waves
//           fill the wait.  tools/ubench_exact.hip: 13.48 against 13.67 T tests/s at 3 waves per SIMD -- and 32 VGPRs fewer:
//           in 136 (RUNS) other streams' small stages can be resident beside this kernel, in 128 (strided items, a batch
//           alone) a SIMD holds four of its waves, 2 % faster alone and 6 % slower with batches in flight (r04d1-r04d5 in
//           profiles/r04_ab_runs.txt).  The default since the end of round 4; PVNET_SCORE_ACC=2 brings the two pairs back.
// RUNS (round 4; cells of one pixel tile only): the workgroup's items are a CONTIGUOUS run of the list -- while the (image,
//           key-point, hypothesis slice) stays the same, the B columns stay in registers, the hypotheses in LDS and the clean cells'
//           votes in their counters: loaded / flushed once per run instead of once per 256-pixel item.  Same-box A/B
//           (profiles/r04_ab_runs.txt): +2 % with six batches in flight (less work), -4.5 % for a batch alone (the contiguous
//           mapping itself: a launch of strided items ends more evenly) -- so it is what calls flagged PVNET_F_CONCURRENT run.
// HEAD (round 6): the body is the FIRST of the two a merged launch runs (score_exact_both_kernel: the dense items here, the
//           disc-culled ones in score_cull_body behind it) -- the kernel's register allocation and its closing clock stamps are
//           the second body's.
template <int MH, int FOLD, bool TIMED, int NACC, bool RUNS_, bool HEAD = false>
__device__ __forceinline__ void score_exact_body(VoteParams P) {
    if (HEAD) { }
    else if (MH == 8 && NACC == 2) PVNET_SPARE_VGPRS(167);
    else if (MH == 8 && RUNS_) PVNET_SPARE_VGPRS(135);
    else if (MH == 8) PVNET_SPARE_VGPRS(127);  // (one pair, strided items: what a batch ALONE runs, four waves per SIMD)
    else if (MH == 4) PVNET_SPARE_VGPRS(143);
    else PVNET_SPARE_VGPRS(111);
    unsigned long long* __restrict__ stamps = reinterpret_cast<unsigned long long*>(P.pix);
    if (TIMED && threadIdx.x == 0) stamps[2 * blockIdx.x] = (unsigned long long)wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int npx = P.wg_s * P.chunk, ntiles = npx >> 5;
    uint4* s_t = reinterpret_cast<uint4*>(smem);                        // A tiles: ntiles x 2 KB  (A_a rows | A_b rows)
    float4* s_raw = reinterpret_cast<float4*>(s_t + ntiles * TILE_U4);  // raw records of the pixel group
    float2* s_hyp = reinterpret_cast<float2*>(s_raw + npx);             // the item's 4 * MH * 32 hypotheses (for the flagged cells)
    unsigned* s_cells = reinterpret_cast<unsigned*>(s_hyp + 4 * MH * 32);  // flagged cells of this item (4 * MH * 64 slots)
    __shared__ int s_ncell;
    const int32_t* __restrict__ ctrl = P.ctrl;
    const int total = ctrl[P.b * CTRL_STRIDE];
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* lbase = s_t + half * 32 + col;

    // TIMED only: shader-clock cycles of this workgroup's wave 0 per phase (0 staging incl. the wait for its loads and the
    // barrier, 1 scoring loop, 2 count flush + cell list + barrier, 3 re-evaluation + the next item's first barrier)
    unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull}, tprev = 0ull;
#define PV_PHASE(i)                                                     \
    do {                                                                \
        if (TIMED) {                                                    \
            const unsigned long long now_ = (unsigned long long)clock64(); \
            ph[i] += now_ - tprev;                                      \
            tprev = now_;                                               \
        }                                                               \
    } while (0)
    if (TIMED) tprev = (unsigned long long)clock64();
    constexpr bool RUNS = RUNS_ && FOLD == 1;
    bf16x8 B[MH];
    unsigned cnt[MH];   // wrapped vote counters of the clean cells (vote8x / votes_of_norm)
#pragma unroll
    for (int t = 0; t < MH; ++t) cnt[t] = 0u;
    long long run_key = -1;      // (image, key-point) * slices + slice of the run the counters belong to
    int run_h0 = 0, run_items = 0;
    size_t run_bk = 0;
    auto flush_counts = [&](size_t fbk, int fh0) {   // the clean cells' votes of a finished item / run, hypothesis tiles in pairs
        int32_t* const pc = P.counts + fbk * P.hn_pad + fh0;
        int lanex = threadIdx.x;
        asm volatile("" : "+v"(lanex));   // (opaque: keeps the eight addresses from being hoisted above the scoring loop)
        lanex &= 63;
        if (MH >= 2) {  // lanes 0..31 finish tile t, lanes 32..63 tile t + 1
#pragma unroll
            for (int t = 0; t + 1 < MH; t += 2) {
                const int c = votes_of_norm(half_wave_sum2(cnt[t], cnt[t + 1]));
                if (c > 0) atomicAdd(pc + t * 32 + lanex, c);
            }
        } else {
            const int c = votes_of_norm(half_wave_sum2(cnt[0], cnt[0]));
            if (lanex < 32 && c > 0) atomicAdd(pc + lanex, c);
        }
#pragma unroll
        for (int t = 0; t < MH; ++t) cnt[t] = 0u;
    };
    const ItemRange ir = my_items<RUNS>(P, total);
    for (int item = ir.first; item < ir.end; item += ir.step) {
        const int4 desc = P.items[item];
        if (item_culled(desc.y)) continue;   // (workgroup-uniform) a key-point the disc-culling body scores
        const int bi = desc.x, k = item_kp(desc.y), cg = desc.z, hq = desc.w;
        const int tn = ctrl[bi * CTRL_STRIDE + C_TN];
        const float rho = band_rho(tn);
        const size_t bk = (size_t)bi * P.vn + k;
        const int32_t* const org = band_origin_ptr(P, bk);   // the band's origin for this key-point (hypothesis_kernel)
        const float ox = (float)org[0], oy = (float)org[1];
        const int tpad = (tn + PAD - 1) / PAD * PAD;
        const int hslice = hq * 4 * MH * 32;            // first hypothesis of this work item
        const int h0 = hslice + wave * MH * 32;         // this wave's first hypothesis
        const long long key = (long long)bk * (P.hgroups / P.wg_g) + hq;
        // workgroup-uniform: a new run.  The packed-norm counters hold < 65536 votes per half: a lane adds 8 votes per half and
        // pixel tile, half_wave_sum2 joins two lanes -- 16 * ntiles per item, so a run is cut after 65535 / (16 * ntiles) items
        // (a fixed 256 overflowed from 16 tiles per item on, PVNET_SCORE_CHUNK >= 256)
        const bool fresh = !RUNS || key != run_key || run_items >= 65535 / (16 * ntiles);

        lds_barrier();  // the previous item's tiles, raw records and cell list have been consumed
        PV_PHASE(3);
        if (threadIdx.x == 0) s_ncell = 0;
        int tid = threadIdx.x;  // opaque copies of the thread index: what staging and re-evaluation derive from it is
        asm volatile("" : "+v"(tid));  // recomputed per item instead of staying in VGPRs across the scoring loop
        float2 hreg[(4 * MH * 32 + 255) / 256];  // the run's hypotheses: loaded now, parked in LDS after the staging arithmetic
        if (fresh) {
            if (RUNS && run_key >= 0) flush_counts(run_bk, run_h0);
            run_key = key;
            run_bk = bk;
            run_h0 = h0;
            run_items = 0;
#pragma unroll
            for (int j = 0; j < (4 * MH * 32 + 255) / 256; ++j)
                hreg[j] = (tid + 256 * j < 4 * MH * 32) ? P.hyp[bk * P.hn_pad + hslice + tid + 256 * j] : make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const uint4 raw = P.hypb[(bk * P.hn_pad + h0 + t * 32 + (tid & 31)) * 2 + ((tid >> 5) & 1)];  // (column, half-wave)
                B[t] = __builtin_bit_cast(bf16x8, raw);
            }
        }
        ++run_items;
        for (int i = tid; i < npx; i += 256) {
            const int p = cg * npx + i;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < tpad) q = P.rec[bk * P.cap + p];
            s_raw[i] = q;
            uint4* t = s_t + (i >> 5) * TILE_U4 + (i & 31);
            uint4 r0, r1, r2, r3;  // (in registers first: by reference into LDS every assignment inside would be a store)
            float mu_unused;
            a_rows_exact(q, P.tau, ox, oy, rho, r0, r1, r2, r3, mu_unused);
            t[0] = r0;
            t[32] = r1;
            t[64] = r2;
            t[96] = r3;
        }
        if (fresh) {
#pragma unroll
            for (int j = 0; j < (4 * MH * 32 + 255) / 256; ++j)
                if (tid + 256 * j < 4 * MH * 32) s_hyp[tid + 256 * j] = hreg[j];
        }
        lds_barrier();
        PV_PHASE(0);

        float dmn[MH];      // min |x| of the open cell so far
        unsigned flg[MH];   // FOLD: bit (nti - 1 - tile) set = pixel tile `tile` holds a test inside the band
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            flg[t] = 0u;
            dmn[t] = 3.0e38f;
        }
        const int left = (tpad - cg * npx + 31) >> 5;
        const int nti = left < ntiles ? left : ntiles;
        bf16x8 Aa = __builtin_bit_cast(bf16x8, lbase[0]), Ab = __builtin_bit_cast(bf16x8, lbase[64]);
        // FOLD: the x of the eight tests whose slow operations are still due (see vote_subs), the open cell's votes / minimum.
        // Before the first step nothing is due: x = -4 counts no vote and flags nothing, the cell it "closes" shifts a zero
        // into a zero flag word of the last hypothesis tile.
        float x0 = -4.f, x1 = -4.f, x2 = -4.f, x3 = -4.f, x4 = -4.f, x5 = -4.f, x6 = -4.f, x7 = -4.f, dmo = 3.0e38f;
        unsigned acc = 0u;
#define PV_XS x0, x1, x2, x3, x4, x5, x6, x7
#define PV_LO(v, w) v[0], w[0], v[1], w[1], v[2], w[2], v[3], w[3], v[4], w[4], v[5], w[5], v[6], w[6], v[7], w[7]
#define PV_HI(v, w) v[8], w[8], v[9], w[9], v[10], w[10], v[11], w[11], v[12], w[12], v[13], w[13], v[14], w[14], v[15], w[15]
        if (NACC == 1) {
            for (int tile = 0; tile < nti; ++tile) {
                const int nt = tile + 1 < nti ? tile + 1 : tile;
                const bf16x8 Na = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4]);
                const bf16x8 Nb = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4 + 64]);
#pragma unroll
                for (int t = 0; t < MH; ++t) {
                    const f32x16 va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa, B[t], zero, 0, 0, 0);
                    const f32x16 vb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, B[t], zero, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (FOLD) {   // the previous step's last 15 operations fill the wait for this step's MFMAs
                        vote_slow_close(cnt[(t + MH - 1) % MH], flg[(t + MH - 1) % MH], acc, dmo, PV_XS);
                        asm volatile("s_nop 3");   // (the votes are inline asm: the wait states are ours, tools/check_mfma_hazard.py)
                        vote_subs(PV_XS, PV_LO(va, vb));
                        vote_slow_open(acc, dmo, PV_XS);
                        vote_subs(PV_XS, PV_HI(va, vb));
                    } else {
                        asm volatile("s_nop 11");
                        __builtin_amdgcn_sched_barrier(0);
                        vote8x(cnt[t], dmn[t], PV_LO(va, vb));
                        vote8x(cnt[t], dmn[t], PV_HI(va, vb));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                Aa = Na;
                Ab = Nb;
            }
        } else {
        f32x16 va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aa, B[0], zero, 0, 0, 0);
        f32x16 vb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, B[0], zero, 0, 0, 0);
        for (int tile = 0; tile < nti; ++tile) {
            const int nt = tile + 1 < nti ? tile + 1 : tile;
            const bf16x8 Na = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4]);
            const bf16x8 Nb = __builtin_bit_cast(bf16x8, lbase[nt * TILE_U4 + 64]);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const f32x16 va2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Aa : Na, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (FOLD) {
                    vote_slow_close(cnt[(t + MH - 1) % MH], flg[(t + MH - 1) % MH], acc, dmo, PV_XS);
                    vote_subs(PV_XS, PV_LO(va, vb));
                } else {
                    vote8x(cnt[t], dmn[t], PV_LO(va, vb));
                }
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 vb2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t + 1 < MH ? Ab : Nb, B[(t + 1) % MH], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (FOLD) {
                    vote_slow_open(acc, dmo, PV_XS);
                    vote_subs(PV_XS, PV_HI(va, vb));
                } else {
                    vote8x(cnt[t], dmn[t], PV_HI(va, vb));
                }
                __builtin_amdgcn_sched_barrier(0);
                va = va2;
                vb = vb2;
            }
            Aa = Na;
            Ab = Nb;
        }
        }
        if (FOLD) vote_slow_close(cnt[MH - 1], flg[MH - 1], acc, dmo, PV_XS);   // the last step's second half
#undef PV_XS
#undef PV_LO
#undef PV_HI
        PV_PHASE(1);
        // ---- clean cells: their counts; flagged cells: into the item's list
        const unsigned all_groups = 1u;  // FOLD = 0: the one cell of the item
        int colx = col;  // opaque copy: keeps the eight per-tile addresses below from being hoisted above the scoring loop,
        asm volatile("" : "+v"(colx));  // where they would cost 20 VGPRs at the point of highest pressure
        const bool padded = h0 + MH * 32 > P.hn;  // wave-uniform: only the last slice can hold padding columns
        if (!FOLD) {
#pragma unroll
            for (int t = 0; t < MH; ++t) cnt[t] = dmn[t] >= BAND_CLEAN ? cnt[t] : 0u;  // a flagged cell's votes are discarded
        }
        if (!RUNS) flush_counts(bk, h0);  // (RUNS: when the run ends)
        // (round 6: ONE slot reservation per wave and item -- the eight ballots and their popcounts are scalar work, the wave's
        //  cells go behind one LDS atomic; one reservation per hypothesis tile, most of them taken, cost ~70 vector operations more)
        unsigned long long bal[MH];
        int ncw = 0;
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            unsigned mask = FOLD ? flg[t] : (dmn[t] >= BAND_CLEAN ? 0u : all_groups);
            if (padded && h0 + t * 32 + colx >= P.hn) mask = 0u;  // padding columns of the last slice: nobody reads their counts
            flg[t] = mask;
            bal[t] = __ballot(mask != 0u);
            ncw += (int)__popcll(bal[t]);
        }
        if (ncw) {  // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ncell, ncw);
            base = __builtin_amdgcn_readfirstlane(base);
            const unsigned cell0 = (unsigned)(wave * MH * 32 + colx) | ((unsigned)half << 10);
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                if (bal[t]) {  // wave-uniform
                    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal[t] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[t], 0u));
                    if (flg[t] != 0u) s_cells[slot] = (cell0 + (unsigned)(t * 32)) | (flg[t] << 11);
                    base += (int)__popcll(bal[t]);
                }
            }
        }
        lds_barrier();
        PV_PHASE(2);
        // ---- flagged cells, decided by the reference's arithmetic: 16 lanes per cell, one pixel row each
        const int ncell = s_ncell;
        if (ncell > 0) {
            int tid2 = threadIdx.x;
            asm volatile("" : "+v"(tid2));
            const int grp = tid2 >> 4, q = tid2 & 15;
            int ntests = 0;
            for (int e = grp; e < ncell; e += 16) {
                const unsigned cell = s_cells[e];
                const int hl = (int)(cell & 1023u), hf = (int)((cell >> 10) & 1u);
                unsigned m = cell >> 11;
                const float2 hv = s_hyp[hl];
                const int row = (q >> 2) * 8 + hf * 4 + (q & 3);  // the 16 rows a lane of that half-wave holds
                int votes = 0;
                while (m) {
                    const int g = __ffs((int)m) - 1;  // FOLD: bit g = pixel tile nti - 1 - g (vote8x_close shifts them in)
                    m &= m - 1u;
                    const int t0 = FOLD ? nti - 1 - g : 0, t1 = FOLD ? t0 + 1 : nti;
                    for (int tile = t0; tile < t1; ++tile) {
                        const float4 r = s_raw[tile * 32 + row];
                        votes += inlier_literal(r.x, r.y, r.z, r.w, hv.x, hv.y, P.thresh) ? 1 : 0;
                        ++ntests;
                    }
                }
                votes += __shfl_xor(votes, 8, 64);
                votes += __shfl_xor(votes, 4, 64);
                votes += __shfl_xor(votes, 2, 64);
                votes += __shfl_xor(votes, 1, 64);
                if (q == 0 && votes > 0) atomicAdd(P.counts + bk * P.hn_pad + hslice + hl, votes);
            }
            if (P.flags & PVNET_F_BAND_STATS) {  // development aid: how much was re-evaluated (tools/exact_probe.py)
                if (tid2 == 0) atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 4, ncell);
                if (q == 0 && ntests > 0) atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 5, ntests);
            }
        }
    }
    if (RUNS && run_key >= 0) flush_counts(run_bk, run_h0);
    if (TIMED && !HEAD) {
        lds_barrier();
        PV_PHASE(3);
        if (threadIdx.x == 0) {
            stamps[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
#pragma unroll
            for (int i = 0; i < 4; ++i) stamps[2 * gridDim.x + 4 * blockIdx.x + i] = ph[i];  // (tools/phase_probe.py)
        }
    }
#undef PV_PHASE
}

}  // namespace
}  // namespace pvd
