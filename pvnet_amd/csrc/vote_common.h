// vote_common.h -- shared by the translation units of libpvnet_vote.so (one per stage: k1_mask.hip ... vote_host.hip): the
// parameter block every kernel takes, the workspace conventions, the arithmetic of the reference's two kernels in its float32
// order, the bf16x3 operands of the matrix-pipe scoring kernels and the exact mode's rounding band.  Everything lives in
// namespace pvd; device helpers are inline, host helpers are defined in vote_host.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "pvnet_rng.h"
#include "pvnet_vote.h"

namespace pvd {

// (double)x < 1e-6   <=>   x <= kF1e6   for float x   [float(1e-6) = 0x1.0c6f7ap-20 < 1e-6]
__device__ constexpr float kF1e6 = 0x1.0c6f7ap-20f;

constexpr int CTRL_STRIDE = 8;
enum { C_TN0 = 0, C_TN = 1, C_STATUS = 2, C_ITEM_BASE = 3, C_NCHUNKS = 4, C_OX = 5, C_OY = 6 };

constexpr int SEG_WORDS = 64;          // a segment = 64 words = 4096 pixels: the unit of K1 / K2 workgroups
#ifndef PVNET_K1_WAVES
#define PVNET_K1_WAVES 8
#endif
#ifndef PVNET_RT
#define PVNET_RT 512
#endif
#ifndef PVNET_SMALL_PRIO
#define PVNET_SMALL_PRIO 3
#endif
#ifndef PVNET_CULL_Q_MILLI
#define PVNET_CULL_Q_MILLI 1000 // a key-point votes for disc culling when all but one of its eight candidate intersections lie within
                                // 1.0 rho tan(theta0) of their median point (k3_hypotheses.hip kp_preamble; profiles/r06y_cull_crossover.txt)
#endif
#ifndef PVNET_CULL_DEFAULT
#define PVNET_CULL_DEFAULT 2   // what PVNET_SCORE_CULL = -1 (not set) means: 0 = never, 1 = every key-point, 2 = the key-points K3 selects
#endif
// the small latency-bound stages ask for issue priority over the co-resident scoring waves of other batches (s_setprio 3).
// Round 1 measured nothing from it (nothing WAS resident beside the scoring kernel); since round 3 their workgroups share
// SIMDs with the one-accumulator scoring kernel of concurrent callers, and their dependent chains finishing sooner is worth
// +2.2 % with six batches in flight (profiles/r03_ab_small_stage_shapes.txt), nothing alone
__device__ __forceinline__ void small_stage_prio() {
    if (PVNET_SMALL_PRIO) __builtin_amdgcn_s_setprio(PVNET_SMALL_PRIO);
}
// Every kernel asks for at least one granule (8) of VGPRs MORE than it uses: an empty asm statement that names a high
// register as clobbered raises .amdhsa_next_free_vgpr without costing an instruction.  Reason (round 2, the compaction
// flake; profiles/r02_compaction_flake_investigation.txt, tools/experiments/k2_flake/): compact_kernel<false,1>
// returned up to 64 records of one wave from pixels a few ranks away in 40-100 % of the runs whenever (a) its code used
// the wave's VGPR allocation up to the last granule and (b) two or more of its workgroups shared a CU.  The SAME
// instruction stream, assembled with .amdhsa_next_free_vgpr raised from 24 to 32 (nothing else changed), never failed in
// 300 runs; moving the four highest registers' roles to v8..v11 at the original allocation did not fail either; with
// one workgroup per CU (100 KB of dynamic LDS) it did not fail.  What exactly goes wrong in the top granule was not found
// (ruled out: wait counts, the barrier, LDS visibility, store-data / LDS-address / 64-bit-shift hazards, loads in flight
// at s_endpgm; a stand-alone register-persistence stress test does not reproduce it), so the rule is empirical -- and
// tools/check_kernel_resources.py enforces it for every kernel of the library at build time.
#define PVNET_SPARE_VGPRS_(r) asm volatile("" ::: "v" #r)
#define PVNET_SPARE_VGPRS(r) PVNET_SPARE_VGPRS_(r)
constexpr int K1_WAVES = PVNET_K1_WAVES;  // waves per K1 workgroup (one workgroup = one segment).  Alone: 4 -> 28 us, 8 -> 25.6 us,
                                          // 16 -> 24.5 us (batch 32, int64 masks); 8 since round 3: a workgroup of 16 waves needs
                                          // 160 VGPRs per SIMD at once, one of 8 fits beside the resident scoring waves of another
                                          // batch (PVNET_F_CONCURRENT): +3 % with six batches in flight for -0.5 % alone
constexpr int K1_WORDS_PER_WAVE = SEG_WORDS / K1_WAVES;  // independent loads in flight per lane
static_assert(SEG_WORDS == 64, "a segment's bit words are stored by the 64 lanes of one wave (mask_bits kernels, compact_kernel's scan)");
static_assert(K1_WAVES <= SEG_WORDS / 2 && (SEG_WORDS / 2) % K1_WAVES == 0,
              "PVNET_K1_WAVES must divide 32: mask_bits_pair_kernel gives every wave (SEG_WORDS / 2) / K1_WAVES double words");
constexpr int K2_WORDS_PER_BLOCK = SEG_WORDS;
// thinning: keep a pixel <=> pvnet_thin_bin(random word) < K (pvnet_rng.h; oracle: subsample_threshold): 1/1024 steps of the
// probability down to 1/64, sixteen steps per octave below (round 4; rounds 2-3: the top ten bits only)
constexpr int THIN_BINS = (PVNET_THIN_LAST + 1 + 127) / 128 * 128;   // 1536: histogram length, an EVEN number of bins per lane
constexpr int PAD = 8;                 // scoring consumes records 8 at a time; tails are padded with sentinels
constexpr int TILE_U4_ = 128;          // uint4 per 32-pixel A tile of the matrix-pipe kernels (= TILE_U4 below)

struct VoteParams {
    const void* mask;
    int64_t ms0, ms1, ms2, ms_c;
    int mask_dtype, mask_linear, num_classes;
    int vertex_type, logits_type;  // VT_* of the field / of the class logits
    const float* vertex;           // (typed by vertex_type)
    int64_t vs0, vs1, vs2, vs3, vs4;
    int b, h, w, vn, hn, npix, words, cap, chunk, max_chunks, hpl, hgroups, hn_pad, wg_g, wg_s, mode, score_xcd, atomic_counts;
    float thresh, tau;
    float kband;   // exact mode: half-width of the rounding band as a fraction of |d| |u| (band_constant())
    int layout_fp; // fingerprint of the workspace layout this call was planned with (layout_fingerprint()), kept in ctrl
    int exact;     // 1: matrix-pipe scoring + literal re-evaluation of the cells that hold a pair inside the band
    int fold1;     // exact mode: 1 = a cell is one pixel tile (16 tests per lane), 0 = the whole work item (band_fold1())
    int min_num, max_num;
    uint64_t seed;
    int image_base;
    const int32_t* idxs;
    uint32_t flags;
    int32_t* ctrl;
    int4* items;
    int32_t* seg;
    int32_t* seg0;
    uint16_t* cum;    // [b][nseg][THIN_BINS] cumulative histograms of the thinning decisions; NULL when max_num >= h*w
    int nseg;
    uint64_t* bits;
    int32_t* pix;
    float4* rec;
    float2* hyp;
    uint4* hypb;      // fast mode: the same hypotheses as bf16x3 B operands of the scoring MFMAs, [b][vn][hn_pad][2]
    uint16_t* partial;
    int32_t* counts;
    int32_t* win;
    float* out;
    int32_t* status;
    // disc culling (round 5; section "K4 -- disc culling" below): hypotheses sorted along a Hilbert curve per key-point
    int cull;            // disc culling (exact mode, 8 tiles per wave, 256-pixel items, hn_pad = 1024): 0 = never, 1 = every key-point
                         // (PVNET_SCORE_CULL=1), 2 = the key-points K3 selects (kp_preamble) -- the default where the layout supports it
    float cull_q;        // selection threshold of cull = 2: spread of the candidate intersections <= cull_q rho tan(theta0)
    int32_t* perm;       // [b][vn][hn_pad] sorted position -> caller's hypothesis index
    float2* hyps;        // [b][vn][hn_pad] the hypotheses in sorted order (literal re-evaluation of flagged cells)
    int32_t* cnts;       // [b][vn][hn_pad] inlier counts of the culled key-points in sorted order (K4 accumulates: 64 consecutive
                         // slots per atomic -- adding at perm[] instead scattered every flush over ~28 cache lines and cost the strided
                         // kernel 90 us, r06b; K5 returns them to caller order)
    uint4* hypc;         // [b][vn][hn_pad / 32][2] B column of every 32-hypothesis tile's CENTRE, scaled by 1 / (radius + band)
    float* hypg;         // [b][vn][hn_pad / 32]    g = radius term / (radius term + band term) of the tile (0: every pixel uncertain)
};

// per-call flags: int32 [8] behind the culling marks.  CF_ANY_CULLED: some image of this call is disc-culled -- zeroed by K2 (the block of
// image 0's last segment), set by K3's plan blocks, read by the merged scoring launch, whose workgroups enter the culling body only then
// (a call without culled key-points pays one scalar load for the merged launch)
constexpr int CF_ANY_CULLED = 0;
// CF_VOTES_NOW: images of this call whose key-points voted for disc culling (K3's plan blocks add theirs); CF_BATCH_OK: may this call's
// images be culled at all?  K2 (the same thread that zeroes CF_ANY_CULLED) decides it from the PREVIOUS call on this workspace: yes when
// at least two thirds of that call's images voted for culling, or when the workspace holds no previous call of this layout (the
// fingerprint in ctrl's global row).  A culled image makes the hypothesis launch 9 us longer for the whole batch (its key-points' sort
// blocks are the launch's longest chain), which ONE culled image of 32 never earns back (profiles/r06y_cull_crossover.txt: 3 % culled,
// 193.7 us against 176): callers that vote batch after batch of one kind of field -- a network's output -- get the selection of the
// previous batch's majority, from the second call on.  Like every selection it changes the time only, never a count.
constexpr int CF_VOTES_NOW = 1, CF_BATCH_OK = 2;
__device__ __forceinline__ int32_t* call_flags_ptr(const VoteParams& P) {
    return P.ctrl + (size_t)(P.b + 1) * CTRL_STRIDE + 3 * (size_t)P.b * P.vn;
}

// ------------------------------------------------------------------------------------------------------------
// arithmetic shared by several kernels
// ------------------------------------------------------------------------------------------------------------

// ransac_voting_kernel.cu:28-48 in its float32 operation order, one rounding per operation (no FMA contraction)
__device__ __forceinline__ void hyp_intersect(float ux0, float uy0, float cx0, float cy0, float ux1, float uy1,
                                              float cx1, float cy1, float& ox, float& oy) {
#pragma clang fp contract(off)
    const float nx0 = uy0, ny0 = -ux0, nx1 = uy1, ny1 = -ux1;
    const float dety = nx1 * ny0 - nx0 * ny1;
    const float detx = ny1 * nx0 - ny0 * nx1;
    ox = 0.f;
    oy = 0.f;
    if (fabsf(dety) <= kF1e6 || fabsf(detx) <= kF1e6) return;
    const float b0 = nx0 * cx0 + ny0 * cy0;
    const float b1 = nx1 * cx1 + ny1 * cy1;
    oy = (nx1 * b0 - nx0 * b1) / dety;
    ox = (ny1 * b0 - ny0 * b1) / detx;
}

// ransac_voting_kernel.cu:107-125, literal float32 order (sqrt and divide correctly rounded)
__device__ __forceinline__ bool inlier_literal(float cx, float cy, float nx, float ny, float hx, float hy,
                                               float thresh) {
#pragma clang fp contract(off)
    const float dx = hx - cx, dy = hy - cy;
    const float norm1 = __builtin_sqrtf(nx * nx + ny * ny);
    const float norm2 = __builtin_sqrtf(dx * dx + dy * dy);
    if (norm1 <= kF1e6 || norm2 <= kF1e6) return false;
    const float ang = (dx * nx + dy * ny) / (norm1 * norm2);
    return ang > thresh;
}

// the reference's norm1 (kernel.cu:119) in its operation order: the |u| < 1e-6 gate must fall exactly where the
// reference's falls, because a record that fails it is stored as a zero record by the matrix-pipe modes
__device__ __forceinline__ float norm1_literal(float nx, float ny) {
#pragma clang fp contract(off)
    return __builtin_sqrtf(nx * nx + ny * ny);
}

// Fast form of the same predicate.  With tau = sqrt(1 - thresh^2) / thresh (0 < thresh < 1) and d = h - c:
//     cos(angle(d, u)) > thresh   <=>   |d x u| < tau * (d . u)          (scale-invariant in |u|: no normalisation)
// The vote is taken on M = 2^k * u, with the power of two chosen PER RECORD so that max(|Mx|, |My|) lies in
// [2^60, 2^61) (vote_scale: an exact exponent shift, so the decision is that of u itself).  With T = tau * M the
// quantity  s = T.d - |M x d|  is then ~2^60 times the margin: any non-zero float32 margin is >= 1 in magnitude
// (a difference of two floats is a multiple of the smaller one's ulp, and at the threshold both terms are
// ~2^60 * tau * |d| >= 2^23 for any |d| >= 1e-9 px at thresh <= 0.9999), so a CLAMP output modifier turns s into
// exactly 1.0f (votes) or 0.0f (does not) -- the vote IS the arithmetic result: no compare, no carry, no scalar op.
// Because the scale follows the record, un-normalised fields (|u| from 1e-6 up to 2^60) behave like unit ones, and a
// term overflows float32 only for hypotheses farther than 2^67 px from the image (the reference's own 1e-6
// determinant gate keeps them below ~1e15 px); a NaN / Inf direction gives NaN margins, which the clamp turns into 0:
// no vote, as the reference's comparison with NaN decides.
// Zero directions (|u| < 1e-6, kernel.cu:121) are stored as zero records by K2 in fast mode and never vote; a
// hypothesis that sits exactly on a pixel gives s = 0 and does not vote either, as in the reference.
// The subtraction d = h - c is folded into per-pixel constants ("expanded form"):
//     cr = hx*My - hy*Mx - Ec,  Ec = cx*My - cy*Mx          s = hx*Tx + hy*Ty - Ed - |cr|,  Ed = cx*Tx + cy*Ty
// = 5 VALU ops (4 fma + 1 sub with |.|) + 1 add to accumulate.  Coordinates are taken relative to a per-image
// origin inside the object (the raster-median foreground pixel), which keeps the cancellation small: measured
// against float64 arithmetic on the benchmark data (tools/precision_study.py) this form decides 3e-8 of the pair
// tests differently, the un-expanded 7-op form 1e-8, and the reference's own float32 sqrt/divide order 6e-7 (the
// tan-based test resolves ~1e-7 rad at the threshold, cos-based float32 only ~1e-6: cos is flat where tan is steep).
//
// Records are (x, y, ux, uy) in both modes: the raw direction as the field holds it (fast mode: zeroed when
// |u| < 1e-6).  Everything that needs u itself -- hypothesis generation, the least-squares normals, the confidence
// epilogue -- reads it back bit for bit.
__device__ __forceinline__ float2 rec_dir(float4 q) { return make_float2(q.z, q.w); }
// the power of two that brings max(|ux|, |uy|) into [2^60, 2^61); any finite value for a zero / denormal direction
// (its products are zero whatever the scale) and for Inf / NaN (whose products are NaN whatever the scale)
__device__ __forceinline__ float vote_scale(float ux, float uy) {
    const float m = fmaxf(fabsf(ux), fabsf(uy));
    const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;   // biased exponent of m
    uint32_t f = 314u - e;                                   // 127 + 60 - (e - 127)
    f = f > 254u ? 254u : f;
    return __uint_as_float(f << 23);
}
// per-pixel constants as staged in LDS: a = (My, -Mx, -Ec, Tx) [ds_read_b128], b = (Ty, -Ed) [ds_read_b64]
__device__ __forceinline__ void make_pixrec(float4 q, float tau, float ox, float oy, float4& a, float2& b) {
    const float cx = q.x - ox, cy = q.y - oy;  // exact: integer pixel coordinates
    const float sc = vote_scale(q.z, q.w);
    const float My = q.w * sc, nMx = -q.z * sc;  // exact: a power-of-two scaling
    const float2 tq = make_float2(tau * -nMx, tau * My);  // T = tan(acos(thresh)) * M
    const float Ec = fmaf(cy, nMx, cx * My);
    const float Ed = fmaf(cy, tq.y, cx * tq.x);
    a = make_float4(My, nMx, -Ec, tq.x);
    b = make_float2(tq.y, -Ed);
}
__device__ __forceinline__ float vote_expanded(float4 a, float2 b, float hx, float hy) {
    const float cr = fmaf(hx, a.x, fmaf(hy, a.y, a.z));
    const float t = b.y - fabsf(cr);
    return __builtin_amdgcn_fmed3f(fmaf(hx, a.w, fmaf(hy, b.x, t)), 0.f, 1.f);  // clamp folds into the fma
}

// ---- bf16x3 operands of the matrix-pipe scoring kernel ------------------------------------------------------
// An fp32 value is the exact sum of three bf16 parts (round-to-nearest each time).  A product x*a keeps the six
// part pairs of relative weight >= 2^-16 (x0a0 x0a1 x1a0 x0a2 x2a0 x1a1; the dropped three are below one fp32
// rounding of the product), so the 3-term dot products of the vote, cr = hx*a + hy*b + c and dt = hx*e + hy*f + g,
// are ONE v_mfma_f32_32x32x16_bf16 each (K = 6 + 6 + 3, one slot spare), accumulated in fp32 by the matrix pipe.
// The K order is free as long as both operands agree; it is chosen so that every dword of a row holds the SAME part of
// the two coefficients -- one v_cvt_pk_bf16_f32 makes it (round 3; the former order needed a pack per dword):
//   A row (pixel)      k = 0..15 : a0 b0 | a1 b1 | a0 b0 | a2 b2 | a0 b0 | a1 b1 | c0 c1 | c2 spare
//   B column (hyp.)    k = 0..15 : x0 y0 | x0 y0 | x1 y1 | x0 y0 | x2 y2 | x1 y1 | 1  1  | 1  spare
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float x, uint32_t& p0, uint32_t& p1, uint32_t& p2) {  // raw bf16 bits
    const __bf16 h0 = (__bf16)x;
    const float r1 = x - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    const __bf16 h2 = (__bf16)r2;
    p0 = __builtin_bit_cast(unsigned short, h0);
    p1 = __builtin_bit_cast(unsigned short, h1);
    p2 = __builtin_bit_cast(unsigned short, h2);
}
__device__ __forceinline__ uint32_t pk(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }
// (bf16(a) | bf16(b) << 16), both rounded to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// the three parts of two values, pairwise packed: q0 = (a0 | b0), q1 = (a1 | b1), q2 = (a2 | b2)
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& q0, uint32_t& q1, uint32_t& q2) {
    q0 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(q0 << 16), rb = b - __uint_as_float(q0 & 0xFFFF0000u);   // exact
    q1 = pk_bf16(ra, rb);
    q2 = pk_bf16(ra - __uint_as_float(q1 << 16), rb - __uint_as_float(q1 & 0xFFFF0000u));
}
// the 16 K-slots of one operand row: (u, v, w) -> u0 v0 | u1 v1 | u0 v0 | u2 v2 | u0 v0 | u1 v1 | w0 w1 | w2 0
__device__ __forceinline__ void a_row(float u, float v, float w, uint4& lo, uint4& hi) {
    uint32_t q0, q1, q2, w0, w1, w2;
    split3_pair(u, v, q0, q1, q2);
    split3(w, w0, w1, w2);
    lo = make_uint4(q0, q1, q0, q2);
    hi = make_uint4(q0, q1, pk(w0, w1), pk(w2, 0u));
}
// the hypothesis side: (x, y) -> x0 y0 | x0 y0 | x1 y1 | x0 y0 | x2 y2 | x1 y1 | 1 1 | 1 0
__device__ __forceinline__ void b_col(float x, float y, uint4& lo, uint4& hi) {
    uint32_t q0, q1, q2;
    split3_pair(x, y, q0, q1, q2);
    const uint32_t one = 0x3F80u;
    lo = make_uint4(q0, q0, q1, q0);
    hi = make_uint4(q2, q1, pk(one, one), pk(one, 0u));
}

// ---- exact mode: the same two MFMAs, arranged so that the epilogue also sees how close every test is to the threshold
// Goal: inlier counts EQUAL to the reference's float32 kernel (kernel.cu:107-125) at matrix-pipe speed.  The reference
// decides  ang = fl(dot / (norm1 * norm2)) > thresh  with nine float32 roundings; against exact arithmetic on the same
// float32 inputs  |ang - cos(angle(d, u))| <= DELTA_LIT = 10 * 2^-24  (derivation: DESIGN.md section 4, "rounding band":
// 8 u from the operations themselves + 1 u from d = fl(h - c), rounded up), as long as no intermediate overflows -- which
// the two range gates below guarantee.  So the reference's decision can differ from the exact predicate only when
//     |m| <= |d| |u| DELTA_LIT / (sin t0 cos t0),   m = tau (d . u) - |d x u|,   tau = tan t0,  cos t0 = thresh
// and the matrix pipe's own evaluation of m (bf16x3 products, float32 accumulation, float32 staging of the per-pixel
// constants about the image origin o) is off by at most K_FAST * (|h - o| + |c - o|) |u|  (band_constant()).  With
//     |d| <= |h - o| + |c - o| <= (R + rho) (1 + r / rho),   R = |h - o|,  r = |c - o|,  any rho > 0
// the band separates into a per-hypothesis and a per-pixel factor, so both fold into the operands at no cost:
//     B column scaled by  s_j     = 0.9 / ((R_j + rho) kband)         (rounded DOWN to a bf16, so s * c parts stay exact)
//     A row    scaled by  sigma_i = (rho / (rho + r_i)) / |u_i|       (any float: the direction is normalised as well)
// give  |s_j sigma_i m| < 1  for every pair inside the band -- also as the matrix pipe computes it.  The two MFMAs return
//     dt' = s sigma dt,   cr' = s sigma cr        (round 3: a' = dt' - cr' and b' = dt' + cr', merged by a three-input minimum)
// so that  x = dt' - |cr'|  [one v_sub_f32 with a source modifier: the fast issue class, tools/ubench_issue.py] is >= 1 for a
// vote outside the band, <= -1 for a non-vote outside the band and strictly between for a test inside it.  Cells with
// min |x| >= 1 hold only tests on which the reference's arithmetic and exact arithmetic agree, and their votes are counted
// from the x's (saturating pknorm); a cell with min |x| < 1 is re-evaluated with inlier_literal() from the raw records and its
// x's are discarded.  = 2.5 VALU operations per test (1.5 in the approximate mode).  Alternatives measured in rounds 3 and 4:
// DESIGN.md section 4 (tools/ubench_exact.hip, tools/ubench_issue.py).
// Range gates: |h - o| >= 2^61 (or not finite) and |u| >= 2^61 would overflow the reference's squares -- such columns /
// rows are sent as zeros: x = 0 flags every cell they touch, which is then decided by the reference's arithmetic
// itself, whatever that does.  Zero records (padding, |u| < 1e-6) and NaN / Inf directions never vote in the reference;
// their dt' rows are zero except for the spare 16th K slot, A[15] = -4 against B[15] = 1: x = -4, no vote, no flag.
// (Exact mode compacts like literal mode: records keep the RAW direction even below the gate, because the reference's
// hypothesis generation reads it -- a 1e-7 direction paired with a 1e12 one has a determinant far above ITS gate.)
constexpr float BAND_TARGET = 0.9f;             // |s sigma m| inside the band (proof obligation: < 1 with the float roundings of the scales)
constexpr float BAND_FAR = 0x1p61f;             // beyond this the reference's float32 squares may overflow
// the length scale that splits |d| <= (R + rho)(1 + r / rho), R = |h - o|, r = |c - o|.  Round 4: the origin o is an estimate of
// the KEY-POINT (per image and key-point, band_origin() in the hypothesis kernel), no longer the image's median pixel: most
// hypotheses then have a small R and the bound is ~(rho + r) for them instead of ~3 |d|.  Simulated on the benchmark field
// (profiles/r04_band_origin_study.txt): 25 % fewer tests inside the band; 0.6 of the radius of a disk of tn pixels is the best
// rho for that origin (0.4 .. 0.8 within 2 %).  Any o and any rho > 0 keep the exactness argument: they only move the bound.
__device__ __forceinline__ float band_rho(int tn) {
    const float r = 0.6f * __builtin_sqrtf(0.3183f * (float)tn);
    return r < 8.f ? 8.f : r;
}
// origin of the exact mode's band per (image, key-point): int32 [b][vn][2] behind the ctrl rows (integer: pixel - origin is exact)
__device__ __forceinline__ int32_t* band_origin_ptr(const VoteParams& P, size_t bk) {
    return P.ctrl + (size_t)(P.b + 1) * CTRL_STRIDE + 2 * bk;
}
// the column of the point o + (hxo, hyo) at scale s (a bf16 value): s (hxo, hyo) as three bf16 parts each, s in the constant slots,
// 1 in the spare 16th slot (against which dead rows carry their -4); s <= 0 / NaN, or a point too far: the ZERO column -- x = 0
// for every live pixel, which the exact kernel flags (decided literally) and the culling kernel's disc test calls uncertain
__device__ __forceinline__ void b_col_scaled(float hxo, float hyo, float R, float s, uint4& lo, uint4& hi) {
    const uint32_t one = 0x3F80u;
    if (!(R < BAND_FAR) || !(s > 0.f)) {  // too far, Inf or NaN: x = 0 for every live pixel -> decided literally
        lo = make_uint4(0u, 0u, 0u, 0u);
        hi = make_uint4(0u, 0u, 0u, pk(0u, one));
        return;
    }
    uint32_t q0, q1, q2;
    split3_pair(hxo * s, hyo * s, q0, q1, q2);
    const uint32_t sb = __float_as_uint(s) >> 16;
    lo = make_uint4(q0, q0, q1, q0);
    hi = make_uint4(q2, q1, pk(sb, sb), pk(sb, one));
}
__device__ __forceinline__ float bf16_floor(float s) {   // round down to bf16: s * (c0 + c1 + c2) stays exact
    return __uint_as_float(__float_as_uint(s) & 0xFFFF0000u);
}
__device__ __forceinline__ void b_col_exact(float hxo, float hyo, float rho, float kband, uint4& lo, uint4& hi) {
    const float R = __builtin_sqrtf(fmaf(hxo, hxo, hyo * hyo)) * 1.000001f;
    b_col_scaled(hxo, hyo, R, bf16_floor(BAND_TARGET / ((R + rho) * kband)), lo, hi);
}
// per-pixel rows of dt' and cr', the direction normalised to |M| = sigma <= rho / (rho + r)
// (mu: an upper bound of the row's scale |M| <= rho / (rho + r) -- what the disc test of the culling kernel needs per pixel;
// 1 for dead and zero rows, whose x does not depend on it)
__device__ __forceinline__ void a_rows_exact(float4 q, float tau, float ox, float oy, float rho, uint4& alo, uint4& ahi,
                                             uint4& blo, uint4& bhi, float& mu) {
    const uint32_t never = 0xC080u;  // bf16 -4 in the spare slot of the dt' row: x = -4
    alo = ahi = blo = bhi = make_uint4(0u, 0u, 0u, 0u);
    mu = 1.f;
    const float m = fmaxf(fabsf(q.z), fabsf(q.w));
    const uint32_t e = (__float_as_uint(m) >> 23) & 0xFFu;
    const bool finite = fabsf(q.z) <= 3.4028235e38f && fabsf(q.w) <= 3.4028235e38f;  // false for NaN and Inf
    // the reference never votes for: padding (zero record), NaN / Inf directions, and |u| below its 1e-6 gate -- decided by
    // the gate's own arithmetic (norm1_literal), which only directions within a factor two of the gate need: records keep
    // the RAW direction in exact mode, as hypothesis generation needs it
    bool dead = !(m > 0.f) || !finite;
    const bool near_gate = !dead && m <= 2.0e-6f;  // (m > 2e-6 implies norm1 > 1e-6 in any rounding)
    if (__ballot(near_gate)) {  // wave-uniform and almost never taken; the empty asm keeps the correctly-rounded sqrt (~25
        float nz = q.z;         // instructions) from being speculated out of the branch, where every pixel would pay for it
        asm volatile("" : "+v"(nz));
        if (near_gate) dead = norm1_literal(nz, q.w) <= kF1e6;
    }
    if (dead) {
        ahi.w = pk(0u, never);  // dt' = -4, cr' = 0: x = -4, no vote and no flag
        return;
    }
    if (e >= 127u + 61u) return;    // finite but >= 2^61: the reference's nx * nx may overflow -- zero rows: decided literally
    const float pre = __uint_as_float((254u - e) << 23);       // 2^(127 - e): max(|ux|, |uy|) -> [1, 2)   (e = 0: denormal, 2^127)
    const float u1x = q.z * pre, u1y = q.w * pre;               // exact
    const float g = __builtin_amdgcn_rsqf(fmaf(u1y, u1y, u1x * u1x));
    const float cx = q.x - ox, cy = q.y - oy;                   // exact: integer pixel coordinates
    const float r = __builtin_amdgcn_sqrtf(fmaf(cy, cy, cx * cx));            // (v_sqrt_f32 / v_rcp_f32: 1 ulp each --
    const float sig = rho * __builtin_amdgcn_rcpf(rho + r);
    const float gs = g * sig * 0.9997f;                                       //  an upper bound is all that is needed)
    // |M| = |u1| gs <= rho / (rho + r)  (and < sig as computed: the 3e-4 of slack is far above the roundings of g, sig and M)
    mu = sig;
    const float Mx = u1x * gs, My = u1y * gs;
    const float Tx = tau * Mx, Ty = tau * My;
    const float Ec = fmaf(cx, My, -cy * Mx);                    // cr = hx My - hy Mx - Ec
    const float Ed = fmaf(cx, Tx, cy * Ty);                     // dt = hx Tx + hy Ty - Ed
    a_row(Tx, Ty, -Ed, alo, ahi);                               // dt' rows
    a_row(My, -Mx, -Ec, blo, bhi);                              // cr' rows
}

// v of lane (l ^ M), M a power of two below 64, without the LDS crossbar's address operand: DPP for 1, 2 and 8 (quad_perm, row_ror:8),
// gfx950's row / half-wave swaps for 16 and 32, ds_swizzle's bit mode for 4
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int m) {
    if (m == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1, 0, 3, 2]
    if (m == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2, 3, 0, 1]
    if (m == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);               // and 0x1F, or 0, xor 4
    if (m == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
    const int lane = (int)(threadIdx.x & 63);
    if (m == 16) {   // rows of 16 lanes: r[0] = (R0, R0, R2, R2), r[1] = (R1, R1, R3, R3)
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0] = (lo, lo), r[1] = (hi, hi)
    return (lane & 32) ? r[0] : r[1];
}

// Wave reductions as a butterfly over lane ^ 1, 2, 4, 8, 16, 32: EVERY lane ends with the result, and no step goes through the LDS
// crossbar with an address operand (round 6: the __shfl_down forms -- 6 ds_bpermute per 32-bit word, each a dependent LDS round trip
// of ~100-200 cycles -- cost select_refine_kernel's five float64 sums and its 64-bit arg-max key 72 of them on its critical path).
// (float64 sums come out in a different order than a sequential loop's: the refinement's tolerance is 1e-3 px, its sums agree to 1e-15)
__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += (int)lane_xor((uint32_t)v, m);
    return v;
}
__device__ __forceinline__ double wave_reduce_add(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned long long o = (unsigned long long)lane_xor((uint32_t)b, m) | ((unsigned long long)lane_xor((uint32_t)(b >> 32), m) << 32);
        v += __longlong_as_double((long long)o);
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_reduce_max(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long t = (unsigned long long)lane_xor((uint32_t)v, m) | ((unsigned long long)lane_xor((uint32_t)(v >> 32), m) << 32);
        v = t > v ? t : v;
    }
    return v;
}

// Workgroup barrier for data shared through LDS ONLY.  __syncthreads() fences every address space: before the barrier each
// wave waits for ALL its outstanding global operations (s_waitcnt vmcnt(0)) -- in the scoring kernels that means the count
// atomics of the previous work item.  The scoring kernels' barriers order nothing but the LDS tiles / lists, so they wait for
// the LDS and scalar counters only.  (Round 4 measured no difference at the benchmark shape, r04c17; a one-item-ahead prefetch
// of records / B columns / hypotheses built on it hid 1 500 of an item's 4 500 staging cycles and lost them again in the
// loop and the re-evaluation, r04c18: the kernel is throughput-bound, a workgroup's waits are filled by the other two.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The vector field (and the class logits of the fused arg-max entry) may be float32, float16 or bfloat16 -- what a
// backbone under autocast emits: elements are widened to float32 where they are read (both conversions are exact), so the
// result is that of the float32 path on `field.float()` without the copy (786 MB written per batch of 32 otherwise).
enum { VT_F32 = 0, VT_F16 = 1, VT_BF16 = 2 };
template <int VT>
__device__ __forceinline__ float ld_elem(const void* base, int64_t off) {
    if (VT == VT_F16) return (float)reinterpret_cast<const _Float16*>(base)[off];
    if (VT == VT_BF16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[off] << 16);
    return reinterpret_cast<const float*>(base)[off];
}
__device__ __forceinline__ float ld_elem_rt(int vt, const void* base, int64_t off) {  // run-time type (cold paths)
    return vt == VT_F16 ? ld_elem<VT_F16>(base, off) : vt == VT_BF16 ? ld_elem<VT_BF16>(base, off) : ld_elem<VT_F32>(base, off);
}

// ------------------------------------------------------------------------------------------------------------
// work items of the scoring launches (planned by K3, consumed by K4)
// ------------------------------------------------------------------------------------------------------------
// item descriptor (image, key-point | culled << 16, chunk group, hypothesis slice): the scoring kernels decode it with these
constexpr int ITEM_CULL_SHIFT = 16;
__device__ __forceinline__ int item_kp(int y) { return y & 0xFFFF; }
__device__ __forceinline__ bool item_culled(int y) { return (y >> ITEM_CULL_SHIFT) != 0; }

constexpr int CULL_NPX = 256;              // pixels per work item of the culling body: 8 pixel tiles, list entries are one byte
constexpr int CULL_HN = 1024;              // hypotheses per key-point (hn_pad) of the layouts that can cull: one hypothesis slice, 32 tiles,
                                           // four sort keys per thread of a 256-thread block
constexpr int CULL_DEAD = 8 * TILE_U4_;    // uint4 index of the dead A row behind the item's 8 tiles (x = -4: no vote, no flag)
constexpr int NCAND = 8, KP_MAX = 32;      // candidate intersections per key-point of the origin estimate; key-points it handles

// which key-points of which image are disc-culled: int32 [b][vn] behind the band origins (the scoring kernels' flag travels in the
// item descriptors; this copy is for the epilogues that read a finished workspace: band_margin_kernel, the debug views)
__device__ __forceinline__ int32_t* kp_cull_ptr(const VoteParams& P, size_t bk) {
    return P.ctrl + (size_t)(P.b + 1) * CTRL_STRIDE + 2 * (size_t)P.b * P.vn + bk;
}

// Work items of a scoring launch owned by this workgroup.  Workgroups go to the 8 XCDs round-robin by linear id
// (observed, for speed only: MI355X_MICROARCH.md "Workgroup dispatch"), and K3 plans the items in (image, key-point,
// pixel group) order, so with score_xcd every XCD takes one contiguous eighth of the list: the B-operand columns and
// records of an (image, key-point) are then fetched through ONE L2 instead of once per XCD.  Any placement gives the
// same result -- the mapping is a permutation of items over workgroups.
struct ItemRange { int first, end, step; };
// CONTIG: a workgroup takes a contiguous run of its XCD's eighth instead of a strided sample -- consecutive items are
// consecutive pixel groups of one (image, key-point), so the run keeps its B columns, hypotheses and vote counters (round 4)
template <bool CONTIG = false>
__device__ __forceinline__ ItemRange my_items(const VoteParams& P, int total) {
    if (P.score_xcd && (gridDim.x & 7u) == 0 && gridDim.x >= 8) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, n = (int)(gridDim.x >> 3);
        const int lo = (int)((long long)total * x >> 3), hi = (int)((long long)total * (x + 1) >> 3);
        if (CONTIG) return {lo + (int)((long long)(hi - lo) * j / n), lo + (int)((long long)(hi - lo) * (j + 1) / n), 1};
        return {lo + j, hi, n};
    }
    if (CONTIG)
        return {(int)((long long)total * blockIdx.x / gridDim.x), (int)((long long)total * (blockIdx.x + 1) / gridDim.x), 1};
    return {(int)blockIdx.x, total, (int)gridDim.x};
}

constexpr int TILE_U4 = 128;  // uint4 per 32-pixel tile: four blocks of 32 x 16 bytes -- first operand K slots 0..7 of rows 0..31, its K slots
                              // 8..15, then the second operand's two halves.  (Round 6: until then a row's two 16-byte halves lay side by
                              // side, so the 16 lanes that one ds_read_b128 cycle serves -- sixteen rows, ONE half -- used only the even or only
                              // the odd 16-byte slots of the 256-byte bank row: two-way conflicts on every dense read, four- to five-way on the
                              // culling kernel's gathered ones.  Now slot = row mod 16: dense reads are conflict-free, gathered ones meet
                              // sixteen slots instead of eight.)
static_assert(TILE_U4_ == TILE_U4, "the disc-culling body's row addresses (CULL_DEAD, tile = pixel >> 5, row = pixel & 31) follow TILE_U4");

// v + (the other half-wave's v): lanes l and l ^ 32 hold different pixel rows of one hypothesis column.  gfx950's
// v_permlane32_swap exchanges the upper row of one operand with the lower row of the other in the VALU -- no trip through
// the LDS crossbar as __shfl_xor (ds_bpermute) takes, eight times per work item.
__device__ __forceinline__ int half_wave_sum(int v) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(r[0] + r[1]);
}
// two hypothesis tiles at once: returns (a + a's other half-wave) in lanes 0..31 and (b + b's other half-wave) in lanes 32..63 --
// one swap and one add for two columns sets, and the decode / atomic that follow run on 64 useful lanes instead of 32.
// Works on the WRAPPED accumulators (their encodings are linear mod 2^32).
__device__ __forceinline__ unsigned half_wave_sum2(unsigned a, unsigned b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // r[0] = (a.lo | b.lo), r[1] = (a.hi | b.hi)
    return r[0] + r[1];
}

constexpr float BAND_CLEAN = 1.0f;     // a cell whose minimum |a'|, |b'| reaches this holds no test inside the band
constexpr int VOTE_WRAP = 512;  // vote8 accumulators (approximate mode) hold their count mod 512

// ------------------------------------------------------------------------------------------------------------
// host side shared by the translation units (defined in vote_host.hip unless noted)
// ------------------------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Tuning knobs.  The environment is read ONCE, at the first call into the library, never on
// the launch path; pvnet_vote_tuning_reload() (host-only, for tests and the tuning tools) reads it again.  A value of
// -1 means "not set: use the shape-dependent default".
struct Tuning {
    int score_mode;     // PVNET_SCORE_MODE        1: matrix-pipe scoring in fast mode, 0: the 6-op VALU kernel
    int wgs_per_cu;     // PVNET_SCORE_WGS_PER_CU  scoring workgroups launched per CU (0: one per work item; -1 (default): 8, and 12 for
                        //                         an exact-mode batch alone at 8 tiles per wave -- four resident per CU, three rounds)
    int hpl;            // PVNET_SCORE_HPL         hypotheses per lane of the VALU kernel / MFMA tiles per wave
    int chunk;          // PVNET_SCORE_CHUNK       pixels per count row
    int compact_kg;     // PVNET_COMPACT_KG        key-points per compaction block
    int score_xcd;      // PVNET_SCORE_XCD         1: contiguous eighths of the work-item list per XCD (L2 affinity)
    int score_lds_kb;   // PVNET_SCORE_LDS_KB      experiment: pad the matrix-pipe kernel's dynamic LDS to this many KB, which
                        //                         caps its resident workgroups per CU (160 KB / value) and leaves registers
                        //                         for other streams' small stages; 0 = no padding
    int score_atomic;   // PVNET_SCORE_ATOMIC      1 (default): K4 adds its counts into `counts` with integer atomics;
                        //                         0: per-chunk uint16 count rows (`partial`) summed by K5
    int score_acc;      // PVNET_SCORE_ACC         exact mode, 8 tiles per wave: accumulator pairs of the scoring loop (2: MFMAs of the
                        //                         next step issued around this step's votes; 1: one pair, 32 VGPRs fewer;
                        //                         -1 (default): 1 -- in 136 VGPRs for calls flagged PVNET_F_CONCURRENT (three waves per
                        //                         SIMD), in 128 for a batch alone (four))
    int score_runs;     // PVNET_SCORE_RUNS        exact mode, 8 tiles per wave: 1 = contiguous item runs per workgroup (B columns, hypotheses and
                        //                         vote counters kept while the (image, key-point) stays), 0 = strided items;
                        //                         -1 (default): runs for calls flagged PVNET_F_CONCURRENT
    int score_cull;     // PVNET_SCORE_CULL        exact mode, 8 tiles per wave, 256-pixel items, hn_pad = 1024: disc culling (hypotheses
                        //                         sorted along a Hilbert curve, per-pixel certainty against every tile's disc, uncertain
                        //                         pixels gathered: the culling body of score_exact_kernel_both) of 2 = the images K3 selects from the
                        //                         spread of their candidate intersections (the default: PVNET_CULL_DEFAULT), 1 = every
                        //                         key-point (tests, probes), 0 = none (the layout then has no culling buffers)
    int cull_q_milli;   // PVNET_CULL_Q_MILLI      the selection threshold of 2, in thousandths (kp_preamble; profiles/r06y_cull_crossover.txt)
    int exact_fold;     // PVNET_EXACT_FOLD        exact mode: -1 (default) = by threshold, 0 = one cell per work item and
                        //                         hypothesis, 1 = one cell per pixel tile (band_fold1())
    int dev_stages;     // PVNET_DEV_STAGES        development aid: bit mask of the stages to launch
    int cus;            // compute units of the device (all GPUs of a node are the same part)
};
Tuning& tuning();
void load_tuning(Tuning& t);
float band_constant(float thresh);
int layout_fingerprint(const PvnetVoteLayout& L);
int fill_params(VoteParams& P, const void* mask, int mask_dtype, const int64_t* ms, const float* vertex, const int64_t* vs, int b,
                int h, int w, int vn, int hn, float thresh, int min_num, int max_num, uint64_t seed, int image_base,
                const int32_t* idxs, uint32_t flags, float* out, int32_t* status, void* ws, size_t ws_bytes);

#define PV_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return (int)e_;               \
    } while (0)
#define PV_HIP(x)                                           \
    do {                                                    \
        hipError_t e_ = (x);                                \
        if (e_ != hipSuccess) return (int)e_;               \
    } while (0)

// one launcher per stage, each in the translation unit of its kernels (0 or a PVNET_E_* / hipError_t code; the launch error itself is
// picked up by the caller's PV_LAUNCH_CHECK)
int launch_mask_bits(const VoteParams& P, hipStream_t s);                                                  // k1_mask.hip
int launch_compact(const VoteParams& P, hipStream_t s, bool literal, int kg);                              // k2_compact.hip
int launch_hypotheses(const VoteParams& P, hipStream_t s, bool literal);                                   // k3_hypotheses.hip
int launch_score_valu(const VoteParams& P, dim3 grid, hipStream_t s, bool literal);                        // k4_score_valu.hip
int launch_score_mfma(const VoteParams& P, dim3 grid, size_t lds, hipStream_t s, bool timed);              // k4_score_mfma.hip
int launch_score_exact(const VoteParams& P, dim3 grid, size_t lds, hipStream_t s, bool timed, bool one_acc, bool runs);  // k4_score_exact.hip
int launch_score_both(const VoteParams& P, dim3 grid, hipStream_t s, bool timed, bool runs);               // k4_score_cull.hip
int launch_select_refine(const VoteParams& P, hipStream_t s, bool literal);                                // k5_refine.hip
int launch_ts_collect(const unsigned long long* stamps, int grid, unsigned long long* acc, int first, hipStream_t s);  // epilogues.hip

}  // namespace pvd
