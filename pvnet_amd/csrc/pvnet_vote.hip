// pvnet_vote.hip -- hand-written gfx950 (MI355X / CDNA4) implementation of PVNet's RANSAC voting layer.
//
// Path replaced (reference tree): lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:514-598
// (ransac_voting_layer_v3) together with the two CUDA kernels it drives,
// src/ransac_voting_kernel.cu:11-49 (generate_hypothesis) and :88-126 (voting_for_hypothesis).
// Not a translation: the reference runs a Python loop per image with ~40 torch launches, materialises an
// [hn,vn,tn] uint8 inlier tensor and syncs with the host 6-8 times per image.  Here a whole batch is six
// launches on the caller's stream, nothing of size hn*tn ever touches HBM, and there is no host sync,
// no allocation and no memset (DESIGN.md has the design history and the measurements behind each choice).
//
// Stages (one launch each, all images of the batch at once):
//   K1 mask_bits      mask (any int dtype / f32, any strides) -> 1 bit per pixel + per-segment counts  [HBM read]
//                     + per segment the cumulative histogram of the thinning decisions (Bernoulli(k / 1024) with
//                     k = ceil(1024 max_num / tn0), decided on the device once tn0 is known)
//   K2 compact        thins its own segment when tn0 > max_num, then order-preserving (raster) compaction: segment counts + wave scan of word popcounts give
//                     every kept pixel its slot; one thread per kept pixel gathers its vn direction vectors
//                     straight from the strided field (planar in practice -> consecutive lanes read
//                     consecutive addresses) and writes ONE float4 record per (pixel, key-point):
//                     (x, y, ux, uy), the raw direction                                                 [HBM read]
//   K3 hypotheses     one thread per (image, kp, h): two pixel draws (counter RNG or caller idxs), 2x2 solve in
//                     the reference's float32 order; also writes each hypothesis as a bf16x3 MFMA operand column and
//                     zeroes its inlier count; one extra block per image plans the scoring work items
//   K4 score          DOMINANT.  The vote is two 3-term fp32 dot products and a compare; every operand is split into
//                     three bf16 parts, so each dot product is ONE v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//                     EXACT mode (default, score_exact_body): the two MFMAs return dt and cr in units of the float32 rounding
//                     band of the reference's test; the lane that owns the hypothesis takes x = dt - |cr| (>= 1 for a vote
//                     outside the band, <= -1 for a non-vote outside it), keeps min |x| and counts the votes from packed-norm
//                     halves: 2 MFMAs + 2.5 VALU ops per test, ordered for the SIMD's two issue ports (vote_subs / vote_slow_*);
//                     cells that hold a test inside the band are re-evaluated with the reference's own arithmetic
//                     (inlier_literal) from the raw records, so every inlier count EQUALS the reference kernel's.
//                     APPROX mode (PVNET_F_APPROX, score_mfma_kernel): t = clamp(dt - |cr|) on 2^60-scaled records,
//                     2 MFMAs + 1.5 VALU ops per test (vote8), no re-evaluation: counts within a few votes.
//                     Records are expanded and staged in LDS once per work item.
//                     Literal mode (score_kernel<HPL,true>): "lane owns hypotheses" on the VALU in the
//                     reference's float32 operation order for every pair, bit-exact with the reference's kernels.
//                     Work items go to a persistent grid XCD by XCD (one contiguous eighth of the list per XCD: the
//                     operands of an (image, key-point) pass through one L2); counts are added into
//                     counts[b][vn][hn] with integer atomics (zeroed by K3; order-independent, deterministic).
//   K5 select+refine  arg-max over the counts with first-index tie-break (wave shuffles), recomputes the
//                     winner's inliers and solves the 2x2 normal equations, accumulated in float64 centred on
//                     the winner (the reference's un-centred float32 sums are ~2e-3 px noisy).
//
// Wave size is 64 everywhere (ballots are 64-bit).  Build: hipcc --offload-arch=gfx950 (see build.py).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "pvnet_rng.h"
#include "pvnet_vote.h"

namespace {

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
#define PVNET_CULL_Q_MILLI 500 // a key-point votes for disc culling when its candidate intersections spread over <= 0.5 rho tan(theta0)
                               // (profiles/r06k_cull_crossover.txt: culling wins up to a median q of 0.5 - 0.65)
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

__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_add(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_reduce_max(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_down(v, o, 64);
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
// K1: mask -> bit mask + foreground count                     (ransac_voting_gpu.py:527-528)
// ------------------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ bool load_fg(const void* m, int64_t off) {
    if (DT == PVNET_MASK_U8) return reinterpret_cast<const uint8_t*>(m)[off] != 0;
    if (DT == PVNET_MASK_I16) return (reinterpret_cast<const uint16_t*>(m)[off] & 0xFFu) != 0;
    if (DT == PVNET_MASK_I32) return (reinterpret_cast<const uint32_t*>(m)[off] & 0xFFu) != 0;
    if (DT == PVNET_MASK_I64)  // read once, never again: non-temporal (keeps the 78 MB of a batch out of L2 / MALL)
        return (__builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(m) + off) & 0xFFull) != 0;
    const float v = reinterpret_cast<const float*>(m)[off];  // torch .byte() of a float: truncate, wrap
    return (static_cast<long long>(v) & 0xFF) != 0;
}

// wave 0 of a K1 workgroup: inclusive prefix over the segment's histogram of thinning bins -- cum[k - 1] = pixels kept at threshold k
__device__ __forceinline__ void thin_hist_prefix(const VoteParams& P, int bi, const int* s_hist, int lane) {
    constexpr int PER = THIN_BINS / 64;  // consecutive bins per lane
    int h[PER], mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        mine += s_hist[PER * lane + i];
        h[i] = mine;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    const int e = incl - mine;
    uint16_t* dst = P.cum + ((size_t)bi * P.nseg + blockIdx.x) * THIN_BINS + PER * lane;
#pragma unroll
    for (int i = 0; i < PER; i += 2)
        *reinterpret_cast<uint32_t*>(dst + i) = (uint32_t)(e + h[i]) | ((uint32_t)(e + h[i + 1]) << 16);
}

template <int DT>
__global__ __launch_bounds__(64 * K1_WAVES) void mask_bits_kernel(VoteParams P) {
    PVNET_SPARE_VGPRS(39);
    small_stage_prio();
    const int bi = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Round 5 (tools/ubench_hbm_read.hip, profiles/r05_ubench_mask_kernel.txt): load i of the workgroup's waves covers ONE contiguous
    // 512 * K1_WAVES bytes -- wave w takes the segment's words w, w + K1_WAVES, ... -- instead of every wave walking its own 4 KB
    // (-1.5 us), and the 64 bit words of the segment leave as one coalesced 512-byte store by wave 0 behind the barrier the count
    // needs anyway, instead of as 64 one-lane stores (-2 us).
    const int seg_word0 = blockIdx.x * SEG_WORDS;
    auto word_of = [&](int i) { return seg_word0 + wave + i * K1_WAVES; };
    bool f[K1_WORDS_PER_WAVE];
#pragma unroll
    for (int i = 0; i < K1_WORDS_PER_WAVE; ++i) {
        const int p = word_of(i) * 64 + lane;
        bool v = false;
        if (p < P.npix) {
            int64_t off;
            if (P.mask_linear) {
                off = (int64_t)bi * P.ms0 + p;
            } else {
                const int y = p / P.w, x = p - y * P.w;
                off = (int64_t)bi * P.ms0 + (int64_t)y * P.ms1 + (int64_t)x * P.ms2;
            }
            if (DT == PVNET_MASK_LOGITS_F32) {  // fused torch.argmax(seg_pred, 1) (tools/demo.py:52): first maximum wins
                float best = ld_elem_rt(P.logits_type, P.mask, off);
                int arg = 0;
                for (int c = 1; c < P.num_classes; ++c) {
                    const float x = ld_elem_rt(P.logits_type, P.mask, off + (int64_t)c * P.ms_c);
                    if (x > best) { best = x; arg = c; }
                }
                v = (arg & 0xFF) != 0;  // then .byte() != 0 (ransac_voting_gpu.py:527)
            } else {
                v = load_fg<DT>(P.mask, off);
            }
        }
        f[i] = v;
    }
    int cnt = 0;
    __shared__ unsigned long long s_words[SEG_WORDS];
#pragma unroll
    for (int i = 0; i < K1_WORDS_PER_WAVE; ++i) {
        const unsigned long long m = __ballot(f[i]);
        if (lane == 0) s_words[wave + i * K1_WAVES] = m;
        cnt += __popcll(m);
    }
    __shared__ int s_cnt[K1_WAVES];
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (wave == 0 && seg_word0 + lane < P.words) P.bits[(size_t)bi * P.words + seg_word0 + lane] = s_words[lane];
    int t = 0;
#pragma unroll
    for (int i = 0; i < K1_WAVES; ++i) t += s_cnt[i];
    if (threadIdx.x == 0) P.seg0[bi * P.nseg + blockIdx.x] = t;  // foreground pixels of this 4096-pixel segment (tn0 = their sum)

    // Thinning (ransac_voting_gpu.py:537-540) keeps a pixel when the bin of its random word (pvnet_thin_bin) is below
    // K = pvnet_thin_bins_kept(max_num, tn0) -- but tn0 is only known when every segment has been counted.  So a segment
    // WITH foreground (one in ten) also counts how many of its pixels EVERY possible k would keep (a cumulative histogram
    // of those bits, 2 KB), and the compaction kernel, which sums the segment counts anyway, picks its column: no separate
    // thinning launch.  Only when thinning can happen at all (max_num < h*w: P.cum is set).
    if (P.cum == nullptr || t == 0) return;  // block-uniform
    __shared__ int s_hist[THIN_BINS];
    for (int i = threadIdx.x; i < THIN_BINS; i += 64 * K1_WAVES) s_hist[i] = 0;
    __syncthreads();
    const uint32_t key = pvnet_rng_key(P.seed, PVNET_TAG_SUB, (uint32_t)(P.image_base + bi));
#pragma unroll
    for (int i = 0; i < K1_WORDS_PER_WAVE; ++i)
        if (f[i]) atomicAdd(&s_hist[pvnet_thin_bin(pvnet_rng_at(key, (uint32_t)(word_of(i) * 64 + lane)))], 1);
    __syncthreads();
    if (wave == 0) thin_hist_prefix(P, bi, s_hist, lane);
}

// K1 for contiguous, 16-byte aligned int64 masks -- what torch.argmax delivers (tools/demo.py:52) and what the benchmark times: ONE
// 16-byte load brings two pixels per lane (a bare read of this size takes 15.4 us that way against 18.6 with 8-byte loads,
// profiles/r05_ubench_mask_kernel.txt).  A wave's ballots then hold the even and the odd pixels of a 128-pixel double word; wave 0
// interleaves them into the segment's 64 bit words behind the barrier and stores those as one coalesced 512 bytes.  Same outputs
// as mask_bits_kernel<PVNET_MASK_I64>: bit words, segment count, thinning histogram.
__device__ __forceinline__ unsigned long long spread_bits(uint32_t v) {   // bit k of v -> bit 2 k
    unsigned long long x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__global__ __launch_bounds__(64 * K1_WAVES) void mask_bits_pair_kernel(VoteParams P) {
    PVNET_SPARE_VGPRS(39);
    small_stage_prio();
    constexpr int DW = SEG_WORDS / 2;     // double words (128 pixels) per segment
    constexpr int DPW = DW / K1_WAVES;    // per wave
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int bi = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg_word0 = blockIdx.x * SEG_WORDS;
    const u64x2* base = reinterpret_cast<const u64x2*>(reinterpret_cast<const uint64_t*>(P.mask) + (int64_t)bi * P.ms0);
    auto pixel_of = [&](int i) { return (blockIdx.x * DW + wave + i * K1_WAVES) * 128 + 2 * lane; };   // (load i of the waves: 1 KB * K1_WAVES contiguous)
    bool f0[DPW], f1[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int p = pixel_of(i);
        f0[i] = f1[i] = false;
        if (p < P.npix) {   // (npix is even here: p + 1 < npix too)
            const u64x2 v = __builtin_nontemporal_load(base + (p >> 1));
            f0[i] = (v.x & 0xFFull) != 0;   // .byte() != 0 (ransac_voting_gpu.py:527)
            f1[i] = (v.y & 0xFFull) != 0;
        }
    }
    int cnt = 0;
    __shared__ unsigned long long s_even[DW], s_odd[DW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const unsigned long long me = __ballot(f0[i]), mo = __ballot(f1[i]);
        if (lane == 0) {
            s_even[wave + i * K1_WAVES] = me;
            s_odd[wave + i * K1_WAVES] = mo;
        }
        cnt += __popcll(me) + __popcll(mo);
    }
    __shared__ int s_cnt[K1_WAVES];
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (wave == 0 && seg_word0 + lane < P.words) {   // word `lane` of the segment: half of a double word, even and odd pixels interleaved
        const unsigned long long me = s_even[lane >> 1], mo = s_odd[lane >> 1];
        const uint32_t e32 = (lane & 1) ? (uint32_t)(me >> 32) : (uint32_t)me, o32 = (lane & 1) ? (uint32_t)(mo >> 32) : (uint32_t)mo;
        P.bits[(size_t)bi * P.words + seg_word0 + lane] = spread_bits(e32) | (spread_bits(o32) << 1);
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < K1_WAVES; ++i) t += s_cnt[i];
    if (threadIdx.x == 0) P.seg0[bi * P.nseg + blockIdx.x] = t;
    if (P.cum == nullptr || t == 0) return;  // block-uniform: the thinning histogram, as in mask_bits_kernel
    __shared__ int s_hist[THIN_BINS];
    for (int i = threadIdx.x; i < THIN_BINS; i += 64 * K1_WAVES) s_hist[i] = 0;
    __syncthreads();
    const uint32_t key = pvnet_rng_key(P.seed, PVNET_TAG_SUB, (uint32_t)(P.image_base + bi));
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        if (f0[i]) atomicAdd(&s_hist[pvnet_thin_bin(pvnet_rng_at(key, (uint32_t)pixel_of(i)))], 1);
        if (f1[i]) atomicAdd(&s_hist[pvnet_thin_bin(pvnet_rng_at(key, (uint32_t)pixel_of(i) + 1u))], 1);
    }
    __syncthreads();
    if (wave == 0) thin_hist_prefix(P, bi, s_hist, lane);
}

// ------------------------------------------------------------------------------------------------------------
// K2: order-preserving compaction + direction gather          (ransac_voting_gpu.py:542-546)
// ------------------------------------------------------------------------------------------------------------
template <bool LITERAL, int K2_KG, int VT>  // K2_KG key-points per block: grid.z = ceil(vn / K2_KG); VT: field element type
__global__ __launch_bounds__(256) void compact_kernel(VoteParams P) {
    if (K2_KG == 1) PVNET_SPARE_VGPRS(39); else if (K2_KG <= 3) PVNET_SPARE_VGPRS(47); else PVNET_SPARE_VGPRS(87);
    small_stage_prio();
    const int bi = blockIdx.y;
    const int w0 = blockIdx.x * K2_WORDS_PER_BLOCK;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t* bw = P.bits + (size_t)bi * P.words;

    __shared__ int s_red[4];
    __shared__ int s_tot[4];
    __shared__ int s_woff[K2_WORDS_PER_BLOCK];
    __shared__ uint64_t s_word[K2_WORDS_PER_BLOCK];
    __shared__ uint16_t s_piece[4 * K2_WORDS_PER_BLOCK];
    __shared__ int s_total;

    // the image's foreground count (tn0) and the pixels kept before this block = sums over the segment counts
    // (<= a few hundred ints)
    const int32_t* sg = P.seg0 + bi * P.nseg;
    const bool last = blockIdx.x == gridDim.x - 1;
    if (!last && sg[blockIdx.x] == 0) return;  // block-uniform: most of the image is background
    int part = 0, tot = 0;
    for (int j = threadIdx.x; j < P.nseg; j += 256) {
        const int c = sg[j];
        tot += c;
        part += j < (int)blockIdx.x ? c : 0;
    }
    part = wave_reduce_add(part);
    tot = wave_reduce_add(tot);
    if (lane == 0) {
        s_red[wave] = part;
        s_tot[wave] = tot;
    }
    auto scan_words = [&](unsigned long long wd) {  // wave 0: exclusive scan of this block's 64 word popcounts
        const int c = __popcll(wd);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        s_word[lane] = wd;
        s_woff[lane] = incl - c;
        if (lane == 63) s_total = incl;
    };
    if (wave == 0) scan_words((w0 + lane < P.words) ? bw[w0 + lane] : 0ull);
    __syncthreads();
    const int tn0 = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    int base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tn0 > P.max_num) {
        // Thinning (block-uniform, rare: objects larger than max_num pixels, or the evaluation call site's max_num = 100):
        // keep a pixel when the bin of its random word is below k (oracle: subsample_threshold).  The pixels kept in
        // earlier segments are column k - 1 of their cumulative histograms (K1); this segment's words are filtered here,
        // 16 bits per thread.
        const int k = pvnet_thin_bins_kept(P.max_num, tn0);  // 0 .. PVNET_THIN_LAST + 1
        int part2 = 0;
        if (k > 0)
            for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256)
                if (sg[j] > 0) part2 += P.cum[((size_t)bi * P.nseg + j) * THIN_BINS + k - 1];
        part2 = wave_reduce_add(part2);
        const int q = threadIdx.x & 3;
        unsigned todo = (unsigned)(s_word[threadIdx.x >> 2] >> (16 * q)) & 0xFFFFu, kept = 0;
        const uint32_t key = pvnet_rng_key(P.seed, PVNET_TAG_SUB, (uint32_t)(P.image_base + bi));
        const uint32_t p0 = (uint32_t)((w0 + (threadIdx.x >> 2)) * 64 + 16 * q);
        while (todo) {
            const int bpos = __ffs((int)todo) - 1;
            todo &= todo - 1;
            if (pvnet_thin_bin(pvnet_rng_at(key, p0 + (uint32_t)bpos)) < k) kept |= 1u << bpos;
        }
        __syncthreads();  // every read of s_word / s_red above has been performed
        s_piece[threadIdx.x] = (uint16_t)kept;
        if (lane == 0) s_red[wave] = part2;
        __syncthreads();
        if (wave == 0)
            scan_words((unsigned long long)s_piece[4 * lane] | ((unsigned long long)s_piece[4 * lane + 1] << 16) |
                       ((unsigned long long)s_piece[4 * lane + 2] << 32) | ((unsigned long long)s_piece[4 * lane + 3] << 48));
        __syncthreads();
        base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
    const int usable = P.cap - PAD;

    // One thread per KEPT pixel (not per mask bit): thread t of the segment finds the word holding its pixel by a
    // binary search over the 64 word offsets (LDS) and the bit by a popcount bisection, so every lane does useful
    // work and consecutive lanes gather consecutive (raster-adjacent) addresses.  blockIdx.z selects a group of
    // K2_KG key-points; two pixels per thread are in flight before the first use (the stage is latency-bound).
    const int k0 = blockIdx.z * K2_KG;
    const int T = s_total;
    auto locate = [&](int t, int& pos, int& p) {
        // branch-free: every decision is the sign bit of a difference turned into an all-ones / zero mask
        int lo = 0;
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) lo += st & ~((t - s_woff[lo + st]) >> 31);  // s_woff[lo + st] <= t; lo + st <= 63
        const unsigned long long wd = s_word[lo];
        int r = t - s_woff[lo], bitpos = 0;
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) {
            const int c = __popcll((wd >> bitpos) & ((1ull << st) - 1ull));
            const int take = ~((r - c) >> 31);  // r >= c
            bitpos += st & take;
            r -= c & take;
        }
        pos = base + t;
        p = (w0 + lo) * 64 + bitpos;
    };
    auto emit = [&](int pos, int x, int y, const float* ux, const float* uy) {
#pragma unroll
        for (int kk = 0; kk < K2_KG; ++kk) {
            if (k0 + kk >= P.vn) break;
            const size_t o = ((size_t)bi * P.vn + k0 + kk) * P.cap + pos;
            if (LITERAL) {
                P.rec[o] = make_float4((float)x, (float)y, ux[kk], uy[kk]);
            } else {
                const bool dead = norm1_literal(ux[kk], uy[kk]) <= kF1e6;  // never votes (:119-121): stored as a zero record
                P.rec[o] = make_float4((float)x, (float)y, dead ? 0.f : ux[kk], dead ? 0.f : uy[kk]);
            }
        }
    };
    for (int t0 = threadIdx.x; t0 < T; t0 += 512) {
        const int t1 = t0 + 256;
        const bool has1 = t1 < T;
        int pos0, p0, pos1 = 0, p1 = 0;
        locate(t0, pos0, p0);
        if (has1) locate(t1, pos1, p1);
        const int y0 = p0 / P.w, x0 = p0 - y0 * P.w;
        const int y1 = p1 / P.w, x1 = p1 - y1 * P.w;
        const int64_t v0 = (int64_t)bi * P.vs0 + (int64_t)y0 * P.vs1 + (int64_t)x0 * P.vs2;  // element offsets
        const int64_t v1 = (int64_t)bi * P.vs0 + (int64_t)y1 * P.vs1 + (int64_t)x1 * P.vs2;
        float ux0[K2_KG], uy0[K2_KG], ux1[K2_KG], uy1[K2_KG];
#pragma unroll
        for (int kk = 0; kk < K2_KG; ++kk) {
            const int k = (k0 + kk < P.vn) ? k0 + kk : P.vn - 1;  // clamp: loads stay in bounds, unconditional
            ux0[kk] = ld_elem<VT>(P.vertex, v0 + (int64_t)k * P.vs3);
            uy0[kk] = ld_elem<VT>(P.vertex, v0 + (int64_t)k * P.vs3 + P.vs4);
            ux1[kk] = ld_elem<VT>(P.vertex, v1 + (int64_t)k * P.vs3);  // (p1 = 0 when there is no second pixel: a valid address)
            uy1[kk] = ld_elem<VT>(P.vertex, v1 + (int64_t)k * P.vs3 + P.vs4);
        }
        if (pos0 < usable) {
            if (k0 == 0) P.pix[(size_t)bi * P.cap + pos0] = p0;
            emit(pos0, x0, y0, ux0, uy0);
        }
        if (has1 && pos1 < usable) {
            if (k0 == 0) P.pix[(size_t)bi * P.cap + pos1] = p1;
            emit(pos1, x1, y1, ux1, uy1);
        }
    }
    if (last) {  // the block that owns the last segment knows the total
        const int total = base + s_total;
        const int tn = total < usable ? total : usable;
        if (k0 == 0 && wave == 0) {  // nothing zero-fills ctrl: this block owns tn0 / tn / status of its image
            if (lane == 0) {
                P.ctrl[bi * CTRL_STRIDE + C_TN0] = tn0;
                P.ctrl[bi * CTRL_STRIDE + C_TN] = tn;
                P.ctrl[bi * CTRL_STRIDE + C_STATUS] = total > usable ? PVNET_S_OVERFLOW : 0;
                if (bi == 0) call_flags_ptr(P)[CF_ANY_CULLED] = 0;   // (K3 sets it)
            }
        }
        const int tpad = (tn + PAD - 1) / PAD * PAD;  // sentinel records: zero direction never votes
        const int kn = (k0 + K2_KG < P.vn ? k0 + K2_KG : P.vn) - k0;
        for (int i = threadIdx.x; i < (tpad - tn) * kn; i += 256) {
            const int kk = i / (tpad - tn), t = tn + i - kk * (tpad - tn);
            P.rec[((size_t)bi * P.vn + k0 + kk) * P.cap + t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// plan (one extra block per image in the hypothesis launch): the image's gates (ransac_voting_gpu.py:531-534),
// chunk count, local origin, its offset in the list of scoring work items (each block sums the item counts of the
// images before it -- b loads, no serial scan) and one 16-byte descriptor per work item.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int plan_item_count(const VoteParams& P, int j, int* nch_out) {
    const int tn0 = P.ctrl[j * CTRL_STRIDE + C_TN0], tn = P.ctrl[j * CTRL_STRIDE + C_TN];
    const bool skip = tn0 < P.min_num || tn <= 0;
    const int nch = skip ? 0 : (tn + P.chunk - 1) / P.chunk;
    if (nch_out) *nch_out = nch;
    return ((nch + P.wg_s - 1) / P.wg_s) * P.vn * (P.hgroups / P.wg_g);
}

// item descriptor (image, key-point | culled << 16, chunk group, hypothesis slice): the scoring kernels decode it with these
constexpr int ITEM_CULL_SHIFT = 16;
__device__ __forceinline__ int item_kp(int y) { return y & 0xFFFF; }
__device__ __forceinline__ bool item_culled(int y) { return (y >> ITEM_CULL_SHIFT) != 0; }

// culled: the disc-culling kernel scores this image's key-points (their items carry the mark)
__device__ __forceinline__ void plan_image(const VoteParams& P, int bi, bool culled) {
    constexpr int NT = 256;
    __shared__ int s_part[NT / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int part = 0;
    for (int j = threadIdx.x; j < bi; j += NT) part += plan_item_count(P, j, nullptr);
    part = wave_reduce_add(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) base += s_part[i];
    int nch;
    const int n = plan_item_count(P, bi, &nch);
    if (threadIdx.x == 0) {
        const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
        P.ctrl[bi * CTRL_STRIDE + C_NCHUNKS] = nch;
        P.ctrl[bi * CTRL_STRIDE + C_ITEM_BASE] = base;
        const int pm = nch ? P.pix[(size_t)bi * P.cap + tn / 2] : 0;  // local origin for the expanded form
        P.ctrl[bi * CTRL_STRIDE + C_OX] = pm % P.w;
        P.ctrl[bi * CTRL_STRIDE + C_OY] = pm / P.w;
        if (!nch) P.ctrl[bi * CTRL_STRIDE + C_STATUS] |= PVNET_S_SKIPPED;
        if (culled && n > 0) call_flags_ptr(P)[CF_ANY_CULLED] = 1;   // (zeroed by K2; every writer writes the same)
        if (bi == P.b - 1) {
            P.ctrl[P.b * CTRL_STRIDE] = base + n;  // total number of work items
            P.ctrl[P.b * CTRL_STRIDE + 6] = P.layout_fp;  // which layout the offsets of this workspace follow (epilogues check)
            P.ctrl[P.b * CTRL_STRIDE + 4] = 0;     // exact mode, PVNET_F_BAND_STATS: flagged cells / literal tests
            P.ctrl[P.b * CTRL_STRIDE + 5] = 0;
            P.ctrl[P.b * CTRL_STRIDE + 1] = 0;     // disc culling, PVNET_F_BAND_STATS: fine steps executed / steps of the full kernel
            P.ctrl[P.b * CTRL_STRIDE + 7] = 0;
        }
    }
    const int HQ = P.hgroups / P.wg_g, nchg = (nch + P.wg_s - 1) / P.wg_s;
    for (int local = threadIdx.x; local < n; local += NT) {
        const int hq = local % HQ, t = local / HQ;
        const int k = t / nchg;
        const int flag = culled ? (1 << ITEM_CULL_SHIFT) : 0;
        P.items[base + local] = make_int4(bi, k | flag, t % nchg, hq);  // (image, key-point | culled, chunk group, hyp slice)
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3: hypotheses                                               (ransac_voting_gpu.py:547,554; kernel.cu:11-49)
// One launch, three kinds of block per image (all of image bi on XCD bi % 8): ceil(hn vn / 256) blocks of one thread per (hypothesis,
// key-point); one block that plans the scoring work items; and, where the layout supports disc culling, one block per key-point
// that -- IF that key-point is culled -- sorts its hypotheses along a Hilbert curve and describes every tile of 32 by a disc.
// ------------------------------------------------------------------------------------------------------------
// position of (x, y) on the Hilbert curve of a 2^bits x 2^bits grid (consecutive positions are neighbouring cells)
__device__ __forceinline__ uint32_t hilbert_index(uint32_t x, uint32_t y, int bits) {
    uint32_t d = 0;
    for (uint32_t sft = 1u << (bits - 1); sft > 0; sft >>= 1) {
        const uint32_t rx = (x & sft) ? 1u : 0u, ry = (y & sft) ? 1u : 0u;
        d += sft * sft * ((3u * rx) ^ ry);
        if (ry == 0u) {
            if (rx == 1u) { x = ~x; y = ~y; }   // (only the bits below sft are looked at from here on)
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

constexpr int CULL_NPX = 256;              // pixels per work item of the culling kernel: 8 pixel tiles, list entries are 16-bit
constexpr int CULL_HN = 1024;              // hypotheses per key-point (hn_pad) of the layouts that can cull: one hypothesis slice, 32 tiles,
                                           // four sort keys per thread of a 256-thread block
constexpr int CULL_DEAD = 8 * TILE_U4_;    // uint4 index of the dead A row behind the item's 8 tiles (x = -4: no vote, no flag)
constexpr int NCAND = 8, KP_MAX = 32;      // candidate intersections per key-point of the origin estimate; key-points it handles

// What every K3 block works out for itself about its image before anything else (cheap -- 8 vn intersections -- and no block then
// waits for another): per key-point the ORIGIN of the exact mode's rounding band and whether the key-point is DISC-CULLED.
//   origin   eight candidate intersections from FIXED pixel pairs spread over the foreground list (records t and t + tn / 2), their
//            component-wise median, rounded to integers.  It only scales the band -- no result depends on it -- so a bad estimate
//            (fewer than three usable candidates: the image's median pixel instead) costs re-evaluations, never correctness.
//   culling  P.cull = 1: every key-point (PVNET_SCORE_CULL=1: tests and probes); P.cull = 2 (the default where the layout supports
//            it): a key-point VOTES for culling when its candidates lie close together -- spread S (median Chebyshev distance
//            from the median point) <= cull_q rho tan(theta0), i.e. the field's angular noise is small against the threshold
//            angle, so most pixels are certain for most hypothesis tiles and the gathered rest is cheaper than the dense sweep
//            (crossover measured: profiles/r06_cull_crossover.txt) -- and the IMAGE's key-points are culled together when the
//            majority votes so (image_culled()): the spread of eight candidates scatters, and a call whose key-points split between
//            the two scoring kernels pays the fixed cost of both (r06d: 220 us against 197 none / 174 all culled at sigma 0.01).
//            Any choice gives the same counts; a wrong one only costs time.
struct KpShared {
    float cand[KP_MAX * NCAND * 2];
    float med[KP_MAX * 2];
    int org[KP_MAX * 2];
    int vote[KP_MAX];      // 1: this key-point's candidates say "cull"
};
__device__ __forceinline__ bool image_culled(const KpShared& S, int vn) {   // (after kp_preamble's last barrier; every thread the same)
    int votes = 0;
    for (int kk = 0; kk < vn; ++kk) votes += S.vote[kk];
    return 2 * votes > vn;
}
__device__ __forceinline__ void kp_preamble(const VoteParams& P, int bi, int tn, bool live, KpShared& S) {
    int pm = 0;
    if (live) pm = P.pix[(size_t)bi * P.cap + tn / 2];
    const int kk = threadIdx.x / NCAND, j = threadIdx.x % NCAND;
    const bool mine = (int)threadIdx.x < P.vn * NCAND;
    if (mine) {
        float cx = __uint_as_float(0x7FC00000u), cy = cx;   // NaN = no candidate
        if (live) {
            const int ta = (int)(((long long)(2 * j + 1) * tn) >> 4);
            int tb = ta + tn / 2;
            tb = tb >= tn ? tb - tn : tb;
            const float4 q0 = P.rec[((size_t)bi * P.vn + kk) * P.cap + ta], q1 = P.rec[((size_t)bi * P.vn + kk) * P.cap + tb];
            float hx0, hy0;
            hyp_intersect(q0.z, q0.w, q0.x, q0.y, q1.z, q1.w, q1.x, q1.y, hx0, hy0);
            if ((hx0 != 0.f || hy0 != 0.f) && fabsf(hx0) < 1048576.f && fabsf(hy0) < 1048576.f) { cx = hx0; cy = hy0; }
        }
        S.cand[threadIdx.x * 2] = cx;
        S.cand[threadIdx.x * 2 + 1] = cy;
    }
    __syncthreads();
    // median by rank, one thread per candidate (no arrays in registers: this kernel's small VGPR allocation is what lets two of
    // its workgroups start beside a resident scoring kernel): candidate j is the median of a coordinate when n / 2 valid
    // ones sort before it (NaN compares false: never counted, never the median)
    const float* cand = S.cand + (mine ? kk : 0) * NCAND * 2;
    int n = 0;
    if (mine) {
        int rx = 0, ry = 0;
        const float vx = cand[2 * j], vy = cand[2 * j + 1];
        for (int m2 = 0; m2 < NCAND; ++m2) {
            const float ux = cand[2 * m2], uy = cand[2 * m2 + 1];
            n += ux == ux ? 1 : 0;
            rx += (ux < vx || (ux == vx && m2 < j)) ? 1 : 0;
            ry += (uy < vy || (uy == vy && m2 < j)) ? 1 : 0;
        }
        if (vx == vx && rx == n / 2) S.med[kk * 2] = vx;
        if (vy == vy && ry == n / 2) S.med[kk * 2 + 1] = vy;
    }
    __syncthreads();
    if (mine && n >= 3) {   // the candidates' spread: the median of their (Chebyshev) distances from the median point
        const float mx = S.med[kk * 2], my = S.med[kk * 2 + 1];
        const float dj = fmaxf(fabsf(cand[2 * j] - mx), fabsf(cand[2 * j + 1] - my));
        int rank = 0;
        for (int m2 = 0; m2 < NCAND; ++m2) {
            const float d2 = fmaxf(fabsf(cand[2 * m2] - mx), fabsf(cand[2 * m2 + 1] - my));
            rank += (d2 < dj || (d2 == dj && m2 < j)) ? 1 : 0;
        }
        if (dj == dj && rank == n / 2) {
            // Is the key-point a better origin than the median pixel?  With hypotheses spread S about it, at distance D from
            // the object (radius Ro), the band bound (R + rho)(1 + r / rho) is about (S + rho)(1 + (D + Ro) / rho) there and
            // (D + S + rho)(1 + Ro / rho) about the median pixel: take the smaller.  (Fields whose lines are nearly parallel
            // scatter their intersections over 1e5 px: S ~ D -- the median pixel; the benchmark field: S ~ 3 px -- the key-point.)
            const float rho = band_rho(tn), ro = rho * (1.f / 0.6f);
            const float dist = fmaxf(fabsf(mx - (float)(pm % P.w)), fabsf(my - (float)(pm / P.w)));
            const bool kp = (dj + rho) * (1.f + (dist + ro) / rho) < (dist + dj + rho) * (1.f + ro / rho);
            S.org[kk * 2] = kp ? (int)rintf(mx) : pm % P.w;
            S.org[kk * 2 + 1] = kp ? (int)rintf(my) : pm / P.w;
            S.vote[kk] = P.cull == 1 || (P.cull == 2 && kp && dj <= P.cull_q * rho * P.tau) ? 1 : 0;
        }
    } else if (mine && j == 0) {   // fewer than three usable candidates
        S.org[kk * 2] = pm % P.w;
        S.org[kk * 2 + 1] = pm / P.w;
        S.vote[kk] = (P.cull == 1 && live) ? 1 : 0;
    }
    __syncthreads();
}

// The block of a CULLED key-point: generates the key-point's hypotheses once more (the same draws, the same arithmetic as the
// hypothesis blocks, which write the caller-order `hyp` array and zero `counts`), SORTS them along a Hilbert curve about the band
// origin so that every 32 consecutive ones -- one MFMA hypothesis tile -- lie close together, and describes each tile by a disc:
// centre q (bounding-box centre), radius rho_T.  The scoring kernel tests every pixel ONCE against the centre of each tile (one
// MFMA pair per 32 pixels x 32 tiles) and only gathers the pixels whose vote is not the same for the whole disc.
//   sorted order : hypb (B columns), hyps (raw hypotheses, for the literal re-evaluation), perm (-> caller index: where the counts go)
//   per tile     : hypc = the B column of the centre at scale s' = 0.9 / (G + E), hypg = g = G / (G + E), with
//                  G = rho_T / thresh   (|m(h) - m(q)| <= |h - q| / cos theta0: the margin's Lipschitz constant) and
//                  E = kband (R_q + rho_T + rho)   (rounding band of every hypothesis of the disc + the matrix pipe's own error)
// A pixel i with row scale |M_i| <= mu_i is CERTAIN for the tile when |x'| >= 1 - g (1 - mu_i), x' = s' |M_i| m_i(q) as the two
// MFMAs return it: then |m_i(q)| > rho_T / thresh + band, so m_i has one sign on the whole disc and the reference's float32 test
// agrees with it for every hypothesis of the tile (derivation: DESIGN.md section 4, "disc culling").
// Padding hypotheses (>= hn) sort to the end; a tile without a real hypothesis is never scored.
// Round 6: 256 threads with four keys each (round 5: 1 024 threads, one key each, 40 us for 288 blocks -- two rounds on 256 CUs and
// sixteen waves per barrier); a block of four waves is resident wherever a hypothesis block is.
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

// (-DPVNET_K3_PROBE, tools/experiments/k3_probe.py: shader-clock stamps of the block's phases into the unused tail of the item list)
#ifdef PVNET_K3_PROBE
#define PV_K3_STAMP(i) do { if (threadIdx.x == 0) k3_stamp[i] = (int)(clock64() - k3_t0); } while (0)
#else
#define PV_K3_STAMP(i) do { } while (0)
#endif
// the two pixels of draw i = h vn + k of image bi (ransac_voting_gpu.py:547: one [hn, vn, 2] draw per image; or the caller's idxs)
__device__ __forceinline__ void draw_pair(const VoteParams& P, int bi, int i, int tn, int& t0, int& t1) {
    if (P.idxs) {
        t0 = P.idxs[((size_t)bi * P.hn * P.vn + i) * 2];
        t1 = P.idxs[((size_t)bi * P.hn * P.vn + i) * 2 + 1];
        t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);  // memory safety only; valid idxs are untouched
        t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
    } else {
        const uint32_t key = pvnet_rng_key(P.seed, PVNET_TAG_HYP, (uint32_t)(P.image_base + bi));
        t0 = (int)pvnet_rng_below(pvnet_rng_at(key, (uint32_t)i * 2u), (uint32_t)tn);
        t1 = (int)pvnet_rng_below(pvnet_rng_at(key, (uint32_t)i * 2u + 1u), (uint32_t)tn);
    }
}
// records of the cull block's hypothesis h = e NT + tid (zero records beyond hn)
__device__ __forceinline__ void cull_block_load(const VoteParams& P, int bi, int k, int tn, int e, float4& qa, float4& qb) {
    const int h = e * 256 + (int)threadIdx.x;
    qa = qb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < P.hn) {
        int t0, t1;
        draw_pair(P, bi, h * P.vn + k, tn, t0, t1);
        qa = P.rec[((size_t)bi * P.vn + k) * P.cap + t0];
        qb = P.rec[((size_t)bi * P.vn + k) * P.cap + t1];
    }
}
// (requesting the first records before the preamble -- the block is one chain of dependent waits -- gained nothing for a culled call
//  and cost a call without culled key-points 0.5 us: r06l)
__device__ __forceinline__ void cull_block(const VoteParams& P, int bi, int k, int tn, const KpShared& S, float2* s_h, uint32_t* s_key,
                                           long long k3_t0) {
    constexpr int NT = 256, E = CULL_HN / NT;
    const int tid = threadIdx.x;
    const size_t bk = (size_t)bi * P.vn + k;
#ifdef PVNET_K3_PROBE
    const long long max_items = (long long)P.b * P.vn * (P.hgroups / P.wg_g) * ((P.max_chunks + P.wg_s - 1) / P.wg_s);
    int* const k3_stamp = reinterpret_cast<int*>(P.items + (max_items - 2 - 2 * (long long)bk));
#endif
    PV_K3_STAMP(0);   // preamble done
    const float rho = band_rho(tn);
    const float ox = (float)S.org[k * 2], oy = (float)S.org[k * 2 + 1];
    // ---- this key-point's hypotheses (kernel.cu:11-49), caller order: thread t takes h = t, t + 256, ...; two pairs of records in
    //      flight at a time (three do not fit the kernel's 40 registers), the next requested as soon as a pair has been consumed
    static_assert(E == 4, "cull_block: four hypotheses per thread");
    {
        auto put = [&](int h, const float4& a, const float4& b) {
            float hx = 0.f, hy = 0.f;
            if (h < P.hn) hyp_intersect(a.z, a.w, a.x, a.y, b.z, b.w, b.x, b.y, hx, hy);
            s_h[h] = make_float2(hx, hy);
        };
        float4 qa[2], qb[2];
        cull_block_load(P, bi, k, tn, 0, qa[0], qb[0]);
        cull_block_load(P, bi, k, tn, 1, qa[1], qb[1]);
        put(tid, qa[0], qb[0]);
        cull_block_load(P, bi, k, tn, 2, qa[0], qb[0]);
        put(NT + tid, qa[1], qb[1]);
        cull_block_load(P, bi, k, tn, 3, qa[1], qb[1]);
        put(2 * NT + tid, qa[0], qb[0]);
        put(3 * NT + tid, qa[1], qb[1]);
    }
    PV_K3_STAMP(1);   // hypotheses
    // ---- sort keys: position on a Hilbert curve of 1/8-pixel cells about the origin (11 bits per coordinate: +-128 px); far and
    //      non-finite hypotheses clamp to the border.  (a thread reads back only what it wrote: no barrier yet; which slot of the
    //      sort a key starts in does not matter)
    constexpr int idxbits = 10, cbits = (32 - idxbits) >> 1;
    uint32_t key[E];
    {
        const float cells = (float)(1 << cbits);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int h = e * NT + tid;
            uint32_t kv = (0xFFFFFFFFu << idxbits) | (uint32_t)h;   // padding: behind every real hypothesis (ties broken by the index)
            if (h < P.hn) {
                const float2 hv = s_h[h];
                const float fx = fminf(fmaxf((hv.x - ox) * 8.f + 0.5f * cells, 0.f), cells - 1.f);   // (NaN -> 0)
                const float fy = fminf(fmaxf((hv.y - oy) * 8.f + 0.5f * cells, 0.f), cells - 1.f);
                kv = (hilbert_index((uint32_t)fx, (uint32_t)fy, cbits) << idxbits) | (uint32_t)h;
            }
            key[e] = kv;
        }
    }
    PV_K3_STAMP(2);   // keys
    // ---- bitonic sort, ascending, of the 1 024 keys at slot i = 4 tid + e (round 6b; the first form, slot = 256 e + tid with
    //      __shfl_xor, took 23 700 cycles: 45 of the 55 stages crossed lanes through the LDS crossbar).  Now 19 stages (partners 1 and 2
    //      slots away) stay in the thread's own registers, 33 cross LANES -- by DPP (lane ^ 1, ^ 2, ^ 8), v_permlane16_swap /
    //      v_permlane32_swap (^ 16, ^ 32) and one ds_swizzle (^ 4) -- and 3 cross WAVES through LDS.
    auto cas = [&](uint32_t& mine_, uint32_t other, int i, int kk, int jj) {
        const bool keep_min = ((i & jj) == 0) == ((i & kk) == 0);
        const uint32_t lo = mine_ < other ? mine_ : other, hi = mine_ < other ? other : mine_;
        mine_ = keep_min ? lo : hi;
    };
#pragma unroll
    for (int kk = 2; kk <= CULL_HN; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            if (jj == 1) {
                { const uint32_t a = key[0], c = key[1]; cas(key[0], c, 4 * tid, kk, 1); cas(key[1], a, 4 * tid + 1, kk, 1); }
                { const uint32_t a = key[2], c = key[3]; cas(key[2], c, 4 * tid + 2, kk, 1); cas(key[3], a, 4 * tid + 3, kk, 1); }
            } else if (jj == 2) {
                { const uint32_t a = key[0], c = key[2]; cas(key[0], c, 4 * tid, kk, 2); cas(key[2], a, 4 * tid + 2, kk, 2); }
                { const uint32_t a = key[1], c = key[3]; cas(key[1], c, 4 * tid + 1, kk, 2); cas(key[3], a, 4 * tid + 3, kk, 2); }
            } else if (jj >= 256) {   // the partner is one / two waves away
#pragma unroll
                for (int e = 0; e < E; ++e) s_key[4 * tid + e] = key[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) cas(key[e], s_key[(4 * tid + e) ^ jj], 4 * tid + e, kk, jj);
                __syncthreads();
            } else {                  // the partner is lane ^ (jj / 4)
#pragma unroll
                for (int e = 0; e < E; ++e) cas(key[e], lane_xor(key[e], jj >> 2), 4 * tid + e, kk, jj);
            }
        }
    }
    __syncthreads();   // (s_h of the other threads: every hypothesis has been written -- the sort's own barriers already saw to it)
    PV_K3_STAMP(3);   // sort
    // ---- sorted outputs + one disc per tile of 32 sorted hypotheses: a tile = the 4 slots of 8 consecutive threads (slot p = 4 tid + e)
    constexpr uint32_t imask = (uint32_t)CULL_HN - 1u;
    constexpr int ntl = CULL_HN >> 5;
    float hxo[E], hyo[E];
    bool real[E];
    int4 pj;
    float mnx = 3.0e38f, mxx = -3.0e38f, mny = 3.0e38f, mxy = -3.0e38f;
    int bad = 0, nreal = 0;
    {
        float2 hv[E];
        int jv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            jv[e] = (int)(key[e] & imask);
            real[e] = jv[e] < P.hn;
            hv[e] = real[e] ? s_h[jv[e]] : make_float2(0.f, 0.f);
            hxo[e] = hv[e].x - ox;
            hyo[e] = hv[e].y - oy;
            // the tile's bounding box, whether it holds a far / non-finite hypothesis (the tile is then scored in full), how many real ones
            mnx = real[e] ? fminf(mnx, hxo[e]) : mnx;
            mxx = real[e] ? fmaxf(mxx, hxo[e]) : mxx;
            mny = real[e] ? fminf(mny, hyo[e]) : mny;
            mxy = real[e] ? fmaxf(mxy, hyo[e]) : mxy;
            bad |= (real[e] && (!(fabsf(hxo[e]) < BAND_FAR) || !(fabsf(hyo[e]) < BAND_FAR))) ? 1 : 0;
            nreal += real[e] ? 1 : 0;
        }
        pj = make_int4(jv[0], jv[1], jv[2], jv[3]);
        *reinterpret_cast<int4*>(P.perm + bk * CULL_HN + 4 * tid) = pj;
        *reinterpret_cast<int4*>(P.cnts + bk * CULL_HN + 4 * tid) = make_int4(0, 0, 0, 0);   // K4 accumulates into them
        float4* const oh = reinterpret_cast<float4*>(P.hyps + bk * CULL_HN + 4 * tid);
        oh[0] = make_float4(hv[0].x, hv[0].y, hv[1].x, hv[1].y);
        oh[1] = make_float4(hv[2].x, hv[2].y, hv[3].x, hv[3].y);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = make_uint4(0u, 0u, 0u, pk(0u, 0x3F80u));
        if (real[e]) b_col_exact(hxo[e], hyo[e], rho, P.kband, lo, hi);
        uint4* o = P.hypb + (bk * CULL_HN + 4 * tid + e) * 2;
        o[0] = lo;
        o[1] = hi;
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {   // over the tile's 8 threads
        mnx = fminf(mnx, __uint_as_float(lane_xor(__float_as_uint(mnx), off)));
        mxx = fmaxf(mxx, __uint_as_float(lane_xor(__float_as_uint(mxx), off)));
        mny = fminf(mny, __uint_as_float(lane_xor(__float_as_uint(mny), off)));
        mxy = fmaxf(mxy, __uint_as_float(lane_xor(__float_as_uint(mxy), off)));
        bad |= (int)lane_xor((uint32_t)bad, off);
        nreal += (int)lane_xor((uint32_t)nreal, off);
    }
    const float qx = 0.5f * (mnx + mxx), qy = 0.5f * (mny + mxy);   // centre, relative to the origin
    float r2 = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float dx = hxo[e] - qx, dy = hyo[e] - qy;
        r2 = (real[e] && !bad) ? fmaxf(r2, fmaf(dx, dx, dy * dy)) : r2;
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) r2 = fmaxf(r2, __uint_as_float(lane_xor(__float_as_uint(r2), off)));
    if ((tid & 7) == 0) {
        uint4 clo = make_uint4(0u, 0u, 0u, 0u), chi = make_uint4(0u, 0u, 0u, pk(0u, 0x3F80u));
        float g = 0.f;
        if (nreal > 0 && !bad) {
            const float Rq = __builtin_sqrtf(fmaf(qx, qx, qy * qy)) * 1.000001f;
            // radius of the disc about the point the column REALLY encodes (fl(q s) / s): the roundings of h - o, q, h - q, the
            // square root and q s are relative 2^-24 each, of |h - o| <= Rq + rt at most
            const float rt = __builtin_sqrtf(r2) * 1.000001f;
            const float rtu = rt + 4.0e-7f * (Rq + rt);
            const float G = rtu / P.thresh * 1.000001f;
            const float Eb = P.kband * (Rq + rtu + rho);
            const float Sg = G + Eb;
            const float sc = bf16_floor(BAND_TARGET / Sg);
            b_col_scaled(qx, qy, Rq + rtu, sc, clo, chi);
            if (sc > 0.f && Rq + rtu < BAND_FAR) g = G / Sg * 0.99999f;   // (rounded down: the certainty threshold 1 - g (1 - mu) only grows)
        }
        const int T = tid >> 3;
        uint4* oc = P.hypc + (bk * ntl + T) * 2;
        oc[0] = clo;
        oc[1] = chi;
        P.hypg[bk * ntl + T] = g;
    }
    PV_K3_STAMP(4);   // outputs issued
}

// which key-points of which image are disc-culled: int32 [b][vn] behind the band origins (the scoring kernels' flag travels in the
// item descriptors; this copy is for the epilogues that read a finished workspace: band_margin_kernel, the debug views)
__device__ __forceinline__ int32_t* kp_cull_ptr(const VoteParams& P, size_t bk) {
    return P.ctrl + (size_t)(P.b + 1) * CTRL_STRIDE + 2 * (size_t)P.b * P.vn + bk;
}

template <bool LITERAL>   // (amdgpu_num_vgpr: 40 usable of the 48 allocated -- the backend doubles the literal on this target)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(20))) void hypothesis_kernel(VoteParams P) {
    PVNET_SPARE_VGPRS(47);
    small_stage_prio();
    // Workgroups go to the 8 XCDs round-robin by linear id; every block of image bi is placed on XCD bi % 8 so that
    // the two random 16-byte record reads per hypothesis (several per 128-byte line of the image's records) hit
    // that XCD's L2 after the first touch instead of crossing the fabric once per XCD.
    const int nbd = (P.hn * P.vn + 255) / 256;               // hypothesis blocks per image
    const int nb = nbd + 1 + (!LITERAL && P.cull ? P.vn : 0);   // + the plan block + one block per key-point that may be culled
    const int slot = blockIdx.x >> 3;
    const int bi = (slot / nb) * 8 + (blockIdx.x & 7);
    const int blk = slot % nb;
    if (bi >= P.b) return;
#ifdef PVNET_K3_PROBE
    const long long k3_t0 = clock64();
#else
    const long long k3_t0 = 0;
#endif
    const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
    const bool live = P.ctrl[bi * CTRL_STRIDE + C_TN0] >= P.min_num && tn > 0;  // gates of :531-534
    __shared__ KpShared S;
    const bool kp_origin = !LITERAL && P.mode && P.exact && P.vn <= KP_MAX;   // block-uniform
    // the thread's own hypothesis: its two records are requested NOW, so that they travel while the origin is worked out
    const int i = blk * 256 + threadIdx.x;
    const int hh = i / P.vn, hk = i - hh * P.vn;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (blk < nbd && live && i < P.hn * P.vn) {
        int t0, t1;
        draw_pair(P, bi, i, tn, t0, t1);
        q0 = P.rec[((size_t)bi * P.vn + hk) * P.cap + t0];  // (x, y, direction) of the two pixels
        q1 = P.rec[((size_t)bi * P.vn + hk) * P.cap + t1];
    }
    if (kp_origin) {
        kp_preamble(P, bi, tn, live, S);
        if (blk == 0 && (int)threadIdx.x < P.vn) {   // for the scoring kernel's staging (a_rows_exact) and the epilogues
            const size_t bk = (size_t)bi * P.vn + threadIdx.x;
            int32_t* o = band_origin_ptr(P, bk);
            o[0] = S.org[threadIdx.x * 2];
            o[1] = S.org[threadIdx.x * 2 + 1];
            *kp_cull_ptr(P, bk) = (P.cull && image_culled(S, P.vn)) ? 1 : 0;
        }
    } else if (!LITERAL && P.mode && P.exact && blk == 0) {   // more than KP_MAX key-points: the image's median pixel for all of them
        int pm = 0;
        if (live) pm = P.pix[(size_t)bi * P.cap + tn / 2];
        for (int kk = threadIdx.x; kk < P.vn; kk += 256) {
            int32_t* o = band_origin_ptr(P, (size_t)bi * P.vn + kk);
            o[0] = pm % P.w;
            o[1] = pm / P.w;
            *kp_cull_ptr(P, (size_t)bi * P.vn + kk) = 0;
        }
    }
    // (fill_params: P.cull implies the exact mode and vn <= KP_MAX) this image's key-points go to the disc-culling kernel
    const bool culling = !LITERAL && P.cull && kp_origin && live && image_culled(S, P.vn);
    if (blk == nbd) {      // one extra block per image plans its scoring work items (consumed by the next launches only)
        plan_image(P, bi, culling);
        return;
    }
    if (blk > nbd) {       // the block of key-point blk - nbd - 1: sorted operands and tile discs, if that key-point is culled
        __shared__ float2 s_h[CULL_HN];
        __shared__ uint32_t s_key[CULL_HN];
        const int k = blk - nbd - 1;
        if (culling) cull_block(P, bi, k, tn, S, s_h, s_key, k3_t0);   // (block-uniform)
        return;
    }
    if (i < P.hn * P.vn) {
    const int h = hh, k = hk;
    float hx = 0.f, hy = 0.f;
    if (live) {
        const float2 d0 = rec_dir(q0), d1 = rec_dir(q1);
        hyp_intersect(d0.x, d0.y, q0.x, q0.y, d1.x, d1.y, q1.x, q1.y, hx, hy);
    }
    P.hyp[((size_t)bi * P.vn + k) * P.hn_pad + h] = make_float2(hx, hy);
    if (P.atomic_counts) P.counts[((size_t)bi * P.vn + k) * P.hn_pad + h] = 0;  // K4 accumulates into it
    if (!LITERAL && P.mode && !culling) {  // the same hypothesis about the band origin, as a bf16x3 B operand column
        float ox = 0.f, oy = 0.f;                         // (a culled key-point's columns are written, in sorted order, by its own block)
        if (kp_origin) {
            ox = (float)S.org[k * 2];
            oy = (float)S.org[k * 2 + 1];
        } else if (live) {
            const int pm = P.pix[(size_t)bi * P.cap + tn / 2];  // the origin plan_image() records for this image
            ox = (float)(pm % P.w);
            oy = (float)(pm / P.w);
        }
        uint4 lo, hi;
        if (P.exact) b_col_exact(hx - ox, hy - oy, band_rho(tn), P.kband, lo, hi);
        else b_col(hx - ox, hy - oy, lo, hi);
        uint4* o = P.hypb + (((size_t)bi * P.vn + k) * P.hn_pad + h) * 2;
        o[0] = lo;
        o[1] = hi;
    }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K4, VALU form: inlier scoring                               (kernel.cu:88-126 + ransac_voting_gpu.py:557-561)
// Used by literal mode (the reference's float32 operation order); its fast-mode instantiation (6-op vote_expanded)
// is what PVNET_SCORE_MODE=0 selects and what the matrix-pipe kernel below replaced (153-166 us -> 107-118 us).
//
// "Lane owns hypotheses": each lane keeps HPL hypotheses and their vote counters in VGPRs and walks the pixels of
// a chunk; 6 VALU ops per (hypothesis, pixel) test in fast mode (vote_expanded), all on VGPR operands.
//
// Measured on gfx950 (profiles/r01_ubench_valu.txt, r01_tune*.txt): a VALU op that takes an SGPR operand issues at
// about half the rate of a VGPR-only one, so streaming the (wave-uniform) pixel records through the scalar cache
// made a 7-op loop no faster than a 9-op one.  The records therefore go through LDS: a workgroup = 4 waves works on
// ONE (image, key-point, chunk group) planned by K3; its 256 threads load the chunk's records (one coalesced
// 16-byte load per thread), turn them into the expanded-form constants about the image origin and park them in
// LDS; every wave then reads them back as broadcast ds_read_b128 + ds_read_b64 (all lanes the same address:
// conflict-free, LDS pipe, not VALU) -- the 4 waves cover G hypothesis groups x S chunks.  Work items are strided
// over a persistent grid; the counts leave as integer atomic adds (or, PVNET_SCORE_ATOMIC=0, as coalesced uint16 rows
// per chunk).  At batch 32 the kernel issues
// ~139 M VALU wave-instructions in ~150 us = ~91 % of the 2-cycles-per-instruction bound at the 1.97 GHz it
// sustains (profiles/r01_streamk_experiment.txt), so what is left is the op count, not the schedule.
// ------------------------------------------------------------------------------------------------------------
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

constexpr int NB = 4;  // pixels per inner-loop step (4 ds_read_b128 + 4 ds_read_b64 in flight)

template <int HPL, bool LITERAL>
__global__ __launch_bounds__(256) void score_kernel(VoteParams P) {
    if (HPL == 8) PVNET_SPARE_VGPRS(95); else PVNET_SPARE_VGPRS(71);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = P.wg_g, S = P.wg_s;  // G * S == 4 waves
    const int npx = S * P.chunk;
    float4* s_a = reinterpret_cast<float4*>(smem);        // fast: (My, -Mx, -Ec, Tx)   literal: (x, y, ux, uy)
    float2* s_b = reinterpret_cast<float2*>(s_a + npx);   // fast: (Ty, -Ed)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t* __restrict__ ctrl = P.ctrl;
    const int total = ctrl[P.b * CTRL_STRIDE];

    const ItemRange ir = my_items(P, total);
    for (int item = ir.first; item < ir.end; item += ir.step) {
        const int4 desc = P.items[item];  // (image, key-point, chunk group, hypothesis slice), planned by K3
        const int bi = desc.x, k = item_kp(desc.y), cg = desc.z, hq = desc.w;
        const int nch = ctrl[bi * CTRL_STRIDE + C_NCHUNKS];
        const int tn = ctrl[bi * CTRL_STRIDE + C_TN];
        const size_t bk = (size_t)bi * P.vn + k;
        const int tpad = (tn + PAD - 1) / PAD * PAD;  // records up to tpad exist (sentinels past tn)

        // ---- stage the chunk group's records in LDS (fast mode: expanded-form constants about the image origin)
        const float ox = (float)ctrl[bi * CTRL_STRIDE + C_OX], oy = (float)ctrl[bi * CTRL_STRIDE + C_OY];
        lds_barrier();  // the previous item's readers are done
        for (int i = threadIdx.x; i < npx; i += 256) {
            const int p = cg * npx + i;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < tpad) q = P.rec[bk * P.cap + p];
            if (LITERAL) {
                s_a[i] = q;
            } else {
                float4 a;
                float2 b;
                make_pixrec(q, P.tau, ox, oy, a, b);
                s_a[i] = a;
                s_b[i] = b;
            }
        }
        lds_barrier();

        const int g = wave % G, sc = wave / G;
        const int c = cg * S + sc;
        if (c >= nch) continue;  // wave-uniform; barriers above are reached by every wave of the next trip
        const int hg = hq * G + g;
        const float2* __restrict__ hb = P.hyp + bk * P.hn_pad + (size_t)hg * 64 * HPL;
        float hx[HPL], hy[HPL], cnt[HPL];
#pragma unroll
        for (int j = 0; j < HPL; ++j) {
            const float2 hv = hb[j * 64 + lane];
            hx[j] = LITERAL ? hv.x : hv.x - ox;
            hy[j] = LITERAL ? hv.y : hv.y - oy;
            cnt[j] = 0.f;
        }
        const int n = (c * P.chunk + P.chunk <= tpad) ? P.chunk : tpad - c * P.chunk;  // multiple of PAD
        const float4* sa = s_a + sc * P.chunk;
        const float2* sb = s_b + sc * P.chunk;
        for (int i = 0; i < n; i += NB) {
            float4 qa[NB];
            float2 qb[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                qa[u] = sa[i + u];
                if (!LITERAL) qb[u] = sb[i + u];
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
#pragma unroll
                for (int j = 0; j < HPL; ++j) {
                    if (LITERAL)
                        cnt[j] += inlier_literal(qa[u].x, qa[u].y, qa[u].z, qa[u].w, hx[j], hy[j], P.thresh) ? 1.f : 0.f;
                    else
                        cnt[j] += vote_expanded(qa[u], qb[u], hx[j], hy[j]);
                }
            }
        }
        if (P.atomic_counts) {  // integer atomics straight into the count of every hypothesis (order-independent)
            int32_t* pc = P.counts + bk * P.hn_pad + (size_t)hg * 64 * HPL;
#pragma unroll
            for (int j = 0; j < HPL; ++j)
                if ((int)cnt[j] > 0) atomicAdd(pc + j * 64 + lane, (int)cnt[j]);
        } else {
            uint16_t* __restrict__ po = P.partial + (bk * P.max_chunks + c) * P.hn_pad + (size_t)hg * 64 * HPL;
#pragma unroll
            for (int j = 0; j < HPL; ++j) po[j * 64 + lane] = (uint16_t)(int)cnt[j];  // exact: counts < 2^24
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K4 (fast mode): matrix-pipe scoring.  A work item = (image, key-point, pixel group of wg_s chunks, slice of
// 4 * MH * 32 hypotheses).  The workgroup turns the group's records into bf16x3 A rows in LDS (32-pixel tiles,
// A_cr | A_dt); each of its 4 waves keeps the B columns of MH * 32 hypotheses in registers (written by K3) and, per
// tile, issues 2 MFMAs per 32 hypotheses: cr and dt of 32 x 32 (pixel, hypothesis) pairs land in the lane that
// owns the hypothesis (column = lane & 31, 16 rows per lane), so the vote is cnt += clamp(dt - |cr|) (vote8):
// 2 VALU operations per test instead of 6, the other four run on the matrix pipe at bf16 rate.
// ------------------------------------------------------------------------------------------------------------
constexpr int TILE_U4 = 128;  // uint4 per 32-pixel tile: four blocks of 32 x 16 bytes -- first operand K slots 0..7 of rows 0..31, its K slots
                              // 8..15, then the second operand's two halves.  (Round 6: until then a row's two 16-byte halves lay side by
                              // side, so the 16 lanes that one ds_read_b128 cycle serves -- sixteen rows, ONE half -- used only the even or only
                              // the odd 16-byte slots of the 256-byte bank row: two-way conflicts on every dense read, four- to five-way on the
                              // culling kernel's gathered ones.  Now slot = row mod 16: dense reads are conflict-free, gathered ones meet
                              // sixteen slots instead of eight.)
static_assert(TILE_U4_ == TILE_U4, "the disc-culling kernel's list addresses (CULL_DEAD, tile = a >> 7, row = a & 31) follow TILE_U4");

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
constexpr int VOTE_WRAP = 512;  // vote8 accumulators hold their count mod 512
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
constexpr float BAND_CLEAN = 1.0f;     // a cell whose minimum |a'|, |b'| reaches this holds no test inside the band

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
// The register allocator fills whatever budget the occupancy target leaves (3 waves per SIMD: up to 168 VGPRs), the library needs
// the top granule of every allocation unused (PVNET_SPARE_VGPRS): amdgpu_num_vgpr -- a literal, hence one definition per
// instantiation -- caps what the code may use one granule below what PVNET_SPARE_VGPRS makes the kernel allocate (the
// backend doubles the attribute's value on targets with a unified VGPR / AGPR file, hence the / 2).
template <int MH, int FOLD, bool TIMED, int NACC, bool RUNS> struct ScoreExact;
#define PV_DEF_SCORE_EXACT(MH_, FOLD_, TIMED_, NACC_, RUNS_, NVGPR_)                                                     \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8), amdgpu_num_vgpr(NVGPR_ / 2))) void       \
        score_exact_kernel_##MH_##_##FOLD_##_##TIMED_##_##NACC_##_##RUNS_(VoteParams P) {                                \
        score_exact_body<MH_, FOLD_, TIMED_ != 0, NACC_, RUNS_ != 0>(P);                                                 \
    }                                                                                                                    \
    template <> struct ScoreExact<MH_, FOLD_, TIMED_ != 0, NACC_, RUNS_ != 0> {                                          \
        static constexpr void (*kernel)(VoteParams) = score_exact_kernel_##MH_##_##FOLD_##_##TIMED_##_##NACC_##_##RUNS_; \
    };
#define PV_DEF_SCORE_EXACT4(MH_, NACC_, RUNS_, NVGPR_)                                                                   \
    PV_DEF_SCORE_EXACT(MH_, 0, 0, NACC_, RUNS_, NVGPR_) PV_DEF_SCORE_EXACT(MH_, 0, 1, NACC_, RUNS_, NVGPR_)              \
    PV_DEF_SCORE_EXACT(MH_, 1, 0, NACC_, RUNS_, NVGPR_) PV_DEF_SCORE_EXACT(MH_, 1, 1, NACC_, RUNS_, NVGPR_)
PV_DEF_SCORE_EXACT4(1, 2, 0, 104) PV_DEF_SCORE_EXACT4(2, 2, 0, 104) PV_DEF_SCORE_EXACT4(4, 2, 0, 136)
PV_DEF_SCORE_EXACT4(8, 1, 0, 120) PV_DEF_SCORE_EXACT4(8, 2, 0, 160)
PV_DEF_SCORE_EXACT4(8, 1, 1, 128) PV_DEF_SCORE_EXACT4(8, 2, 1, 160)
#undef PV_DEF_SCORE_EXACT4
#undef PV_DEF_SCORE_EXACT

// ------------------------------------------------------------------------------------------------------------
// K4 -- disc culling (round 5; exact mode, 8 hypothesis tiles per wave, 256-pixel work items): the exact kernel's two MFMAs and
// 40 vector operations are only spent on the (pixel, hypothesis tile) pairs whose outcome geometry does not already fix.
// Hypotheses arrive sorted along a Hilbert curve (hypothesis_cull_kernel): a tile of 32 is a disc (centre q, radius rho_T).
//   coarse pass  per item, ONE MFMA pair per pixel tile against the 32 tile CENTRES of the item's hypothesis slice (the eight
//                pixel tiles are shared out over the four waves): x' = s' |M_i| m_i(q).  |x'| >= 1 - g (1 - mu_i) means the
//                pixel's margin has one sign on the whole disc, outside the rounding band for every hypothesis in it: the pixel
//                votes for all 32 hypotheses (x' > 0: one count per tile, s_cv) or for none.  Every other pixel is UNCERTAIN
//                for that tile and its A-row address joins the tile's list (s_list, 16-bit LDS row addresses).
//   fine pass    a wave walks its eight hypothesis tiles; for each it scores ceil(uncertain / 32) GATHERED pixel groups -- lane
//                `col` of the MFMA's A operand reads the row the list names, so any 32 pixels of the item form a tile -- with
//                the exact kernel's epilogue (vote_subs / vote_slow_open / vote_slow_close: x = dt' - |cr'|, cells of 16 tests,
//                flagged cells re-evaluated literally).  Lists are padded to whole groups with a dead row (x = -4).
// Every count is the same integer as before: certain pixels add what the full test would have added (proof: DESIGN.md section 4),
// uncertain ones run the very same arithmetic.  What changes is the work: on the noisy benchmark field 59 % of the steps remain
// at thresh 0.99 (simulation: tools/cull_study.py, profiles/r05_cull_study.txt), none on a clean field.
// ------------------------------------------------------------------------------------------------------------
// TAIL (round 6): the second body of a merged launch (the dense items were scored by score_exact_body<..., HEAD> before it, which also
//           took the opening clock stamp)
// Items are always STRIDED over the workgroups here, whatever the dense body's mapping: contiguous runs (B columns and counters kept
// while the key-point stays) cost this body 27 % on the clean field (55 -> 70 us, r06k) -- its items are chains of waits, and a run
// puts the long ones of one key-point into one workgroup.  WIDE: the kernel allocates 136 VGPRs (the dense body runs contiguous
// runs: batches in flight), else 128.
template <bool TIMED, bool TAIL, bool WIDE>
__device__ __forceinline__ void score_cull_body(VoteParams P) {
    constexpr int MH = 8;
    // the merged kernel's allocation is the dense kernel's of the same item mapping: 136 VGPRs for contiguous runs (batches in flight: what
    // is left of the SIMD's 512 holds other streams' small stages -- at 144 the six-stream rate fell 2.8 %, r06g), 128 for a batch
    // alone (four waves per SIMD, four workgroups of 40 KB per CU).  This body spills a few per-item constants to fit.
    if (WIDE) PVNET_SPARE_VGPRS(135); else PVNET_SPARE_VGPRS(127);
    unsigned long long* __restrict__ stamps = reinterpret_cast<unsigned long long*>(P.pix);
    if (TIMED && !TAIL && threadIdx.x == 0) stamps[2 * blockIdx.x] = (unsigned long long)wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4* s_t = reinterpret_cast<uint4*>(smem);                                  // 8 A tiles + the dead row's tile: 9 x 2 KB
    unsigned* s_cells = reinterpret_cast<unsigned*>(s_t + 9 * TILE_U4);           // flagged cells of this item (4 * MH * 64 slots)
    // (no copy of the raw records here: the few flagged cells read theirs back from HBM / L2 -- with it the workgroup was 4 KB above the
    //  dense kernel's LDS, and the merged launch must not hold fewer workgroups per CU than the dense one)
    uint8_t* s_list = reinterpret_cast<uint8_t*>(s_cells + 4 * MH * 64);          // [32 tiles][256] the uncertain pixels (index in the item: one byte
                                                                                  // -- with 16-bit row addresses the workgroup took 48 KB, three per CU)
    float* s_sig = reinterpret_cast<float*>(s_list + 32 * CULL_NPX);              // [256] 1 - mu_i
    int* s_nu = reinterpret_cast<int*>(s_sig + CULL_NPX);                         // [32] uncertain pixels per hypothesis tile
    int* s_cv = s_nu + 32;                                                        // [32] certain votes per hypothesis tile
    __shared__ int s_ncell;
    const int32_t* __restrict__ ctrl = P.ctrl;
    const int total = ctrl[P.b * CTRL_STRIDE];
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* lbase = s_t + half * 32 + col;
    const int ntl = P.hn_pad >> 5;

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
    bf16x8 B[MH];
    bf16x8 Bc = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));   // centre column of hypothesis tile `col` of the slice
    float gcol = 0.f;                                                     // its g (0: every live pixel is uncertain)
    bool tile_live = false;                                               // tile `col` holds a real hypothesis
    unsigned cnt[MH];   // packed-norm vote counters (votes_of_norm): fine votes of the clean cells + the certain votes
#pragma unroll
    for (int t = 0; t < MH; ++t) cnt[t] = 0u;
    unsigned st_steps = 0u, st_full = 0u;   // PVNET_F_BAND_STATS: fine steps executed / steps the exact kernel would execute (this wave)
    auto flush_counts = [&](size_t fbk, int fh0) {
        int32_t* const pc = P.cnts + fbk * P.hn_pad + fh0;
        int lanex = threadIdx.x;
        asm volatile("" : "+v"(lanex));
        lanex &= 63;
#pragma unroll
        for (int t = 0; t + 1 < MH; t += 2) {  // lanes 0..31 finish tile t, lanes 32..63 tile t + 1
            const int c = votes_of_norm(half_wave_sum2(cnt[t], cnt[t + 1]));
            if (c > 0) atomicAdd(pc + t * 32 + lanex, c);
        }
#pragma unroll
        for (int t = 0; t < MH; ++t) cnt[t] = 0u;
    };
    // Round 6: the record of this thread's pixel of the NEXT item is requested while the current item is scored.  On the fields where
    // culling pays, an item is a chain of waits (descriptor -> pixel count -> record -> barrier -> centres -> lists -> ...), three
    // workgroups per CU deep, not a stream of instructions (profiles/r06d_phase_probe_cull.txt: 13 300 cycles per item on the clean
    // field, 3 000 of them the staging); the full kernel, which is bound by instructions issued, gained nothing from the same
    // prefetch in round 4.
    float4 q_next = make_float4(0.f, 0.f, 0.f, 0.f);
    int next_item = -1;   // the item q_next belongs to
    const ItemRange ir = my_items<false>(P, total);
    for (int item = ir.first; item < ir.end; item += ir.step) {
        const int4 desc = P.items[item];
        if (!item_culled(desc.y)) continue;   // (workgroup-uniform) a key-point the full exact kernel scores
        const int bi = desc.x, k = item_kp(desc.y), cg = desc.z, hq = desc.w;
        const int tn = ctrl[bi * CTRL_STRIDE + C_TN];
        const float rho = band_rho(tn);
        const size_t bk = (size_t)bi * P.vn + k;
        const int32_t* const org = band_origin_ptr(P, bk);
        const float ox = (float)org[0], oy = (float)org[1];
        const int tpad = (tn + PAD - 1) / PAD * PAD;
        const int hslice = hq * 4 * MH * 32;
        const int h0 = hslice + wave * MH * 32;
        // (the counters hold < 65536 votes per half: per item and lane pair at most 16 fine votes per group (8 groups) and the certain
        //  votes of the item's 256 pixels -- 384)

        lds_barrier();  // the previous item's tiles, lists and cells have been consumed
        PV_PHASE(3);
        if (threadIdx.x == 0) s_ncell = 0;
        if (threadIdx.x < 64) s_nu[threadIdx.x] = 0;   // s_nu and s_cv
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        {
            const int c2 = tid & 31, h2 = (tid >> 5) & 1;
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                const uint4 raw = P.hypb[(bk * P.hn_pad + h0 + t * 32 + c2) * 2 + h2];
                B[t] = __builtin_bit_cast(bf16x8, raw);
            }
            Bc = __builtin_bit_cast(bf16x8, P.hypc[(bk * ntl + hq * 32 + c2) * 2 + h2]);
            gcol = P.hypg[bk * ntl + hq * 32 + c2];
            tile_live = hslice + c2 * 32 < P.hn;
        }
        {   // thread = pixel: its A rows, its raw record, its 1 - mu
            const int i = tid;
            const int p = cg * CULL_NPX + i;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (next_item == item) q = q_next;   // (workgroup-uniform) requested during the previous item
            else if (p < tpad) q = P.rec[bk * P.cap + p];
            uint4* t = s_t + (i >> 5) * TILE_U4 + (i & 31);
            uint4 r0, r1, r2, r3;
            float mu;
            a_rows_exact(q, P.tau, ox, oy, rho, r0, r1, r2, r3, mu);
            t[0] = r0;
            t[32] = r1;
            t[64] = r2;
            t[96] = r3;
            s_sig[i] = 1.f - mu;
            if (i == 0) {   // the dead row the lists are padded with: dt' = -4, cr' = 0 -- no vote, no flag
                s_t[CULL_DEAD] = make_uint4(0u, 0u, 0u, 0u);
                s_t[CULL_DEAD + 32] = make_uint4(0u, 0u, 0u, pk(0u, 0xC080u));
                s_t[CULL_DEAD + 64] = make_uint4(0u, 0u, 0u, 0u);
                s_t[CULL_DEAD + 96] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        lds_barrier();
        PV_PHASE(0);
        if (item + ir.step < ir.end) {   // the next item's record for this thread, if the next item is one of this kernel's
            const int4 nd = P.items[item + ir.step];
            if (item_culled(nd.y)) {     // (workgroup-uniform)
                const int ntn = ctrl[nd.x * CTRL_STRIDE + C_TN];
                const int np = nd.z * CULL_NPX + tid;
                q_next = make_float4(0.f, 0.f, 0.f, 0.f);
                if (np < (ntn + PAD - 1) / PAD * PAD) q_next = P.rec[((size_t)nd.x * P.vn + item_kp(nd.y)) * P.cap + np];
                next_item = item + ir.step;
            }
        }

        const int left = (tpad - cg * CULL_NPX + 31) >> 5;
        const int nti = left < 8 ? left : 8;
        // ---- coarse pass: this wave's two pixel tiles against the 32 tile centres (both MFMA pairs issued before either is consumed)
        f32x16 cvd[2], cvc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pt = wave * 2 + u;
            cvd[u] = zero;
            cvc[u] = zero;
            if (pt < nti) {   // wave-uniform
                const bf16x8 Ad = __builtin_bit_cast(bf16x8, lbase[pt * TILE_U4]);
                const bf16x8 Ac = __builtin_bit_cast(bf16x8, lbase[pt * TILE_U4 + 64]);
                cvd[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, Bc, zero, 0, 0, 0);
                cvc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac, Bc, zero, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pt = wave * 2 + u;
            if (pt >= nti) break;   // wave-uniform
            const f32x16 vd = cvd[u], vc = cvc[u];
            unsigned um = 0u;
            int nv = 0;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 sg = *reinterpret_cast<const float4*>(s_sig + pt * 32 + r4 * 8 + half * 4);   // rows r4 * 8 + half * 4 + 0..3
                const float se[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = r4 * 4 + e;
                    const float x = vd[r] - fabsf(vc[r]);
                    const float thr = fmaf(-gcol, se[e], 1.f);   // 1 - g (1 - mu)
                    const bool vote = x >= thr, none = x <= -thr;   // (NaN: neither -- uncertain)
                    nv += vote ? 1 : 0;
                    um |= (vote || none) ? 0u : (1u << r);
                }
            }
            if (!tile_live) {   // a tile of padding hypotheses only: never scored, nobody reads its counts
                um = 0u;
                nv = 0;
            }
            if (nv) atomicAdd(&s_cv[col], nv);
            if (um) {
                int at = atomicAdd(&s_nu[col], __popc(um));
                uint8_t* const dst = s_list + col * CULL_NPX;
                while (um) {
                    const int r = __ffs((int)um) - 1;
                    um &= um - 1u;
                    dst[at++] = (uint8_t)(pt * 32 + (r >> 2) * 8 + half * 4 + (r & 3));   // the pixel (accumulator row r of this half-wave)
                }
            }
        }
        lds_barrier();
        PV_PHASE(1);   // (TIMED, this kernel: 0 staging, 1 coarse pass + barrier, 2 fine pass, 3 flush + cell list + re-evaluation + barriers)
        // ---- fine pass: the uncertain pixels of each of this wave's eight hypothesis tiles, gathered into groups of 32
        // the wave's eight list lengths and certain-vote counts in four 16-byte reads (one wait) -- read one by one, each behind the
        // store before it, they were a chain of sixteen LDS round trips per item: 3 600 cycles with nothing to score (r06d)
        // (the lengths are wave-uniform: into SGPRs at once; the tile's certain votes -- the same for its 32 hypotheses -- join the
        //  counters here, in ONE of the two half-waves that half_wave_sum2 joins)
        int nu8[MH];
        {
            const int4 a0 = *reinterpret_cast<const int4*>(s_nu + wave * MH), a1 = *reinterpret_cast<const int4*>(s_nu + wave * MH + 4);
            const int4 c0 = *reinterpret_cast<const int4*>(s_cv + wave * MH), c1 = *reinterpret_cast<const int4*>(s_cv + wave * MH + 4);
            const int av[MH] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, cv[MH] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int t = 0; t < MH; ++t) {
                nu8[t] = __builtin_amdgcn_readfirstlane(av[t]);
                cnt[t] += half == 0 ? (unsigned)cv[t] * 0xFFFFu : 0u;
            }
        }
        unsigned flg[MH];   // bit (groups - 1 - g) set = gathered group g of tile t holds a test inside the band
        float x0, x1, x2, x3, x4, x5, x6, x7, dmo;
        unsigned acc;
#define PV_XS x0, x1, x2, x3, x4, x5, x6, x7
#define PV_LO(v, w) v[0], w[0], v[1], w[1], v[2], w[2], v[3], w[3], v[4], w[4], v[5], w[5], v[6], w[6], v[7], w[7]
#define PV_HI(v, w) v[8], w[8], v[9], w[9], v[10], w[10], v[11], w[11], v[12], w[12], v[13], w[13], v[14], w[14], v[15], w[15]
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            flg[t] = 0u;
            const int j = wave * MH + t;
            const int nu = nu8[t];
            const int ng = (nu + 31) >> 5;
            if (TIMED || (P.flags & PVNET_F_BAND_STATS)) {
                st_steps += (unsigned)ng;
                st_full += (h0 + t * 32 < P.hn) ? (unsigned)nti : 0u;
            }
            if (ng > 0) {   // wave-uniform
                // Software pipeline over the tile's gathered groups: a group's A rows are requested one trip ahead (list entry -> row
                // address -> two 16-byte reads: a dependent LDS chain of ~200 cycles that would otherwise open every step) and the
                // previous group's last 15 vote operations fill the wait for this group's MFMAs, as in the exact kernel.
                // lane `col` of group g takes list entry 32 g + col -- a pixel index; its A rows start at uint4 (pixel tile) * TILE_U4 +
                // (row); beyond the list's end: the dead row (x = -4: no vote, no flag)
                const uint8_t* const lst = s_list + j * CULL_NPX + col;
                auto row_of = [&](int g) -> unsigned {
                    const unsigned e = lst[g * 32];
                    const unsigned a = ((e & 0xE0u) << 2) | (e & 31u);
                    return g * 32 + col < nu ? a : (unsigned)CULL_DEAD;
                };
                const unsigned a0 = row_of(0);
                uint4 Ra = s_t[a0 + half * 32], Rb = s_t[a0 + half * 32 + 64];
                unsigned an = row_of(ng > 1 ? 1 : 0);
                {
                    const f32x16 va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra), B[t], zero, 0, 0, 0);
                    const f32x16 vb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Rb), B[t], zero, 0, 0, 0);
                    Ra = s_t[an + half * 32];
                    Rb = s_t[an + half * 32 + 64];
                    an = row_of(ng > 2 ? 2 : ng - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 11");   // (the votes are inline asm: the wait states are ours, tools/check_mfma_hazard.py)
                    vote_subs(PV_XS, PV_LO(va, vb));
                    vote_slow_open(acc, dmo, PV_XS);
                    vote_subs(PV_XS, PV_HI(va, vb));
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (int g = 1; g < ng; ++g) {
                    const f32x16 va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra), B[t], zero, 0, 0, 0);
                    const f32x16 vb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Rb), B[t], zero, 0, 0, 0);
                    Ra = s_t[an + half * 32];        // (the last trip re-reads the last group: harmless)
                    Rb = s_t[an + half * 32 + 64];
                    an = row_of(g + 2 < ng ? g + 2 : ng - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    vote_slow_close(cnt[t], flg[t], acc, dmo, PV_XS);   // the previous group's last 15 operations fill the wait
                    asm volatile("s_nop 3");
                    vote_subs(PV_XS, PV_LO(va, vb));
                    vote_slow_open(acc, dmo, PV_XS);
                    vote_subs(PV_XS, PV_HI(va, vb));
                    __builtin_amdgcn_sched_barrier(0);
                }
                vote_slow_close(cnt[t], flg[t], acc, dmo, PV_XS);
            }
        }
#undef PV_XS
#undef PV_LO
#undef PV_HI
        PV_PHASE(2);
        int colx = col;
        asm volatile("" : "+v"(colx));
        const bool padded = h0 + MH * 32 > P.hn;
        flush_counts(bk, h0);
        unsigned long long bal[MH];   // (one slot reservation per wave and item, as in the dense body)
        int ncw = 0;
#pragma unroll
        for (int t = 0; t < MH; ++t) {
            if (padded && h0 + t * 32 + colx >= P.hn) flg[t] = 0u;   // padding columns: nobody reads their counts
            bal[t] = __ballot(flg[t] != 0u);
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
        // ---- flagged cells, decided by the reference's arithmetic: 16 lanes per cell, one gathered pixel each
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
                const float2 hv = P.hyps[bk * P.hn_pad + hslice + hl];
                const int j = hl >> 5;
                const int nu = s_nu[j], ng = (nu + 31) >> 5;
                const int row = (q >> 2) * 8 + hf * 4 + (q & 3);  // the 16 rows a lane of that half-wave holds
                int votes = 0;
                while (m) {
                    const int gb = __ffs((int)m) - 1;   // bit gb = gathered group ng - 1 - gb (vote_slow_close shifts them in)
                    m &= m - 1u;
                    const int slot = (ng - 1 - gb) * 32 + row;
                    if (slot < nu) {   // (beyond: the dead row)
                        const int px = cg * CULL_NPX + (int)s_list[j * CULL_NPX + slot];   // < tpad: rows beyond it are zero rows, never uncertain
                        const float4 r = P.rec[bk * P.cap + px];
                        votes += inlier_literal(r.x, r.y, r.z, r.w, hv.x, hv.y, P.thresh) ? 1 : 0;
                        ++ntests;
                    }
                }
                votes += __shfl_xor(votes, 8, 64);
                votes += __shfl_xor(votes, 4, 64);
                votes += __shfl_xor(votes, 2, 64);
                votes += __shfl_xor(votes, 1, 64);
                if (q == 0 && votes > 0) atomicAdd(P.cnts + bk * P.hn_pad + hslice + hl, votes);
            }
            if (P.flags & PVNET_F_BAND_STATS) {
                if (tid2 == 0) atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 4, ncell);
                if (q == 0 && ntests > 0) atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 5, ntests);
            }
        }
    }
    if ((P.flags & PVNET_F_BAND_STATS) && lane == 0) {   // development aid: how much of the exact kernel's work was left
        atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 1, (int)st_steps);
        atomicAdd(P.ctrl + P.b * CTRL_STRIDE + 7, (int)st_full);
    }
    if (TIMED) {
        lds_barrier();
        PV_PHASE(3);
        if (threadIdx.x == 0) {
            stamps[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
#pragma unroll
            for (int i = 0; i < 4; ++i) stamps[2 * gridDim.x + 4 * blockIdx.x + i] = ph[i];
        }
    }
#undef PV_PHASE
}
// ONE launch for a call whose key-points K3 may split between the two scoring forms (P.cull = 2, the default where the layout supports
// culling): every workgroup walks its work items twice -- the dense ones with the exact kernel's body, then the disc-culled ones.  A
// second launch for the culled items cost 7.5 us on the stream and 10 % of the six-stream rate when NOTHING was culled (2 304
// workgroups of 48 KB that read one descriptor each, profiles/r06f_stage_ab.txt), and a call split between two launches ran each
// at part of the machine (256 us against 195 / 172 for either form alone, r06e); here an item costs what its form costs, wherever it is.
#define PV_DEF_SCORE_BOTH(TIMED_, RUNS_, NVGPR_)                                                                          \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8), amdgpu_num_vgpr(NVGPR_ / 2))) void        \
        score_exact_kernel_both_##TIMED_##_##RUNS_(VoteParams P) {                                                        \
        const int any_culled = call_flags_ptr(P)[CF_ANY_CULLED];   /* (scalar load, long back when it is needed) */       \
        score_exact_body<8, 1, TIMED_ != 0, 1, RUNS_ != 0, true>(P);                                                      \
        if (any_culled || TIMED_) score_cull_body<TIMED_ != 0, true, RUNS_ != 0>(P);   /* (TIMED: the closing stamps) */  \
    }
PV_DEF_SCORE_BOTH(0, 0, 120) PV_DEF_SCORE_BOTH(1, 0, 120) PV_DEF_SCORE_BOTH(0, 1, 128) PV_DEF_SCORE_BOTH(1, 1, 128)
#undef PV_DEF_SCORE_BOTH
constexpr size_t CULL_LDS_BYTES = 9 * TILE_U4 * sizeof(uint4) + 4 * 8 * 64 * sizeof(unsigned) +
                                  32 * CULL_NPX * sizeof(uint8_t) + CULL_NPX * sizeof(float) + 64 * sizeof(int);
static_assert(4 * (CULL_LDS_BYTES + 64) <= 160 * 1024, "four workgroups of the merged scoring kernel per CU");

// ------------------------------------------------------------------------------------------------------------
// Development aid (pvnet_vote_band_margin, tools/band_margin.py): the exactness argument of the exact mode, MEASURED.
// On the workspace a complete exact-mode call left behind, every (pixel, hypothesis) test is evaluated twice: x = dt' - |cr'|
// from the very MFMAs, operands and subtraction the scoring kernel uses (same instructions on the same bits: the same x), and
// inlier_literal() on the raw record.  The scoring kernel trusts x wherever |x| >= 1; so the largest |x| among the tests whose
// matrix-pipe vote (x > 0) DIFFERS from the literal vote says how close an unflagged disagreement ever comes to the flag
// threshold: it must stay below 1, and the distance to 1 is the safety margin of the band (band_constant()).
// out[(image, key-point)][4] (uint32): max |x| over the disagreeing tests (float bits), their number, the tests with |x| < 1
// (the band as the kernel sees it), all tests (the last two mod 2^32).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_margin_kernel(VoteParams P, unsigned* __restrict__ out) {
    PVNET_SPARE_VGPRS(119);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* s_t = reinterpret_cast<uint4*>(smem);                 // 8 pixel tiles x 2 KB
    float4* s_raw = reinterpret_cast<float4*>(s_t + 8 * TILE_U4);
    const int k = blockIdx.x % P.vn, bi = blockIdx.x / P.vn, grp = blockIdx.y;
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
    if (P.ctrl[bi * CTRL_STRIDE + C_NCHUNKS] <= 0 || grp * 256 >= tn) return;   // block-uniform
    const size_t bk = (size_t)bi * P.vn + k;
    const int tpad = (tn + PAD - 1) / PAD * PAD;
    const int32_t* const org = band_origin_ptr(P, bk);
    const float ox = (float)org[0], oy = (float)org[1];
    const float rho = band_rho(tn);
    {
        const int i = threadIdx.x, p = grp * 256 + i;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < tpad) q = P.rec[bk * P.cap + p];
        s_raw[i] = q;
        uint4 r0, r1, r2, r3;
        float mu_unused;
        a_rows_exact(q, P.tau, ox, oy, rho, r0, r1, r2, r3, mu_unused);
        uint4* t = s_t + (i >> 5) * TILE_U4 + (i & 31);
        t[0] = r0;
        t[32] = r1;
        t[64] = r2;
        t[96] = r3;
    }
    __syncthreads();
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4* lbase = s_t + half * 32 + col;
    const int left = (tpad - grp * 256 + 31) >> 5, nti = left < 8 ? left : 8;
    float worst = 0.f;
    unsigned ndis = 0u, nband = 0u, ntest = 0u;
    const bool culled = P.cull && *kp_cull_ptr(P, bk) != 0;   // this key-point's operands are in Hilbert order
    for (int ht = wave; ht * 32 < (culled ? P.hn_pad : P.hn); ht += 4) {     // hypothesis tiles of this wave
        const int h = ht * 32 + col;
        const bf16x8 Bc = __builtin_bit_cast(bf16x8, P.hypb[(bk * P.hn_pad + h) * 2 + half]);
        const float2 hv = (culled ? P.hyps : P.hyp)[bk * P.hn_pad + h];   // (disc culling: hypb is in Hilbert order, and so is hyps)
        for (int tile = 0; tile < nti; ++tile) {
            const bf16x8 Ad = __builtin_bit_cast(bf16x8, lbase[tile * TILE_U4]);
            const bf16x8 Ac = __builtin_bit_cast(bf16x8, lbase[tile * TILE_U4 + 64]);
            const f32x16 vd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ad, Bc, zero, 0, 0, 0);
            const f32x16 vc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac, Bc, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r >> 2) * 8 + half * 4 + (r & 3);
                const int p = grp * 256 + tile * 32 + row;
                if (p >= tn || (culled ? P.perm[bk * P.hn_pad + h] >= P.hn : h >= P.hn)) continue;    // padding rows / columns: nobody reads their counts
                const float x = vd[r] - fabsf(vc[r]);  // (one IEEE subtraction, as v_sub_f32 x, d, |c|)
                const float4 q = s_raw[tile * 32 + row];
                const bool lit = inlier_literal(q.x, q.y, q.z, q.w, hv.x, hv.y, P.thresh);
                ++ntest;
                if (!(fabsf(x) >= BAND_CLEAN)) ++nband;
                if ((x > 0.f) != lit) {
                    ++ndis;
                    worst = fmaxf(worst, fabsf(x));
                }
            }
        }
    }
    unsigned* o = out + bk * 4;
    if (worst > 0.f) atomicMax(o, __float_as_uint(worst));   // (non-negative floats order like their bit patterns)
    if (ndis) atomicAdd(o + 1, ndis);
    if (nband) atomicAdd(o + 2, nband);
    atomicAdd(o + 3, ntest);
}

// profiling helper of pvnet_vote_v3_stage_repeat: acc[0] += (max end - min start) over the n workgroup slots
__global__ __launch_bounds__(256) void ts_collect_kernel(const unsigned long long* __restrict__ stamps, int n,
                                                         unsigned long long* __restrict__ acc, int clear) {
    PVNET_SPARE_VGPRS(39);
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int i = threadIdx.x; i < n; i += 256) {
        const unsigned long long a = stamps[2 * i], b = stamps[2 * i + 1];
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    __shared__ unsigned long long s_lo[256], s_hi[256];
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) {
            lo = s_lo[i] < lo ? s_lo[i] : lo;
            hi = s_hi[i] > hi ? s_hi[i] : hi;
        }
        const unsigned long long prev = clear ? 0ull : acc[0];
        acc[0] = prev + (hi > lo ? hi - lo : 0ull);
    }
}

// ------------------------------------------------------------------------------------------------------------
// K5: arg-max + least-squares refinement                        (ransac_voting_gpu.py:561-569, 579-595, 503-512)
// ------------------------------------------------------------------------------------------------------------
constexpr int RT = PVNET_RT;  // threads per (image, key-point) (measured: 256 -> 19 us, 1024 -> 23 us at batch 32)
constexpr int RW = RT / 64;
template <bool LITERAL>
__global__ __launch_bounds__(RT) void select_refine_kernel(VoteParams P) {
    PVNET_SPARE_VGPRS(71);  // (62 used.  Round 3: was 111 -- with 112 registers a workgroup of 8 waves needs 224 per SIMD and starts late in
                            //  the tail of another batch's scoring launch; 72: +3 % with six batches in flight)
    small_stage_prio();
    const int k = blockIdx.x, bi = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t bk = (size_t)bi * P.vn + k;
    const int nchunks = P.ctrl[bi * CTRL_STRIDE + C_NCHUNKS];
    // rows of partial counts to sum: one per chunk, or one per chunk group when the matrix-pipe kernel scored
    const int nch = (!LITERAL && P.mode) ? (nchunks + P.wg_s - 1) / P.wg_s : nchunks;
    int status = P.ctrl[bi * CTRL_STRIDE + C_STATUS];

    __shared__ unsigned long long s_best[RW];
    __shared__ double s_sum[RW][5];
    __shared__ int s_n[RW];

    if (nch == 0) {  // fewer than min_num foreground pixels: zeros (:531-534)
        if (threadIdx.x < 2) P.out[bk * 2 + threadIdx.x] = 0.f;
        if (threadIdx.x == 0) {
            if (P.status) P.status[bk] = status | PVNET_S_SKIPPED;
            P.win[bk * 2] = 0;
            P.win[bk * 2 + 1] = 0;
        }
        for (int h = threadIdx.x; h < P.hn; h += RT) P.counts[bk * P.hn_pad + h] = 0;
        return;
    }
    // ---- counts = sum over chunks; winner = first maximum (:561-562).  A thread sums two adjacent hypotheses
    // (one 32-bit load per chunk row) with eight loads in flight: the rows are latency-, not bandwidth-bound.
    unsigned long long best = 0;
    const size_t row = (size_t)(P.hn_pad >> 1);  // hn_pad is even
    if (!LITERAL && P.cull && *kp_cull_ptr(P, bk) != 0) {  // a disc-culled key-point: K4 counted in Hilbert order -- back to the caller's
                                                           // order (what every reader of `counts` expects), the first CALLER index winning ties
        for (int p = threadIdx.x; p < P.hn_pad; p += RT) {
            const int h = P.perm[bk * P.hn_pad + p];
            if (h >= P.hn) continue;   // padding
            const uint32_t c = (uint32_t)P.cnts[bk * P.hn_pad + p];
            P.counts[bk * P.hn_pad + h] = (int32_t)c;
            const unsigned long long key = ((unsigned long long)c << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)h);
            best = key > best ? key : best;
        }
    } else if (P.atomic_counts) {  // K4 already summed: one value per hypothesis
        for (int h = threadIdx.x; h < P.hn; h += RT) {
            const uint32_t c = (uint32_t)P.counts[bk * P.hn_pad + h];
            const unsigned long long key = ((unsigned long long)c << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)h);
            best = key > best ? key : best;
        }
    } else
    for (int h2 = threadIdx.x; 2 * h2 < P.hn; h2 += RT) {
        const uint32_t* pp = reinterpret_cast<const uint32_t*>(P.partial + bk * P.max_chunks * P.hn_pad) + h2;
        int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        int c = 0;
        for (; c < nch; c += 8) {  // the last trip re-reads the final row for the slots past it and discards them
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(c + u < nch ? c + u : nch - 1) * row];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t x = c + u < nch ? v[u] : 0u;
                lo[u & 3] += (int)(x & 0xFFFFu);
                hi[u & 3] += (int)(x >> 16);
            }
        }
        const int s0 = (lo[0] + lo[1]) + (lo[2] + lo[3]), s1 = (hi[0] + hi[1]) + (hi[2] + hi[3]);
        const int h = 2 * h2;
        *reinterpret_cast<int2*>(P.counts + bk * P.hn_pad + h) = make_int2(s0, s1);
        const unsigned long long k0 = ((unsigned long long)(uint32_t)s0 << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)h);
        const unsigned long long k1 =
            ((unsigned long long)(uint32_t)s1 << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(h + 1));
        best = k0 > best ? k0 : best;
        if (h + 1 < P.hn) best = k1 > best ? k1 : best;
    }
    best = wave_reduce_max(best);
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    best = s_best[0];
#pragma unroll
    for (int i = 1; i < RW; ++i) best = s_best[i] > best ? s_best[i] : best;
    const int wcnt = (int)(best >> 32);
    const int widx = (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));
    float wx = 0.f, wy = 0.f;  // all_win_pts starts at zero and only a strictly larger ratio replaces it (:548-569)
    if (wcnt > 0) {
        const float2 hv = P.hyp[bk * P.hn_pad + widx];
        wx = hv.x;
        wy = hv.y;
    } else {
        status |= PVNET_S_NO_INLIER;
    }
    if (threadIdx.x == 0) {
        P.win[bk * 2] = widx;
        P.win[bk * 2 + 1] = wcnt;
    }
    if (P.flags & PVNET_F_NO_REFINE) {
        if (threadIdx.x == 0) {
            P.out[bk * 2] = wx;
            P.out[bk * 2 + 1] = wy;
            if (P.status) P.status[bk] = status;
        }
        return;
    }
    // ---- inliers of the winner, normal equations centred on the winner, float64 (:579-594)
    const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
    const float ox = (float)P.ctrl[bi * CTRL_STRIDE + C_OX], oy = (float)P.ctrl[bi * CTRL_STRIDE + C_OY];
    double a = 0, bb = 0, d = 0, r0 = 0, r1 = 0;
    int n = 0;
#pragma unroll 4
    for (int t = threadIdx.x; t < tn; t += RT) {
        const float4 q = P.rec[bk * P.cap + t];
        const float2 u = rec_dir(q);
        bool in;
        if (LITERAL || P.exact) {  // (uniform) exact mode: the winner's inliers as the reference's own test finds them (:582-584)
            in = inlier_literal(q.x, q.y, u.x, u.y, wx, wy, P.thresh);
        } else {
            float4 ra;
            float2 rb;
            make_pixrec(q, P.tau, ox, oy, ra, rb);
            in = vote_expanded(ra, rb, wx - ox, wy - oy) > 0.5f;  // the very predicate that scored
        }
        // predicated, not branched; a select, not a product: a NaN / Inf direction never votes and must not leak
        const double nx = in ? (double)u.y : 0.0, ny = in ? -(double)u.x : 0.0;  // normal = (dy, -dx) (:580-581)
        const double bv = nx * ((double)q.x - (double)wx) + ny * ((double)q.y - (double)wy);
        a += nx * nx;
        bb += nx * ny;
        d += ny * ny;
        r0 += nx * bv;
        r1 += ny * bv;
        n += in ? 1 : 0;
    }
    a = wave_reduce_add(a);
    bb = wave_reduce_add(bb);
    d = wave_reduce_add(d);
    r0 = wave_reduce_add(r0);
    r1 = wave_reduce_add(r1);
    n = wave_reduce_add(n);
    if (lane == 0) {
        s_sum[wave][0] = a; s_sum[wave][1] = bb; s_sum[wave][2] = d; s_sum[wave][3] = r0; s_sum[wave][4] = r1;
        s_n[wave] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = bb = d = r0 = r1 = 0;
        n = 0;
        for (int i = 0; i < RW; ++i) {
            a += s_sum[i][0]; bb += s_sum[i][1]; d += s_sum[i][2]; r0 += s_sum[i][3]; r1 += s_sum[i][4];
            n += s_n[i];
        }
        const double det = a * d - bb * bb;
        float rx = wx, ry = wy;  // refined point (the winner itself when the system is singular)
        if (n == 0 || det == 0.0 || !isfinite(det)) {
            status |= PVNET_S_SINGULAR;  // torch.gesv raises here (:511); we return the winner and flag it
        } else {
            rx = (float)((double)wx + (d * r0 - bb * r1) / det);
            ry = (float)((double)wy + (a * r1 - bb * r0) / det);
        }
        P.out[bk * 2] = rx;
        P.out[bk * 2 + 1] = ry;
        if (P.status) P.status[bk] = status;
    }
}

// ------------------------------------------------------------------------------------------------------------
// epilogues of the reference's sibling functions, run on the workspace a pvnet_vote_v3 call left behind
// ------------------------------------------------------------------------------------------------------------
// ransac_voting_layer_v5's extra output (ransac_voting_gpu.py:846-850): fraction of the image's kept pixels that
// vote (literal float32 test, threshold `thresh`, 0.999 in the reference) for the given points.
__global__ __launch_bounds__(256) void confidence_kernel(VoteParams P, const float* __restrict__ pts, float thresh,
                                                         float* __restrict__ conf) {
    PVNET_SPARE_VGPRS(47);
    const int k = blockIdx.x, bi = blockIdx.y;
    const size_t bk = (size_t)bi * P.vn + k;
    if (P.ctrl[P.b * CTRL_STRIDE + 6] != P.layout_fp) {  // the workspace was written under another layout (tuning reloaded)
        if (threadIdx.x == 0) conf[bk] = __uint_as_float(0x7FC00000u);
        return;
    }
    const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
    const bool live = P.ctrl[bi * CTRL_STRIDE + C_NCHUNKS] > 0;
    const float px = pts[bk * 2], py = pts[bk * 2 + 1];
    int n = 0;
    if (live)
        for (int t = threadIdx.x; t < tn; t += 256) {
            const float4 q = P.rec[bk * P.cap + t];
            const float2 u = rec_dir(q);
            n += inlier_literal(q.x, q.y, u.x, u.y, px, py, thresh) ? 1 : 0;
        }
    n = wave_reduce_add(n);
    __shared__ int s_n[4];
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) conf[bk] = live ? (float)(s_n[0] + s_n[1] + s_n[2] + s_n[3]) / (float)tn : 0.f;
}

// estimate_voting_distribution_with_mean's epilogue (ransac_voting_gpu.py:389-404): ratio-weighted 2x2 covariance
// of the hypotheses about `mean`, weights = inlier ratio where it is within 0.1 of the key-point's best, else 0.
__global__ __launch_bounds__(256) void distribution_kernel(VoteParams P, const float* __restrict__ mean,
                                                           float* __restrict__ cov) {
    PVNET_SPARE_VGPRS(47);
    const int k = blockIdx.x, bi = blockIdx.y;
    const size_t bk = (size_t)bi * P.vn + k;
    if (P.ctrl[P.b * CTRL_STRIDE + 6] != P.layout_fp) {  // the workspace was written under another layout (tuning reloaded)
        if (threadIdx.x < 4) cov[bk * 4 + threadIdx.x] = __uint_as_float(0x7FC00000u);
        return;
    }
    const int tn = P.ctrl[bi * CTRL_STRIDE + C_TN];
    const bool live = P.ctrl[bi * CTRL_STRIDE + C_NCHUNKS] > 0;
    const float mx = mean[bk * 2], my = mean[bk * 2 + 1];
    // skipped image: the reference substitutes zero hypotheses with ratio one (:343-349)
    const float best = live ? (float)P.win[bk * 2 + 1] / (float)tn : 1.f;
    const float cut = best - 0.1f;  // :394
    double sxx = 0, sxy = 0, syy = 0, sw = 0;
    for (int h = threadIdx.x; h < P.hn; h += 256) {
        float r = 1.f, hx = 0.f, hy = 0.f;
        if (live) {
            r = (float)P.counts[bk * P.hn_pad + h] / (float)tn;  // :378-379
            const float2 hv = P.hyp[bk * P.hn_pad + h];
            hx = hv.x;
            hy = hv.y;
        }
        if (r < cut) r = 0.f;  // :395
        const double dx = (double)hx - mx, dy = (double)hy - my;
        sxx += r * dx * dx;
        sxy += r * dx * dy;
        syy += r * dy * dy;
        sw += r;
    }
    sxx = wave_reduce_add(sxx); sxy = wave_reduce_add(sxy); syy = wave_reduce_add(syy); sw = wave_reduce_add(sw);
    __shared__ double s_acc[4][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_acc[wave][0] = sxx; s_acc[wave][1] = sxy; s_acc[wave][2] = syy; s_acc[wave][3] = sw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) a[j] += s_acc[i][j];
        const double den = a[3] + 1e-3;  // :401
        cov[bk * 4 + 0] = (float)(a[0] / den);
        cov[bk * 4 + 1] = (float)(a[1] / den);
        cov[bk * 4 + 2] = (float)(a[1] / den);
        cov[bk * 4 + 3] = (float)(a[2] / den);
    }
}

// ------------------------------------------------------------------------------------------------------------
// ransac_motion_voting (ransac_voting_gpu.py:960-981): per image and key-point, the mean over the foreground pixels
// of (vertex + pixel coordinate).  Reads the field only where the bit mask of K1 is set: one block per 4096-pixel
// segment sums its pixels in float64 (lane = pixel of a 64-pixel word, so a wave reads 256 contiguous bytes per plane
// of the planar field), one block per (image, key-point) adds the segment sums in order and divides.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void motion_partial_kernel(VoteParams P, double* __restrict__ part) {
    PVNET_SPARE_VGPRS(55);
    const int sgi = blockIdx.x, bi = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (P.seg0[bi * P.nseg + sgi] == 0) return;  // block-uniform; the final kernel skips this segment's slots too
    constexpr int WPW = SEG_WORDS / 4;  // words per wave
    unsigned long long my = 0;  // bit i: this lane's pixel of the wave's word i is foreground
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int wd = (sgi * 4 + wave) * WPW + i;
        const unsigned long long bits = wd < P.words ? P.bits[(size_t)bi * P.words + wd] : 0ull;
        my |= ((bits >> lane) & 1ull) << i;
    }
    __shared__ double s_part[4][2];
    for (int k = 0; k < P.vn; ++k) {
        double sx = 0.0, sy = 0.0;
#pragma unroll 4
        for (int i = 0; i < WPW; ++i) {
            if (!((my >> i) & 1ull)) continue;
            const int p = ((sgi * 4 + wave) * WPW + i) * 64 + lane;
            const int y = p / P.w, x = p - y * P.w;
            const int64_t v = (int64_t)bi * P.vs0 + (int64_t)y * P.vs1 + (int64_t)x * P.vs2 + (int64_t)k * P.vs3;
            sx += (double)(ld_elem_rt(P.vertex_type, P.vertex, v) + (float)x);      // the reference adds in float32 (:975), then averages
            sy += (double)(ld_elem_rt(P.vertex_type, P.vertex, v + P.vs4) + (float)y);
        }
        sx = wave_reduce_add(sx);
        sy = wave_reduce_add(sy);
        if (lane == 0) { s_part[wave][0] = sx; s_part[wave][1] = sy; }
        __syncthreads();
        if (threadIdx.x < 2)
            part[(((size_t)bi * P.nseg + sgi) * P.vn + k) * 2 + threadIdx.x] =
                (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void motion_final_kernel(VoteParams P, const double* __restrict__ part,
                                                          float* __restrict__ out) {
    PVNET_SPARE_VGPRS(31);
    const int k = blockIdx.x, bi = blockIdx.y, lane = threadIdx.x;
    double sx = 0.0, sy = 0.0;
    int n = 0;
    for (int sgi = lane; sgi < P.nseg; sgi += 64) {
        const int c = P.seg0[bi * P.nseg + sgi];
        if (c == 0) continue;
        n += c;
        sx += part[(((size_t)bi * P.nseg + sgi) * P.vn + k) * 2];
        sy += part[(((size_t)bi * P.nseg + sgi) * P.vn + k) * 2 + 1];
    }
    sx = wave_reduce_add(sx);
    sy = wave_reduce_add(sy);
    n = wave_reduce_add(n);
    if (lane == 0) {  // an image without foreground returns zeros (:969-971)
        out[((size_t)bi * P.vn + k) * 2] = n ? (float)(sx / n) : 0.f;
        out[((size_t)bi * P.vn + k) * 2 + 1] = n ? (float)(sy / n) : 0.f;
    }
}


// ------------------------------------------------------------------------------------------------------------
// op-level kernels with the reference extension's layouts
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void op_generate_hypothesis_kernel(const float* __restrict__ direct,
                                                                     const float* __restrict__ coords,
                                                                     const int32_t* __restrict__ idxs,
                                                                     float* __restrict__ hyp, int tn, int vn, int hn) {
    PVNET_SPARE_VGPRS(31);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hn * vn) return;
    const int k = i % vn;
    int t0 = idxs[i * 2], t1 = idxs[i * 2 + 1];
    t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
    t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
    float ox, oy;
    hyp_intersect(direct[((size_t)t0 * vn + k) * 2], direct[((size_t)t0 * vn + k) * 2 + 1], coords[t0 * 2],
                  coords[t0 * 2 + 1], direct[((size_t)t1 * vn + k) * 2], direct[((size_t)t1 * vn + k) * 2 + 1],
                  coords[t1 * 2], coords[t1 * 2 + 1], ox, oy);
    hyp[i * 2] = ox;
    hyp[i * 2 + 1] = oy;
}

// grid (ceil(tn/256), vn, hyp-slices): lane owns a pixel, walks a slice of hypotheses (wave-uniform -> SGPRs),
// byte stores along tn are contiguous per hypothesis row.
__global__ __launch_bounds__(256) void op_voting_kernel(const float* __restrict__ direct,
                                                        const float* __restrict__ coords,
                                                        const float* __restrict__ hyp, uint8_t* __restrict__ inliers,
                                                        int tn, int vn, int hn, float thresh, int hslice) {
    PVNET_SPARE_VGPRS(31);
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    const int h0 = blockIdx.z * hslice;
    const int h1 = h0 + hslice < hn ? h0 + hslice : hn;
    if (t >= tn) return;
    const float cx = coords[t * 2], cy = coords[t * 2 + 1];
    const float nx = direct[((size_t)t * vn + k) * 2], ny = direct[((size_t)t * vn + k) * 2 + 1];
    for (int h = h0; h < h1; ++h) {
        const float hx = hyp[((size_t)h * vn + k) * 2], hy = hyp[((size_t)h * vn + k) * 2 + 1];
        if (inlier_literal(cx, cy, nx, ny, hx, hy, thresh)) inliers[((size_t)h * vn + k) * tn + t] = 1;
    }
}

// The vanishing-point pair of the reference's extension (ransac_voting_kernel.cu:170-229, :268-310): hypotheses are
// homogeneous points (x, y, z) -- the cross product of the two pixels' line coordinates, so that parallel rays give a
// point at infinity (z = 0) instead of the (0, 0) of the affine op -- and a pixel votes when |cos| of the angle
// between its direction and (h.xy - c * h.z) exceeds the threshold with both component products non-negative.
// Float32 in the reference's operation order, one rounding per operation; the `< 1e-6` gates compare in double as the
// reference's float-against-double-literal comparisons do.
__global__ __launch_bounds__(256) void op_generate_hypothesis_vp_kernel(const float* __restrict__ direct,
                                                                        const float* __restrict__ coords,
                                                                        const int32_t* __restrict__ idxs,
                                                                        float* __restrict__ hyp, int tn, int vn, int hn) {
#pragma clang fp contract(off)
    PVNET_SPARE_VGPRS(31);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hn * vn) return;
    const int k = i % vn;
    int t0 = idxs[i * 2], t1 = idxs[i * 2 + 1];
    t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
    t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
    const float dx0 = direct[((size_t)t0 * vn + k) * 2], dy0 = direct[((size_t)t0 * vn + k) * 2 + 1];
    const float dx1 = direct[((size_t)t1 * vn + k) * 2], dy1 = direct[((size_t)t1 * vn + k) * 2 + 1];
    const float cx0 = coords[t0 * 2], cy0 = coords[t0 * 2 + 1], cx1 = coords[t1 * 2], cy1 = coords[t1 * 2 + 1];
    const float lx0 = dy0, ly0 = -dx0, lz0 = cy0 * dx0 - cx0 * dy0;  // the line through c along d: l . (x, y, 1) = 0
    const float lx1 = dy1, ly1 = -dx1, lz1 = cy1 * dx1 - cx1 * dy1;
    float x = ly0 * lz1 - lz0 * ly1;
    float y = lz0 * lx1 - lx0 * lz1;
    float z = lx0 * ly1 - ly0 * lx1;
    const float vx0 = dx0 * (x - z * cx0), vx1 = dx1 * (x - z * cx1);
    const float vy0 = dy0 * (y - z * cy0), vy1 = dy1 * (y - z * cy1);
    if (vx0 < 0 && vx1 < 0 && vy0 < 0 && vy1 < 0) {  // both rays point away from the intersection: flip the point
        z = -z;
        x = -x;
        y = -y;
    }
    if (vx0 * vx1 < 0 || vy0 * vy1 < 0) x = y = z = 0.f;  // the rays do not meet
    hyp[(size_t)i * 3] = x;
    hyp[(size_t)i * 3 + 1] = y;
    hyp[(size_t)i * 3 + 2] = z;
}

// same shape as op_voting_kernel: lane owns a pixel and walks a slice of hypotheses
__global__ __launch_bounds__(256) void op_voting_vp_kernel(const float* __restrict__ direct,
                                                           const float* __restrict__ coords,
                                                           const float* __restrict__ hyp, uint8_t* __restrict__ inliers,
                                                           int tn, int vn, int hn, float thresh, int hslice) {
#pragma clang fp contract(off)
    PVNET_SPARE_VGPRS(31);
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    const int h0 = blockIdx.z * hslice;
    const int h1 = h0 + hslice < hn ? h0 + hslice : hn;
    if (t >= tn) return;
    const float cx = coords[t * 2], cy = coords[t * 2 + 1];
    const float ux = direct[((size_t)t * vn + k) * 2], uy = direct[((size_t)t * vn + k) * 2 + 1];
    const float norm1 = __builtin_sqrtf(ux * ux + uy * uy);
    if (norm1 < 1e-6) return;
    for (int h = h0; h < h1; ++h) {
        const float* hp = hyp + ((size_t)h * vn + k) * 3;
        const float hx = hp[0], hy = hp[1], hz = hp[2];
        const float dx = hx - cx * hz, dy = hy - cy * hz;
        const float norm2 = __builtin_sqrtf(dx * dx + dy * dy);
        if (norm2 < 1e-6) continue;
        const float ang = (ux * dx + uy * dy) / (norm1 * norm2);
        const float vx = dx * ux, vy = dy * uy;
        if (vx < 0 || vy < 0) continue;  // the direction is wrong (:306)
        if (fabsf(ang) > thresh) inliers[((size_t)h * vn + k) * tn + t] = 1;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// the op-level voting kernels: hypotheses per z-slice so that the launch has >= 2048 workgroups
inline void op_voting_grid(int tn, int vn, int hn, int* hslice, int* slices) {
    const int tblocks = (tn + 255) / 256;
    int n = (2048 + tblocks * vn - 1) / (tblocks * vn);
    if (n > hn) n = hn;
    if (n > 65535) n = 65535;
    if (n < 1) n = 1;
    *hslice = (hn + n - 1) / n;
    *slices = (hn + *hslice - 1) / *hslice;
}

// Host side of the exact mode's rounding band (device side: b_col_exact / a_rows_exact): half-width of the band as a
// fraction of |d| |u|, i.e. a test whose exact margin  m = tau (d . u) - |d x u|  satisfies |m| > kband |d| |u| is decided the
// same way by exact arithmetic, by the reference's float32 kernel and by the matrix pipe.
//   K_LIT  = DELTA_LIT (1 + tau^2) / tau = DELTA_LIT / (sin t0 cos t0),  DELTA_LIT = 10 u  (u = 2^-24): the reference's
//            |ang - cos| <= 8 u (dot 2 u, the two norms 2 u each, their product and the quotient 1 u each, times cos <= 1)
//            + 1 u for d = fl(h - c), rounded up; d(m / |d||u|) / d(cos) = tau + 1 / tau at the threshold;
//   K_FAST = u (1 + tau) (1.43 C_M + 8): the matrix pipe sums 16 products with |error| <= C_M u sum |terms| (measured
//            on the MI355X: 4.87, tools/ubench_exact.hip -- taken as 10), sum |terms| <= sqrt(2) (1 + 2^-7) (R + r) |M| per dot
//            product; the eight further u cover the roundings of M = u g sigma, T = tau M, T +- N, the two constants
//            Ec, Ed (2 u r |M| each), h - o, (h - o) s, tau itself and the three dropped part pairs of the bf16x3 split
//            (0.52 u).
// The terms are worst-case bounds (the matrix pipe's constant is twice what was measured): no further factor is applied,
// because the band's width is what the exact mode costs -- 2e-4 of the tests lie inside it at thresh 0.99 on the noisy
// benchmark field (tools/exact_probe.py), each flagging its cell.
float band_constant(float thresh) {
    const double u = ldexp(1.0, -24), t = (double)thresh;
    const double tau = sqrt(1.0 - t * t) / t, t0 = acos(t), delta = 10.0 * u;
    // the reference can disagree with exact arithmetic only for cos(theta) in [t - delta, t + delta]; in units of |d| |u| the
    // margin is m = tau cos(theta) - sin(theta) = sin(t0 - theta) / cos(t0): its extreme values over that interval, BOTH sides
    // (the first-order form delta / (sin t0 cos t0) is 1.5 % short on the vote side at thresh 0.99999 and
    // 20 % at 0.999999, where t0 is no longer large against the interval)
    const double lo = acos(t + delta < 1.0 ? t + delta : 1.0), hi = acos(t - delta);
    const double k_side = sin(t0 - lo) > sin(hi - t0) ? sin(t0 - lo) : sin(hi - t0);
    const double k_lit = k_side / t * 1.001;   // (cos t0 = t)
    const double k_fast = u * (1.0 + tau) * (1.43 * 10.0 + 8.0);
    return (float)(k_lit + k_fast);
}
// cell size of the exact mode: one pixel tile (16 tests per lane; vote8x_open / vote8x_close, 40 VALU operations per step)
// by default; PVNET_EXACT_FOLD=0 selects one cell per work item (36 operations per step, but a flagged cell re-evaluates
// 16 x tiles tests).  Measured at the benchmark shape (tools/exact_probe.py, profiles/r03_exact_probe.txt), item / tile
// cells: thresh 0.9 -- 135 / 137 us; 0.99 -- 169 / 158 us; 0.999 -- 313 / 184 us (the threshold angle, 2.6 degrees, sits
// inside the field's noise there: 6e-4 of the tests are re-evaluated); approximate mode on the same box: 90-102 us.
int band_fold1(int forced, float thresh) {
    (void)thresh;
    return forced == 0 ? 0 : 1;
}

// the workspace layout depends on process-wide tuning (score mode, count atomics, chunk, hpl), which
// pvnet_vote_tuning_reload() may change between a vote and an epilogue that reads the vote's workspace.  Every vote stamps
// the layout it used into ctrl's global row; the epilogue kernels compare it with the layout THEY were handed and answer NaN
// instead of reading the old workspace at new offsets.
int layout_fingerprint(const PvnetVoteLayout& L) {
    uint64_t x = 0x9E3779B97F4A7C15ull;
    const uint64_t v[] = {(uint64_t)L.chunk, (uint64_t)L.hpl, (uint64_t)L.wg_g, (uint64_t)L.reserved_, (uint64_t)L.cap,
                          (uint64_t)L.hn_pad, (uint64_t)L.off_rec, (uint64_t)L.off_hyp, (uint64_t)L.off_counts,
                          (uint64_t)L.off_win, (uint64_t)L.total_bytes, (uint64_t)L.cull, (uint64_t)L.off_perm, (uint64_t)L.off_hypc};
    for (uint64_t e : v) { x ^= e + 0x9E3779B97F4A7C15ull + (x << 6) + (x >> 2); }
    const int fp = (int)(x ^ (x >> 32));
    return fp ? fp : 1;
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

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
                        //                         pixels gathered: score_exact_kernel_cull) of 2 = the key-points K3 selects from the
                        //                         spread of their candidate intersections (the default: PVNET_CULL_DEFAULT), 1 = every
                        //                         key-point (tests, probes), 0 = none (the layout then has no culling buffers)
    int cull_q_milli;   // PVNET_CULL_Q_MILLI      the selection threshold of 2, in thousandths (kp_preamble; profiles/r06_cull_crossover.txt)
    int exact_fold;     // PVNET_EXACT_FOLD        exact mode: -1 (default) = by threshold, 0 = one cell per work item and
                        //                         hypothesis, 1 = one cell per pixel tile (band_fold1())
    int dev_stages;     // PVNET_DEV_STAGES        development aid: bit mask of the stages to launch
    int cus;            // compute units of the device (all GPUs of a node are the same part)
};
void load_tuning(Tuning& t) {
    t.score_mode = env_int("PVNET_SCORE_MODE", 1);
    t.wgs_per_cu = env_int("PVNET_SCORE_WGS_PER_CU", -1);
    t.hpl = env_int("PVNET_SCORE_HPL", -1);
    t.chunk = env_int("PVNET_SCORE_CHUNK", -1);
    t.compact_kg = env_int("PVNET_COMPACT_KG", 3);
    t.score_xcd = env_int("PVNET_SCORE_XCD", 1);
    t.score_atomic = env_int("PVNET_SCORE_ATOMIC", 1);
    t.score_lds_kb = env_int("PVNET_SCORE_LDS_KB", 0);
    t.score_acc = env_int("PVNET_SCORE_ACC", -1);
    t.exact_fold = env_int("PVNET_EXACT_FOLD", -1);
    t.score_runs = env_int("PVNET_SCORE_RUNS", -1);
    t.score_cull = env_int("PVNET_SCORE_CULL", -1);
    t.cull_q_milli = env_int("PVNET_CULL_Q_MILLI", PVNET_CULL_Q_MILLI);
    t.dev_stages = env_int("PVNET_DEV_STAGES", 0x3F);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
    t.cus = n;
}
Tuning& tuning() {
    static Tuning t = [] { Tuning x; load_tuning(x); return x; }();  // thread-safe one-time initialisation
    return t;
}

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

template <bool LITERAL>
int launch_score(const VoteParams& P, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)P.wg_s * P.chunk * (sizeof(float4) + sizeof(float2));
    switch (P.hpl) {
        case 1: hipLaunchKernelGGL((score_kernel<1, LITERAL>), grid, dim3(256), lds, s, P); break;
        case 2: hipLaunchKernelGGL((score_kernel<2, LITERAL>), grid, dim3(256), lds, s, P); break;
        case 4: hipLaunchKernelGGL((score_kernel<4, LITERAL>), grid, dim3(256), lds, s, P); break;
        case 8: hipLaunchKernelGGL((score_kernel<8, LITERAL>), grid, dim3(256), lds, s, P); break;
        default: return PVNET_E_UNSUPPORTED;
    }
    return 0;
}

int launch_mask_bits(const VoteParams& P, hipStream_t s) {
    dim3 grid(P.nseg, P.b);
    switch (P.mask_dtype) {
        case PVNET_MASK_U8: hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_U8>, grid, dim3(64 * K1_WAVES), 0, s, P); break;
        case PVNET_MASK_I16: hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_I16>, grid, dim3(64 * K1_WAVES), 0, s, P); break;
        case PVNET_MASK_I32: hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_I32>, grid, dim3(64 * K1_WAVES), 0, s, P); break;
        case PVNET_MASK_I64:
            // contiguous images at 16-byte aligned addresses, an even number of pixels: two pixels per 16-byte load
            if (P.mask_linear && (reinterpret_cast<uintptr_t>(P.mask) & 15u) == 0 && (P.ms0 & 1) == 0 && (P.npix & 1) == 0)
                hipLaunchKernelGGL(mask_bits_pair_kernel, grid, dim3(64 * K1_WAVES), 0, s, P);
            else
                hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_I64>, grid, dim3(64 * K1_WAVES), 0, s, P);
            break;
        case PVNET_MASK_F32: hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_F32>, grid, dim3(64 * K1_WAVES), 0, s, P); break;
        case PVNET_MASK_LOGITS_F32: hipLaunchKernelGGL(mask_bits_kernel<PVNET_MASK_LOGITS_F32>, grid, dim3(64 * K1_WAVES), 0, s, P); break;
        default: return PVNET_E_BADARG;
    }
    return 0;
}

int launch_all(const VoteParams& P, hipStream_t s, hipEvent_t* ev, int stage_mask = -1, bool timed_score = false,
               int* score_grid = nullptr) {
    const bool literal = (P.flags & PVNET_F_LITERAL) != 0;
    auto mark = [&](int i) -> hipError_t { return ev ? hipEventRecord(ev[i], s) : hipSuccess; };
    // bit i set = launch stage i (K1, -, K2, K3, K4, K5; slot 1 is empty since round 2); a workspace left by a complete call stays valid, so single
    // stages can be re-run on it in isolation (pvnet_vote_v3_stage_repeat; development aid: PVNET_DEV_STAGES)
    const Tuning& T = tuning();
    const int stages = stage_mask >= 0 ? stage_mask : T.dev_stages;
    PV_HIP(mark(0));
    {   // K1
        if (stages & 1) {
            int rc = launch_mask_bits(P, s);
            if (rc) return rc;
        }
        PV_LAUNCH_CHECK();
        PV_HIP(mark(1));
        // (stage slot 1 was the thinning launch of round 1: the mask kernel's histograms + the compaction kernel do it now)
        PV_HIP(mark(2));
    }
    if (stages & 4) {   // K2
        const int kg = T.compact_kg;
        dim3 grid(P.nseg, P.b, (P.vn + kg - 1) / kg);
        const dim3 g3(P.nseg, P.b, (P.vn + 2) / 3);
#define PV_K2(VT)                                                                                                  \
    do {                                                                                                           \
        if (literal || P.exact) hipLaunchKernelGGL((compact_kernel<true, 3, VT>), g3, dim3(256), 0, s, P);         \
        else if (kg == 1) hipLaunchKernelGGL((compact_kernel<false, 1, VT>), grid, dim3(256), 0, s, P);            \
        else if (kg == 9) hipLaunchKernelGGL((compact_kernel<false, 9, VT>), grid, dim3(256), 0, s, P);            \
        else hipLaunchKernelGGL((compact_kernel<false, 3, VT>), g3, dim3(256), 0, s, P);                           \
    } while (0)
        if (P.vertex_type == VT_F16) PV_K2(VT_F16);
        else if (P.vertex_type == VT_BF16) PV_K2(VT_BF16);
        else PV_K2(VT_F32);
#undef PV_K2
        PV_LAUNCH_CHECK();
    }
    PV_HIP(mark(3));
    if (stages & 8) {   // K3
        // per image: the hypothesis blocks, the plan block and -- where key-points may be disc-culled -- one block per key-point
        dim3 grid((unsigned)(((P.hn * P.vn + 255) / 256 + 1 + (!literal && P.cull ? P.vn : 0)) * ((P.b + 7) / 8) * 8));
        if (literal) hipLaunchKernelGGL(hypothesis_kernel<true>, grid, dim3(256), 0, s, P);
        else hipLaunchKernelGGL(hypothesis_kernel<false>, grid, dim3(256), 0, s, P);
        PV_LAUNCH_CHECK();
    }
    PV_HIP(mark(4));
    if (stages & 16) {   // K4: persistent grid, work items strided over its waves
        const long long max_items =
            (long long)P.b * P.vn * (P.hgroups / P.wg_g) * ((P.max_chunks + P.wg_s - 1) / P.wg_s);
        // (12: the four-waves-per-SIMD scoring kernel of a batch alone, three rounds of four resident workgroups; measured with it only)
        const bool alone8 = P.exact && P.wg_g * P.hpl / 2 == 8 && !(P.flags & PVNET_F_CONCURRENT) && T.score_acc != 2 && T.score_runs != 1;
        const int wgs_per_cu = T.wgs_per_cu >= 0 ? T.wgs_per_cu : (alone8 ? 12 : 8);
        long long wgs = wgs_per_cu > 0 ? (long long)T.cus * wgs_per_cu : max_items;  // 0: one workgroup per item
        if (wgs > max_items) wgs = max_items;
        if (wgs < 1) wgs = 1;
        if (score_grid) *score_grid = (int)wgs;
        if (P.exact && P.cull) {
            // key-points may be disc-culled (K3 decides; PVNET_SCORE_CULL=1: all of them): ONE launch scores both kinds of item
            // (score_exact_kernel_both_*), with the dense kernel's registers, LDS and grid: 12 workgroups per CU queued for a batch
            // alone (four resident, three rounds), 8 beside other batches
            const bool conc = (P.flags & PVNET_F_CONCURRENT) != 0;
            const bool runs = T.score_runs == 1 || (T.score_runs < 0 && conc);
            long long w2 = T.wgs_per_cu >= 0 ? wgs : (long long)T.cus * (conc ? 8 : 12);
            if (w2 > max_items) w2 = max_items;
            if (w2 < 1) w2 = 1;
            if (score_grid) *score_grid = (int)w2;
            const dim3 g((unsigned)w2), t(256);
            if (timed_score) {
                if (runs) hipLaunchKernelGGL(score_exact_kernel_both_1_1, g, t, CULL_LDS_BYTES, s, P);
                else hipLaunchKernelGGL(score_exact_kernel_both_1_0, g, t, CULL_LDS_BYTES, s, P);
            } else {
                if (runs) hipLaunchKernelGGL(score_exact_kernel_both_0_1, g, t, CULL_LDS_BYTES, s, P);
                else hipLaunchKernelGGL(score_exact_kernel_both_0_0, g, t, CULL_LDS_BYTES, s, P);
            }
        } else if (P.exact) {
            const int mh = P.wg_g * P.hpl / 2;
            const int npx = P.wg_s * P.chunk;
            size_t lds = (size_t)(npx / 32) * TILE_U4 * sizeof(uint4) + (size_t)npx * sizeof(float4) +
                         (size_t)4 * mh * 32 * sizeof(float2) + (size_t)4 * mh * 64 * sizeof(unsigned);
            if (T.score_lds_kb > 0 && T.score_lds_kb <= 64 && lds < (size_t)T.score_lds_kb * 1024) lds = (size_t)T.score_lds_kb * 1024;
            const dim3 g((unsigned)wgs), t(256);
            const int fold = P.fold1;
            // calls flagged PVNET_F_CONCURRENT (other batches in flight): contiguous runs, one accumulator pair, three waves per
            // SIMD (136 VGPRs; four cost 6 % there: profiles/r04_ab_runs.txt); a batch alone: strided items, one pair in 128
            // VGPRs = four waves per SIMD (kernel -2 %), 12 workgroups per CU.  PVNET_SCORE_ACC=2 / PVNET_SCORE_RUNS force the
            // round-3 form (two pairs, 168 VGPRs) / either mapping; runs need cells of one pixel tile.
            const bool conc = (P.flags & PVNET_F_CONCURRENT) != 0;
            const bool one_acc = T.score_acc == 1 || T.score_acc < 0;
            const bool runs = T.score_runs == 1 || (T.score_runs < 0 && conc);  // (cells of a whole item: the same 136-VGPR kernel, strided items)
#define PV_EXACT3(MH_, NACC_, RUNS_)                                                                                \
    do {                                                                                                            \
        if (timed_score) {                                                                                          \
            if (fold == 1) hipLaunchKernelGGL((ScoreExact<MH_, 1, true, NACC_, RUNS_>::kernel), g, t, lds, s, P);   \
            else hipLaunchKernelGGL((ScoreExact<MH_, 0, true, NACC_, RUNS_>::kernel), g, t, lds, s, P);             \
        } else {                                                                                                    \
            if (fold == 1) hipLaunchKernelGGL((ScoreExact<MH_, 1, false, NACC_, RUNS_>::kernel), g, t, lds, s, P);  \
            else hipLaunchKernelGGL((ScoreExact<MH_, 0, false, NACC_, RUNS_>::kernel), g, t, lds, s, P);            \
        }                                                                                                           \
    } while (0)
#define PV_EXACT(MH_, NACC_) PV_EXACT3(MH_, NACC_, false)
            switch (mh) {
                case 1: PV_EXACT(1, 2); break;
                case 2: PV_EXACT(2, 2); break;
                case 4: PV_EXACT(4, 2); break;
                case 8:
                    if (runs) { if (one_acc) PV_EXACT3(8, 1, true); else PV_EXACT3(8, 2, true); }
                    else { if (one_acc) PV_EXACT(8, 1); else PV_EXACT(8, 2); }
                    break;
                default: return PVNET_E_UNSUPPORTED;
            }
#undef PV_EXACT3
#undef PV_EXACT
        } else if (!literal && P.mode) {
            const int mh = P.wg_g * P.hpl / 2;  // hypotheses per item = wg_g * 64 * hpl = 4 waves * mh * 32
            size_t lds = (size_t)(P.wg_s * P.chunk / 32) * TILE_U4 * sizeof(uint4);
            if (T.score_lds_kb > 0 && T.score_lds_kb <= 64 && lds < (size_t)T.score_lds_kb * 1024) lds = (size_t)T.score_lds_kb * 1024;
            const dim3 g((unsigned)wgs), t(256);
            if (timed_score) {  // same code + two clock stamps per workgroup (pvnet_vote_v3_stage_repeat)
                switch (mh) {
                    case 1: hipLaunchKernelGGL((score_mfma_kernel<1, true>), g, t, lds, s, P); break;
                    case 2: hipLaunchKernelGGL((score_mfma_kernel<2, true>), g, t, lds, s, P); break;
                    case 4: hipLaunchKernelGGL((score_mfma_kernel<4, true>), g, t, lds, s, P); break;
                    case 8: hipLaunchKernelGGL((score_mfma_kernel<8, true>), g, t, lds, s, P); break;
                    default: return PVNET_E_UNSUPPORTED;
                }
            } else {
                switch (mh) {
                    case 1: hipLaunchKernelGGL((score_mfma_kernel<1, false>), g, t, lds, s, P); break;
                    case 2: hipLaunchKernelGGL((score_mfma_kernel<2, false>), g, t, lds, s, P); break;
                    case 4: hipLaunchKernelGGL((score_mfma_kernel<4, false>), g, t, lds, s, P); break;
                    case 8: hipLaunchKernelGGL((score_mfma_kernel<8, false>), g, t, lds, s, P); break;
                    default: return PVNET_E_UNSUPPORTED;
                }
            }
        } else {
            int rc = literal ? launch_score<true>(P, dim3((unsigned)wgs), s) : launch_score<false>(P, dim3((unsigned)wgs), s);
            if (rc) return rc;
        }
        PV_LAUNCH_CHECK();
        PV_LAUNCH_CHECK();
    }
    PV_HIP(mark(5));
    if (stages & 32) {   // K5
        dim3 grid(P.vn, P.b);
        if (literal) hipLaunchKernelGGL(select_refine_kernel<true>, grid, dim3(RT), 0, s, P);
        else hipLaunchKernelGGL(select_refine_kernel<false>, grid, dim3(RT), 0, s, P);
        PV_LAUNCH_CHECK();
    }
    PV_HIP(mark(6));
    return 0;
}

int fill_params(VoteParams& P, const void* mask, int mask_dtype, const int64_t* ms, const float* vertex,
                const int64_t* vs, int b, int h, int w, int vn, int hn, float thresh, int min_num, int max_num,
                uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags, float* out, int32_t* status,
                void* ws, size_t ws_bytes) {
    if (!mask || !vertex || !ms || !vs || !out || !ws) return PVNET_E_BADARG;
    if (mask_dtype < PVNET_MASK_U8 || mask_dtype > PVNET_MASK_LOGITS_F32) return PVNET_E_BADARG;
    PvnetVoteLayout L;
    int rc = pvnet_vote_layout(b, h, w, vn, hn, max_num, &L);
    if (rc) return rc;
    if (ws_bytes < L.total_bytes) return PVNET_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(ws) & 255u) != 0) return PVNET_E_BADARG;
    // the sqrt-free predicate folds 1/thresh into the records: needs thresh > 0; otherwise score literally
    // (and tau = sqrt(1 - t^2) / t below ~1e3, or the scaled matrix operands leave float32's range)
    if (!(thresh >= 1e-3f && thresh < 1.f)) flags |= PVNET_F_LITERAL;
    char* base = static_cast<char*>(ws);
    P.mask = mask; P.ms0 = ms[0]; P.ms1 = ms[1]; P.ms2 = ms[2];
    P.ms_c = 0; P.num_classes = 1;  // only the logits entry point sets these
    P.mask_dtype = mask_dtype;
    P.mask_linear = (ms[2] == 1 && ms[1] == w) ? 1 : 0;
    P.vertex = vertex; P.vs0 = vs[0]; P.vs1 = vs[1]; P.vs2 = vs[2]; P.vs3 = vs[3]; P.vs4 = vs[4];
    if ((flags & PVNET_F_VERTEX_F16) && (flags & PVNET_F_VERTEX_BF16)) return PVNET_E_BADARG;
    if ((flags & PVNET_F_LOGITS_F16) && (flags & PVNET_F_LOGITS_BF16)) return PVNET_E_BADARG;
    P.vertex_type = (flags & PVNET_F_VERTEX_F16) ? VT_F16 : (flags & PVNET_F_VERTEX_BF16) ? VT_BF16 : VT_F32;
    P.logits_type = (flags & PVNET_F_LOGITS_F16) ? VT_F16 : (flags & PVNET_F_LOGITS_BF16) ? VT_BF16 : VT_F32;
    P.b = b; P.h = h; P.w = w; P.vn = vn; P.hn = hn; P.npix = h * w;
    P.words = L.words; P.cap = L.cap; P.chunk = L.chunk; P.max_chunks = L.max_chunks;
    P.hpl = L.hpl; P.hgroups = L.hgroups; P.hn_pad = L.hn_pad; P.wg_g = L.wg_g; P.wg_s = L.wg_s;
    P.mode = L.reserved_;
    P.layout_fp = layout_fingerprint(L);
    P.score_xcd = tuning().score_xcd;
    P.atomic_counts = tuning().score_atomic;
    P.thresh = thresh;
    P.tau = (thresh > 0.f && thresh < 1.f) ? (float)(sqrt(1.0 - (double)thresh * thresh) / (double)thresh) : 0.f;
    // exact mode (the default): matrix-pipe scoring + literal re-evaluation inside the rounding band; needs the B-operand
    // buffer (PVNET_SCORE_MODE=1) and counts by atomics (the re-evaluated cells add theirs the same way)
    //  without the B-operand buffer (PVNET_SCORE_MODE=0) the default mode cannot run on the matrix pipe: it is
    // scored literally -- the same counts -- instead of silently falling back to the approximate VALU predicate
    if (!(flags & (PVNET_F_LITERAL | PVNET_F_APPROX)) && !L.reserved_) flags |= PVNET_F_LITERAL;
    P.exact = (!(flags & (PVNET_F_LITERAL | PVNET_F_APPROX)) && L.reserved_) ? 1 : 0;
    P.kband = P.exact ? band_constant(thresh) : 0.f;
    P.fold1 = P.exact ? band_fold1(tuning().exact_fold, thresh) : 0;
    //  cells of one pixel tile list a flagged (hypothesis, half-wave) as a tile mask above 11 index bits: 21 tiles;
    // work items of more tiles (PVNET_SCORE_CHUNK 192 ... 480) use the cell = work item form, which has no such limit
    if (P.exact && P.fold1 && (L.wg_s * L.chunk) / 32 > 21) P.fold1 = 0;
    if (P.exact) P.atomic_counts = 1;
    P.min_num = min_num; P.max_num = max_num; P.seed = seed; P.image_base = image_base; P.idxs = idxs; P.flags = flags;
    P.ctrl = reinterpret_cast<int32_t*>(base + L.off_ctrl);
    P.items = reinterpret_cast<int4*>(base + L.off_items);
    P.seg = reinterpret_cast<int32_t*>(base + L.off_seg);
    P.seg0 = P.seg + (size_t)L.b * L.nseg;
    P.cum = max_num < h * (long long)w
                ? reinterpret_cast<uint16_t*>(base + L.off_seg + align_up(sizeof(int32_t) * 2 * (size_t)L.b * L.nseg, 16))
                : nullptr;
    P.nseg = L.nseg;
    P.bits = reinterpret_cast<uint64_t*>(base + L.off_bits);
    P.pix = reinterpret_cast<int32_t*>(base + L.off_pix);
    P.rec = reinterpret_cast<float4*>(base + L.off_rec);
    P.hyp = reinterpret_cast<float2*>(base + L.off_hyp);
    P.hypb = reinterpret_cast<uint4*>(base + L.off_hypb);
    P.partial = reinterpret_cast<uint16_t*>(base + L.off_partial);
    P.counts = reinterpret_cast<int32_t*>(base + L.off_counts);
    P.win = reinterpret_cast<int32_t*>(base + L.off_win);
    P.out = out; P.status = status;
    // disc culling: the layout has its buffers, the call runs the exact mode with cells of one pixel tile; every key-point
    // (PVNET_SCORE_CULL=1) or the ones K3 selects (2: the default)
    const int cull_knob = tuning().score_cull >= 0 ? tuning().score_cull : PVNET_CULL_DEFAULT;
    P.cull = (L.cull && P.exact && P.fold1 && vn <= KP_MAX) ? (cull_knob == 1 ? 1 : 2) : 0;
    P.cull_q = 1e-3f * (float)tuning().cull_q_milli;
    P.perm = reinterpret_cast<int32_t*>(base + L.off_perm);
    P.hyps = reinterpret_cast<float2*>(base + L.off_hyps);
    P.cnts = reinterpret_cast<int32_t*>(base + L.off_cnts);
    P.hypc = reinterpret_cast<uint4*>(base + L.off_hypc);
    P.hypg = reinterpret_cast<float*>(base + L.off_hypc + align_up(sizeof(uint4) * 2 * (size_t)b * vn * (L.hn_pad / 32), 256));
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int pvnet_vote_abi_version(void) { return PVNET_VOTE_ABI_VERSION; }
const char* pvnet_vote_build_info(void) { return "pvnet_vote gfx950 hip " __DATE__ " " __TIME__; }
void pvnet_vote_tuning_reload(void) { load_tuning(tuning()); }

int pvnet_vote_layout(int b, int h, int w, int vn, int hn, int max_num, PvnetVoteLayout* L) {
    if (!L || b <= 0 || h <= 0 || w <= 0 || vn <= 0 || hn <= 0 || max_num < 0) return PVNET_E_BADARG;
    if ((long long)h * w > (1ll << 30) || b > 65535 || vn > 65535 || hn > (1 << 20)) return PVNET_E_UNSUPPORTED;
    if ((long long)hn * vn > (1ll << 24)) return PVNET_E_UNSUPPORTED;  // grid sizes and 32-bit indices
    const long long npix = (long long)h * w;
    long long cap = npix;
    if (max_num < npix) {  // tn ~ Binomial(tn0, p'), p' = max_num / tn0 rounded up to the next bin edge (pvnet_thin_bin): mean <
                           // max_num + tn0 / 1024 where the steps are 1/1024, < max_num 17/16 below (there max_num / 16 < tn0 / 1024); 8 sigma
        const long long mean = (long long)max_num + (npix + 1023) / 1024;
        const long long c = mean + 8ll * (long long)ceil(sqrt((double)mean)) + 64;
        cap = c < npix ? c : npix;
    }
    cap = (cap + PAD - 1) / PAD * PAD + PAD;
    const Tuning& T = tuning();
    const int mode = T.score_mode;  // 1: matrix-pipe scoring in fast mode, 0: VALU scoring
    int hpl = hn >= 768 ? 8 : (hn >= 384 ? 4 : (hn >= 128 || mode ? 2 : 1));  // tuned at hn = 1024 (profiles/r01_tune13)
    if (T.hpl >= 0) hpl = T.hpl;
    if (hpl != 1 && hpl != 2 && hpl != 4 && hpl != 8) return PVNET_E_UNSUPPORTED;
    if (mode && hpl == 1) return PVNET_E_UNSUPPORTED;  // a matrix-pipe work item holds >= 128 hypotheses
    int hgroups = (hn + 64 * hpl - 1) / (64 * hpl);
    // a scoring workgroup (4 waves) covers wg_g hypothesis groups x wg_s chunks of one (image, key-point)
    const int wg_g = mode ? (hgroups >= 2 ? 2 : 1) : (hgroups >= 3 ? 4 : hgroups);
    hgroups = (hgroups + wg_g - 1) / wg_g * wg_g;

    const long long units = (long long)b * vn * hgroups;
    int chunk = units >= 128 ? 128 : 64;
    // disc culling works on 256-pixel items (two chunks of 128); PVNET_SCORE_CULL=1 (every key-point culled: tests, probes) gives
    // small batches that shape too, the default (2: K3 selects) leaves their layout alone
    const int cull_knob = T.score_cull >= 0 ? T.score_cull : PVNET_CULL_DEFAULT;
    if (cull_knob == 1 && mode && T.score_atomic && wg_g * hpl / 2 == 8 && hgroups * 64 * hpl == CULL_HN) chunk = CULL_NPX / (4 / wg_g);
    if (T.chunk >= 0) chunk = T.chunk;
    if (chunk < PAD || chunk % PAD != 0 || chunk > 1024) return PVNET_E_UNSUPPORTED;  // LDS: 4 * 1024 * 32 B
    if (mode && chunk % 32 != 0) return PVNET_E_UNSUPPORTED;  // whole 32-pixel MFMA tiles
    // a matrix-pipe work item is (4 / wg_g) * chunk pixels; its wrapped vote accumulators hold 16 votes per 32-pixel tile
    if (mode && (4 / wg_g) * chunk / 2 >= VOTE_WRAP) return PVNET_E_UNSUPPORTED;
    L->b = b; L->h = h; L->w = w; L->vn = vn; L->hn = hn;
    L->cap = (int)cap;
    L->words = (int)((npix + 63) / 64);
    L->chunk = chunk;
    L->max_chunks = (int)((cap + chunk - 1) / chunk);
    L->hpl = hpl;
    L->hgroups = hgroups;
    L->hn_pad = hgroups * 64 * hpl;
    L->wg_g = wg_g;
    L->wg_s = 4 / wg_g;
    L->reserved_ = mode ? 1 : 0;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L->nseg = (L->words + SEG_WORDS - 1) / SEG_WORDS;
    // ctrl rows [b + 1][8], then the exact mode's band origins int32 [b][vn][2] (band_origin_ptr()), then which key-points are
    // disc-culled int32 [b][vn] (kp_cull_ptr()), then the call's flags int32 [8] (call_flags_ptr())
    L->off_ctrl = take(sizeof(int32_t) * (CTRL_STRIDE * (size_t)(b + 1) + 3 * (size_t)b * vn + 8));
    // [2][b][nseg] int32 (the second array holds the mask's segment counts; the first is unused since round 2), then,
    // when thinning is possible (max_num < h*w), the segments' cumulative histograms uint16 [b][nseg][THIN_BINS]
    L->off_seg = take(align_up(sizeof(int32_t) * 2 * (size_t)b * L->nseg, 16) +
                      (max_num < npix ? sizeof(uint16_t) * THIN_BINS * (size_t)b * L->nseg : 0));
    L->off_items = take(sizeof(int32_t) * 4 * (size_t)b * vn * (hgroups / wg_g) *
                        (size_t)((L->max_chunks + L->wg_s - 1) / L->wg_s));
    L->off_bits = take(sizeof(uint64_t) * (size_t)b * L->words);
    L->off_pix = take(sizeof(int32_t) * (size_t)b * cap);
    L->off_rec = take(sizeof(float) * 4 * (size_t)b * vn * cap);
    L->off_hyp = take(sizeof(float) * 2 * (size_t)b * vn * L->hn_pad);
    L->off_hypb = take(mode ? sizeof(uint4) * 2 * (size_t)b * vn * L->hn_pad : 0);
    // per-chunk count rows exist only when K4 does not add into `counts` directly (PVNET_SCORE_ATOMIC=0)
    L->off_partial = take(T.score_atomic ? 0 : sizeof(uint16_t) * (size_t)b * vn * L->max_chunks * L->hn_pad);
    L->off_counts = take(sizeof(int32_t) * (size_t)b * vn * L->hn_pad);
    L->off_win = take(sizeof(int32_t) * 2 * (size_t)b * vn);
    // disc culling (exact mode): 8 hypothesis tiles per wave, 256-pixel work items, one slice of 1 024 hypotheses per key-point (four
    // sort keys per thread of a K3 block), at most KP_MAX key-points (the origin estimate's arrays)
    L->cull = (cull_knob && mode && T.score_atomic && wg_g * hpl / 2 == 8 && L->wg_s * chunk == CULL_NPX && L->hn_pad == CULL_HN && vn <= KP_MAX) ? 1 : 0;
    L->off_perm = take(L->cull ? sizeof(int32_t) * (size_t)b * vn * L->hn_pad : 0);
    L->off_hyps = take(L->cull ? sizeof(float) * 2 * (size_t)b * vn * L->hn_pad : 0);
    L->off_cnts = take(L->cull ? sizeof(int32_t) * (size_t)b * vn * L->hn_pad : 0);
    // tile centres uint4 [b][vn][hn_pad / 32][2], then their g float [b][vn][hn_pad / 32]
    L->off_hypc = take(L->cull ? (sizeof(uint4) * 2 + sizeof(float)) * (size_t)b * vn * (L->hn_pad / 32) + 256 : 0);
    L->total_bytes = off;
    return 0;
}

size_t pvnet_vote_workspace_bytes(int b, int h, int w, int vn, int hn, int max_num) {
    PvnetVoteLayout L;
    return pvnet_vote_layout(b, h, w, vn, hn, max_num, &L) == 0 ? L.total_bytes : 0;
}

int pvnet_vote_v3(const void* mask, int mask_dtype, const int64_t mask_strides[3], const float* vertex,
                  const int64_t vertex_strides[5], int b, int h, int w, int vn, int hn, float inlier_thresh,
                  int min_num, int max_num, uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                  float* out_kpts, int32_t* out_status, void* workspace, size_t workspace_bytes, void* stream) {
    VoteParams P;
    int rc = fill_params(P, mask, mask_dtype, mask_strides, vertex, vertex_strides, b, h, w, vn, hn, inlier_thresh,
                         min_num, max_num, seed, image_base, idxs, flags, out_kpts, out_status, workspace,
                         workspace_bytes);
    if (rc) return rc;
    return launch_all(P, static_cast<hipStream_t>(stream), nullptr);
}

int pvnet_vote_v3_logits(const float* seg_pred, const int64_t seg_strides[4], int num_classes, const float* vertex,
                         const int64_t vertex_strides[5], int b, int h, int w, int vn, int hn, float inlier_thresh,
                         int min_num, int max_num, uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                         float* out_kpts, int32_t* out_status, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (!seg_strides || num_classes < 1) return PVNET_E_BADARG;
    const int64_t ms[3] = {seg_strides[0], seg_strides[2], seg_strides[3]};  // (b, y, x); class stride separately
    VoteParams P;
    int rc = fill_params(P, seg_pred, PVNET_MASK_LOGITS_F32, ms, vertex, vertex_strides, b, h, w, vn, hn,
                         inlier_thresh, min_num, max_num, seed, image_base, idxs, flags, out_kpts, out_status,
                         workspace, workspace_bytes);
    if (rc) return rc;
    P.ms_c = seg_strides[1];
    P.num_classes = num_classes;
    return launch_all(P, static_cast<hipStream_t>(stream), nullptr);
}

int pvnet_vote_v3_profiled(const void* mask, int mask_dtype, const int64_t mask_strides[3], const float* vertex,
                           const int64_t vertex_strides[5], int b, int h, int w, int vn, int hn,
                           float inlier_thresh, int min_num, int max_num, uint64_t seed, int image_base,
                           const int32_t* idxs, uint32_t flags, float* out_kpts, int32_t* out_status,
                           void* workspace, size_t workspace_bytes, void* stream, float* stage_ms) {
    if (!stage_ms) return PVNET_E_BADARG;
    VoteParams P;
    int rc = fill_params(P, mask, mask_dtype, mask_strides, vertex, vertex_strides, b, h, w, vn, hn, inlier_thresh,
                         min_num, max_num, seed, image_base, idxs, flags, out_kpts, out_status, workspace,
                         workspace_bytes);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t ev[PVNET_NUM_STAGES + 1];
    int created = 0;
    for (; created <= PVNET_NUM_STAGES; ++created)
        if (hipEventCreate(&ev[created]) != hipSuccess) break;
    if (created <= PVNET_NUM_STAGES) {
        for (int i = 0; i < created; ++i) (void)hipEventDestroy(ev[i]);
        return (int)hipErrorOutOfMemory;
    }
    rc = launch_all(P, s, ev);
    hipError_t e = hipStreamSynchronize(s);
    if (rc == 0 && e != hipSuccess) rc = (int)e;
    if (rc == 0)
        for (int i = 0; i < PVNET_NUM_STAGES; ++i) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            stage_ms[i] = (e == hipSuccess) ? ms : -1.f;
            if (i == PVNET_STAGE_SUBSAMPLE) stage_ms[i] = 0.f;  // an empty slot since ABI 5 (two event records back to back)
        }
    for (int i = 0; i <= PVNET_NUM_STAGES; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

int pvnet_vote_v3_stage_repeat(const void* mask, int mask_dtype, const int64_t mask_strides[3], const float* vertex,
                               const int64_t vertex_strides[5], int b, int h, int w, int vn, int hn,
                               float inlier_thresh, int min_num, int max_num, uint64_t seed, int image_base,
                               const int32_t* idxs, uint32_t flags, float* out_kpts, int32_t* out_status,
                               void* workspace, size_t workspace_bytes, void* stream, int stage, int repeats,
                               float* avg_ms) {
    if (!avg_ms || stage < 0 || stage >= PVNET_NUM_STAGES || repeats < 1) return PVNET_E_BADARG;
    VoteParams P;
    int rc = fill_params(P, mask, mask_dtype, mask_strides, vertex, vertex_strides, b, h, w, vn, hn, inlier_thresh,
                         min_num, max_num, seed, image_base, idxs, flags, out_kpts, out_status, workspace,
                         workspace_bytes);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t ev[2];
    if (hipEventCreate(&ev[0]) != hipSuccess) return (int)hipErrorOutOfMemory;
    if (hipEventCreate(&ev[1]) != hipSuccess) { (void)hipEventDestroy(ev[0]); return (int)hipErrorOutOfMemory; }
    rc = launch_all(P, s, nullptr, 0x3F);  // one complete pass: the workspace now holds what every stage consumes
    if (rc == 0) rc = (int)hipEventRecord(ev[0], s);
    for (int i = 0; rc == 0 && i < repeats; ++i) rc = launch_all(P, s, nullptr, 1 << stage);
    if (rc == 0) rc = (int)hipEventRecord(ev[1], s);
    // the matrix-pipe scoring kernel once more, `repeats` times, stamping the device clock itself (fast mode only)
    // ticks accumulate in the spare words of ctrl's global row; the stamps live in `pix` (consumed by K3 only; a later
    // complete call rewrites it), when the scoring grid's slots fit there
    const int wgs_cu = tuning().wgs_per_cu >= 0 ? tuning().wgs_per_cu : 12;
    const long long score_wgs = wgs_cu > 0 ? (long long)tuning().cus * wgs_cu : (1ll << 40);
    const bool device_clock = stage == PVNET_STAGE_SCORE && !(P.flags & PVNET_F_LITERAL) && P.mode &&
                              score_wgs * 48 <= (long long)sizeof(int32_t) * P.b * P.cap;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(P.ctrl + P.b * CTRL_STRIDE + 2);
    if (rc == 0 && device_clock) {
        for (int i = 0; rc == 0 && i < repeats; ++i) {
            int grid = 0;
            rc = launch_all(P, s, nullptr, 1 << stage, true, &grid);
            if (rc == 0)
                hipLaunchKernelGGL(ts_collect_kernel, dim3(1), dim3(256), 0, s,
                                   reinterpret_cast<const unsigned long long*>(P.pix), grid, acc, i == 0 ? 1 : 0);
        }
        if (rc == 0) rc = (int)hipGetLastError();
    }
    hipError_t e = hipStreamSynchronize(s);
    if (rc == 0 && e != hipSuccess) rc = (int)e;
    if (rc == 0) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (e != hipSuccess) rc = (int)e;
        avg_ms[0] = avg_ms[1] = ms / (float)repeats;
    }
    if (rc == 0 && device_clock) {
        unsigned long long ticks = 0;
        int khz = 0, dev = 0;
        e = hipMemcpy(&ticks, acc, sizeof(ticks), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
        if (e != hipSuccess) rc = (int)e;
        else if (khz > 0 && ticks > 0) avg_ms[0] = (float)((double)ticks / (double)khz / (double)repeats);
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return rc;
}

static int params_for_workspace(VoteParams& P, int b, int h, int w, int vn, int hn, int max_num, void* ws,
                                size_t ws_bytes) {
    static const int64_t ms[3] = {0, 0, 1}, vs[5] = {0, 0, 0, 0, 1};
    static float dummy;
    return fill_params(P, &dummy, PVNET_MASK_U8, ms, &dummy, vs, b, h, w, vn, hn, 0.5f, 0, max_num, 0, 0, nullptr, 0,
                       &dummy, nullptr, ws, ws_bytes);
}

int pvnet_vote_confidence(const float* kpts, float thresh, float* out_conf, uint32_t vote_flags, int b, int h, int w,
                          int vn, int hn, int max_num, void* workspace, size_t workspace_bytes, void* stream) {
    if (!kpts || !out_conf) return PVNET_E_BADARG;
    VoteParams P;
    int rc = params_for_workspace(P, b, h, w, vn, hn, max_num, workspace, workspace_bytes);
    if (rc) return rc;
    (void)vote_flags;  // records hold the raw direction in both scoring modes: nothing depends on the mode any more
    hipLaunchKernelGGL(confidence_kernel, dim3(vn, b), dim3(256), 0, static_cast<hipStream_t>(stream), P, kpts, thresh,
                       out_conf);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_vote_distribution(const float* mean, float* out_cov, int b, int h, int w, int vn, int hn, int max_num,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!mean || !out_cov) return PVNET_E_BADARG;
    VoteParams P;
    int rc = params_for_workspace(P, b, h, w, vn, hn, max_num, workspace, workspace_bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(distribution_kernel, dim3(vn, b), dim3(256), 0, static_cast<hipStream_t>(stream), P, mean,
                       out_cov);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_vote_band_margin(float thresh, uint32_t* out_stats, int b, int h, int w, int vn, int hn, int max_num,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!out_stats) return PVNET_E_BADARG;
    VoteParams P;
    int rc = params_for_workspace(P, b, h, w, vn, hn, max_num, workspace, workspace_bytes);
    if (rc) return rc;
    if (!(thresh >= 1e-3f && thresh < 1.f) || !P.mode) return PVNET_E_UNSUPPORTED;  // the matrix-pipe modes' range
    P.thresh = thresh;
    P.tau = (float)(sqrt(1.0 - (double)thresh * thresh) / (double)thresh);
    P.kband = band_constant(thresh);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PV_HIP(hipMemsetAsync(out_stats, 0, sizeof(uint32_t) * 4 * (size_t)b * vn, s));
    const size_t lds = 8 * TILE_U4 * sizeof(uint4) + 256 * sizeof(float4);
    hipLaunchKernelGGL(band_margin_kernel, dim3((unsigned)(b * vn), (unsigned)((P.cap + 255) / 256)), dim3(256), lds, s, P,
                       reinterpret_cast<unsigned*>(out_stats));
    PV_LAUNCH_CHECK();
    return 0;
}

// ---- ransac_motion_voting: its own small workspace (bit mask, segment counts, per-segment float64 sums) ----------
static int motion_layout(int b, int h, int w, int vn, size_t off[4], int* words, int* nseg) {
    if (b <= 0 || h <= 0 || w <= 0 || vn <= 0) return PVNET_E_BADARG;
    if ((long long)h * w > (1ll << 30) || b > 65535 || vn > 65535) return PVNET_E_UNSUPPORTED;
    *words = (int)(((long long)h * w + 63) / 64);
    *nseg = (*words + SEG_WORDS - 1) / SEG_WORDS;
    off[0] = 0;                                                                    // bits  u64 [b][words]
    off[1] = align_up(off[0] + sizeof(uint64_t) * (size_t)b * *words, 256);         // seg   i32 [2][b][nseg]
    off[2] = align_up(off[1] + sizeof(int32_t) * 2 * (size_t)b * *nseg, 256);       // part  f64 [b][nseg][vn][2]
    off[3] = align_up(off[2] + sizeof(double) * 2 * (size_t)b * *nseg * vn, 256);   // total
    return 0;
}

size_t pvnet_motion_workspace_bytes(int b, int h, int w, int vn) {
    size_t off[4];
    int words, nseg;
    return motion_layout(b, h, w, vn, off, &words, &nseg) == 0 ? off[3] : 0;
}

int pvnet_motion_voting(const void* mask, int mask_dtype, const int64_t mask_strides[3], const float* vertex,
                        const int64_t vertex_strides[5], int b, int h, int w, int vn, float* out_pts, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return pvnet_motion_voting_typed(mask, mask_dtype, mask_strides, vertex, vertex_strides, b, h, w, vn, 0u, out_pts,
                                     workspace, workspace_bytes, stream);
}

int pvnet_motion_voting_typed(const void* mask, int mask_dtype, const int64_t mask_strides[3], const void* vertex,
                              const int64_t vertex_strides[5], int b, int h, int w, int vn, uint32_t flags, float* out_pts,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!mask || !mask_strides || !vertex || !vertex_strides || !out_pts || !workspace) return PVNET_E_BADARG;
    if ((flags & PVNET_F_VERTEX_F16) && (flags & PVNET_F_VERTEX_BF16)) return PVNET_E_BADARG;
    if (flags & ~(uint32_t)(PVNET_F_VERTEX_F16 | PVNET_F_VERTEX_BF16)) return PVNET_E_BADARG;
    if (mask_dtype < PVNET_MASK_U8 || mask_dtype > PVNET_MASK_F32) return PVNET_E_BADARG;
    size_t off[4];
    int words, nseg;
    int rc = motion_layout(b, h, w, vn, off, &words, &nseg);
    if (rc) return rc;
    if (workspace_bytes < off[3]) return PVNET_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return PVNET_E_BADARG;
    char* base = static_cast<char*>(workspace);
    VoteParams P = {};
    P.mask = mask; P.ms0 = mask_strides[0]; P.ms1 = mask_strides[1]; P.ms2 = mask_strides[2];
    P.num_classes = 1;
    P.mask_dtype = mask_dtype;
    P.mask_linear = (mask_strides[2] == 1 && mask_strides[1] == w) ? 1 : 0;
    P.vertex = static_cast<const float*>(vertex);  // (typed by vertex_type)
    P.vertex_type = (flags & PVNET_F_VERTEX_F16) ? VT_F16 : (flags & PVNET_F_VERTEX_BF16) ? VT_BF16 : VT_F32;
    P.vs0 = vertex_strides[0]; P.vs1 = vertex_strides[1]; P.vs2 = vertex_strides[2]; P.vs3 = vertex_strides[3];
    P.vs4 = vertex_strides[4];
    P.b = b; P.h = h; P.w = w; P.vn = vn; P.npix = h * w; P.words = words; P.nseg = nseg;
    P.bits = reinterpret_cast<uint64_t*>(base + off[0]);
    P.seg = reinterpret_cast<int32_t*>(base + off[1]);
    P.seg0 = P.seg + (size_t)b * nseg;
    double* part = reinterpret_cast<double*>(base + off[2]);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = launch_mask_bits(P, s);
    if (rc) return rc;
    PV_LAUNCH_CHECK();
    hipLaunchKernelGGL(motion_partial_kernel, dim3(nseg, b), dim3(256), 0, s, P, part);
    PV_LAUNCH_CHECK();
    hipLaunchKernelGGL(motion_final_kernel, dim3(vn, b), dim3(64), 0, s, P, part, out_pts);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_generate_hypothesis(const float* direct, const float* coords, const int32_t* idxs, float* hypo_pts, int tn,
                              int vn, int hn, void* stream) {
    if (!direct || !coords || !idxs || !hypo_pts || tn <= 0 || vn <= 0 || hn <= 0) return PVNET_E_BADARG;
    hipLaunchKernelGGL(op_generate_hypothesis_kernel, dim3((hn * vn + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), direct, coords, idxs, hypo_pts, tn, vn, hn);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_voting_for_hypothesis(const float* direct, const float* coords, const float* hypo_pts, uint8_t* inliers,
                                int tn, int vn, int hn, float inlier_thresh, void* stream) {
    if (!direct || !coords || !hypo_pts || !inliers || tn <= 0 || vn <= 0 || hn <= 0) return PVNET_E_BADARG;
    if (vn > 65535) return PVNET_E_UNSUPPORTED;
    int hslice, slices;
    op_voting_grid(tn, vn, hn, &hslice, &slices);
    hipLaunchKernelGGL(op_voting_kernel, dim3((tn + 255) / 256, vn, slices), dim3(256), 0,
                       static_cast<hipStream_t>(stream), direct, coords, hypo_pts, inliers, tn, vn, hn, inlier_thresh,
                       hslice);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_generate_hypothesis_vanishing_point(const float* direct, const float* coords, const int32_t* idxs,
                                              float* hypo_pts, int tn, int vn, int hn, void* stream) {
    if (!direct || !coords || !idxs || !hypo_pts || tn <= 0 || vn <= 0 || hn <= 0) return PVNET_E_BADARG;
    hipLaunchKernelGGL(op_generate_hypothesis_vp_kernel, dim3((hn * vn + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), direct, coords, idxs, hypo_pts, tn, vn, hn);
    PV_LAUNCH_CHECK();
    return 0;
}

int pvnet_voting_for_hypothesis_vanishing_point(const float* direct, const float* coords, const float* hypo_pts,
                                                uint8_t* inliers, int tn, int vn, int hn, float inlier_thresh,
                                                void* stream) {
    if (!direct || !coords || !hypo_pts || !inliers || tn <= 0 || vn <= 0 || hn <= 0) return PVNET_E_BADARG;
    if (vn > 65535) return PVNET_E_UNSUPPORTED;
    int hslice, slices;
    op_voting_grid(tn, vn, hn, &hslice, &slices);
    hipLaunchKernelGGL(op_voting_vp_kernel, dim3((tn + 255) / 256, vn, slices), dim3(256), 0,
                       static_cast<hipStream_t>(stream), direct, coords, hypo_pts, inliers, tn, vn, hn, inlier_thresh,
                       hslice);
    PV_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
