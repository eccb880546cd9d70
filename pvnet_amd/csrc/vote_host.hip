// vote_host.hip -- host side of libpvnet_vote.so: the hand-written gfx950 (MI355X / CDNA4) implementation of PVNet's RANSAC voting layer.
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
//
// Translation units (pvnet_amd/build.py compiles them in parallel and links libpvnet_vote.so):
//   vote_common.h        parameter block, workspace conventions, the reference's arithmetic, bf16x3 operands, the rounding band
//   k1_mask.hip          K1    k2_compact.hip  K2    k3_hypotheses.hip  K3 (+ plan, band origin, culling selection, sort)
//   k4_score_valu.hip    K4 literal mode       k4_score_mfma.hip  K4 approximate mode
//   k4_exact_body.h / k4_score_exact.hip   K4 exact mode, dense        k4_score_cull.hip  K4 exact mode, dense + disc-culled items in one launch
//   k5_refine.hip        K5    epilogues.hip   confidence / distribution / motion voting / the four ops / band margin + their C entries
//   vote_host.hip        this file: rounding-band constant, tuning, layout, parameter block, the launch sequence, the C ABI of the layer
//   pvnet_nn.hip, pvnet_rccl.hip   nearest-neighbour search (ADD-S), the library's own RCCL all-gather
#include "vote_common.h"

namespace pvd {

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

// Tuning knobs.  RELEASE builds (libpvnet_vote.so): constants -- the shape-dependent defaults below; nothing reads the environment and
// pvnet_vote_tuning_reload() does nothing.  DEVELOPMENT builds (-DPVNET_DEV: libpvnet_vote_dev.so, what the knob tests, the fuzz
// matrix and the tuning tools load): the environment is read ONCE, at the first call into the library, never on the launch path;
// pvnet_vote_tuning_reload() (host-only) reads it again.
#ifdef PVNET_DEV
int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
#else
int env_int(const char*, int dflt) { return dflt; }
#endif

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

int launch_all(const VoteParams& P, hipStream_t s, hipEvent_t* ev, int stage_mask = -1, bool timed_score = false,
               int* score_grid = nullptr) {
    const bool literal = (P.flags & PVNET_F_LITERAL) != 0;
    auto mark = [&](int i) -> hipError_t { return ev ? hipEventRecord(ev[i], s) : hipSuccess; };
    // bit i set = launch stage i (K1, -, K2, K3, K4, K5; slot 1 is empty since round 2); a workspace left by a complete call stays valid, so single
    // stages can be re-run on it in isolation (pvnet_vote_v3_stage_repeat; development aid: PVNET_DEV_STAGES)
    const Tuning& T = tuning();
    const int stages = stage_mask >= 0 ? stage_mask : T.dev_stages;
    int rc = 0;
    PV_HIP(mark(0));
    if (stages & 1) rc = launch_mask_bits(P, s);   // K1
    if (rc) return rc;
    PV_LAUNCH_CHECK();
    PV_HIP(mark(1));
    // (stage slot 1 was the thinning launch of round 1: the mask kernel's histograms + the compaction kernel do it now)
    PV_HIP(mark(2));
    if (stages & 4) rc = launch_compact(P, s, literal, T.compact_kg);   // K2
    if (rc) return rc;
    PV_LAUNCH_CHECK();
    PV_HIP(mark(3));
    if (stages & 8) rc = launch_hypotheses(P, s, literal);   // K3
    if (rc) return rc;
    PV_LAUNCH_CHECK();
    PV_HIP(mark(4));
    if (stages & 16) {   // K4: persistent grid, work items strided over its waves
        const long long max_items =
            (long long)P.b * P.vn * (P.hgroups / P.wg_g) * ((P.max_chunks + P.wg_s - 1) / P.wg_s);
        // (12: the four-waves-per-SIMD scoring kernel of a batch alone, three rounds of four resident workgroups; measured with it only)
        const bool conc = (P.flags & PVNET_F_CONCURRENT) != 0;
        const bool alone8 = P.exact && P.wg_g * P.hpl / 2 == 8 && !conc && T.score_acc != 2 && T.score_runs != 1;
        const int wgs_per_cu = T.wgs_per_cu >= 0 ? T.wgs_per_cu : (alone8 ? 12 : 8);
        long long wgs = wgs_per_cu > 0 ? (long long)T.cus * wgs_per_cu : max_items;  // 0: one workgroup per item
        if (wgs > max_items) wgs = max_items;
        if (wgs < 1) wgs = 1;
        if (score_grid) *score_grid = (int)wgs;
        const dim3 g((unsigned)wgs);
        // exact mode, 8 tiles per wave -- calls flagged PVNET_F_CONCURRENT (other batches in flight): contiguous runs, one accumulator
        // pair, three waves per SIMD (136 VGPRs; four cost 6 % there: profiles/r04_ab_runs.txt); a batch alone: strided items, one pair
        // in 128 VGPRs = four waves per SIMD (kernel -2 %), 12 workgroups per CU.  Development builds: PVNET_SCORE_ACC=2 /
        // PVNET_SCORE_RUNS force the round-3 form (two pairs, 168 VGPRs) / either mapping; runs need cells of one pixel tile.
        const bool one_acc = T.score_acc == 1 || T.score_acc < 0;
        const bool runs = T.score_runs == 1 || (T.score_runs < 0 && conc);  // (cells of a whole item: the same 136-VGPR kernel, strided items)
        if (P.exact && P.cull) {
            // key-points may be disc-culled (K3 decides per image; PVNET_F_CULL_ALL: all of them): ONE launch scores both kinds of
            // item (score_exact_kernel_both_*), with the dense kernel's registers, LDS and grid
            rc = launch_score_both(P, g, s, timed_score, runs);
        } else if (P.exact) {
            const int mh = P.wg_g * P.hpl / 2;
            const int npx = P.wg_s * P.chunk;
            size_t lds = (size_t)(npx / 32) * TILE_U4 * sizeof(uint4) + (size_t)npx * sizeof(float4) +
                         (size_t)4 * mh * 32 * sizeof(float2) + (size_t)4 * mh * 64 * sizeof(unsigned);
            if (T.score_lds_kb > 0 && T.score_lds_kb <= 64 && lds < (size_t)T.score_lds_kb * 1024) lds = (size_t)T.score_lds_kb * 1024;
            rc = launch_score_exact(P, g, lds, s, timed_score, one_acc, runs);
        } else if (!literal && P.mode) {
            size_t lds = (size_t)(P.wg_s * P.chunk / 32) * TILE_U4 * sizeof(uint4);
            if (T.score_lds_kb > 0 && T.score_lds_kb <= 64 && lds < (size_t)T.score_lds_kb * 1024) lds = (size_t)T.score_lds_kb * 1024;
            rc = launch_score_mfma(P, g, lds, s, timed_score);
        } else {
            rc = launch_score_valu(P, g, s, literal);
        }
        if (rc) return rc;
        PV_LAUNCH_CHECK();
    }
    PV_HIP(mark(5));
    if (stages & 32) rc = launch_select_refine(P, s, literal);   // K5
    if (rc) return rc;
    PV_LAUNCH_CHECK();
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
    // (development builds: PVNET_SCORE_CULL = 0 / 1 / 2 also shapes the layout); PVNET_F_CULL_ALL / PVNET_F_CULL_NONE override the
    // selection per call where the layout has the buffers -- every choice gives the same counts
    const int cull_knob = tuning().score_cull >= 0 ? tuning().score_cull : PVNET_CULL_DEFAULT;
    if ((flags & PVNET_F_CULL_ALL) && (flags & PVNET_F_CULL_NONE)) return PVNET_E_BADARG;
    P.cull = (L.cull && P.exact && P.fold1 && vn <= KP_MAX) ? ((cull_knob == 1 || (flags & PVNET_F_CULL_ALL)) ? 1 : 2) : 0;
    if (flags & PVNET_F_CULL_NONE) P.cull = 0;
    P.cull_q = 1e-3f * (float)tuning().cull_q_milli;
    P.perm = reinterpret_cast<int32_t*>(base + L.off_perm);
    P.hyps = reinterpret_cast<float2*>(base + L.off_hyps);
    P.cnts = reinterpret_cast<int32_t*>(base + L.off_cnts);
    P.hypc = reinterpret_cast<uint4*>(base + L.off_hypc);
    P.hypg = reinterpret_cast<float*>(base + L.off_hypc + align_up(sizeof(uint4) * 2 * (size_t)b * vn * (L.hn_pad / 32), 256));
    return 0;
}

}  // namespace pvd

using namespace pvd;

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int pvnet_vote_abi_version(void) { return PVNET_VOTE_ABI_VERSION; }
#ifndef PVNET_KERNEL_COUNT
#define PVNET_KERNEL_COUNT 0   /* (pvnet_amd/build.py counts the kernel descriptors of the other translation units) */
#endif
#define PV_STR_(x) #x
#define PV_STR(x) PV_STR_(x)
#ifdef PVNET_DEV
const char* pvnet_vote_build_info(void) {
    return "pvnet_vote gfx950 hip development build (environment knobs, every kernel variant), " PV_STR(PVNET_KERNEL_COUNT) " kernels, " __DATE__ " " __TIME__;
}
#else
const char* pvnet_vote_build_info(void) {
    return "pvnet_vote gfx950 hip release build (knobs are constants), " PV_STR(PVNET_KERNEL_COUNT) " kernels, " __DATE__ " " __TIME__;
}
#endif
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
                rc = launch_ts_collect(reinterpret_cast<const unsigned long long*>(P.pix), grid, acc, i == 0 ? 1 : 0, s);
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

}  // extern "C"
