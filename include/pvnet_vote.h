/* pvnet_vote.h -- C ABI of the MI355X-native RANSAC voting layer (libpvnet_vote.so).
 *
 * Drop-in boundary for the reference's `lib/ransac_voting_gpu_layer`:
 *   - pvnet_vote_v3()                 replaces the whole of ransac_voting_layer_v3
 *                                     (lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:514-598), i.e. the Python
 *                                     loop plus the ~40 torch launches and the two extension ops per image;
 *   - pvnet_generate_hypothesis()     replaces the pybind op `generate_hypothesis`
 *                                     (src/ransac_voting.cpp:20-31 -> src/ransac_voting_kernel.cu:51-86);
 *   - pvnet_voting_for_hypothesis()   replaces the pybind op `voting_for_hypothesis`
 *                                     (src/ransac_voting.cpp:41-55 -> src/ransac_voting_kernel.cu:129-167);
 *   - pvnet_generate_hypothesis_vanishing_point() / pvnet_voting_for_hypothesis_vanishing_point()
 *                                     replace the other two pybind ops of the extension (src/ransac_voting.cpp:57-99).
 * Plain pointers and sizes only (no torch types).  All pointers are DEVICE pointers unless marked host.
 * Every entry point only enqueues work on `stream` (a hipStream_t passed as void*): no allocation, no
 * host synchronisation, no global state -- re-entrant and hipGraph-capturable.  (pvnet_vote_v3_profiled is the
 * one exception: it records events and synchronises, and exists for bench.py / profiling only.)
 *
 * Return value: 0 on success, a positive hipError_t from the runtime, or a negative PVNET_E_* code.
 * Unlike the reference (src/cuda_common.h:19-26 prints and exit()s) nothing here ever terminates the process.
 */
#ifndef PVNET_VOTE_H_
#define PVNET_VOTE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVNET_VOTE_ABI_VERSION 9

/* negative library error codes */
#define PVNET_E_BADARG      (-1)   /* null pointer / non-positive size / unsupported dtype or stride */
#define PVNET_E_WORKSPACE   (-2)   /* workspace_bytes smaller than pvnet_vote_workspace_bytes() */
#define PVNET_E_UNSUPPORTED (-3)   /* e.g. vn or hn beyond the compiled limits */

/* mask element types (foreground <=> (uint8)value != 0, i.e. torch's `.byte()` of ransac_voting_gpu.py:527) */
#define PVNET_MASK_U8   0          /* uint8 / int8 / bool */
#define PVNET_MASK_I16  1
#define PVNET_MASK_I32  2
#define PVNET_MASK_I64  3          /* what torch.argmax delivers (tools/demo.py:52) */
#define PVNET_MASK_F32  4
#define PVNET_MASK_LOGITS_F32 5     /* internal: pvnet_vote_v3_logits (fused arg-max over class planes) */

/* flags */
#define PVNET_F_LITERAL   1u       /* score with the reference's float32 operation order (sqrt + divide, one
                                      rounding per op) on the VALU for EVERY (pixel, hypothesis) pair: bit-exact with
                                      oracle32 and with the reference's own kernels, ~10x slower.
                                      DEFAULT (neither this flag nor PVNET_F_APPROX; "exact mode", ABI 6+): the same
                                      integers -- inlier counts and winners EQUAL the reference kernels' (the reading of
                                      ransac_voting_kernel.cu with one rounding per operation, -ffp-contract=off; nvcc's
                                      --fmad=true moves ~1e-6 of the reference's own flags) -- at matrix-pipe speed: every
                                      pair is tested as |d x u| < tan(acos(thresh)) * (d . u) by two bf16x3 MFMAs in units of
                                      the provable float32 rounding band of ransac_voting_kernel.cu:107-125 plus the matrix
                                      pipe's own error, and only the cells that hold a pair INSIDE the band (~6e-5 of the
                                      pairs at thresh 0.99) are re-evaluated in the reference's operation order
                                      (pvnet_vote.hip: score_exact_body; the band's measured safety
                                      margin: pvnet_vote_band_margin below).  Without the matrix-pipe buffers
                                      (PVNET_SCORE_MODE=0) the default is scored literally (ABI 7) */
#define PVNET_F_NO_REFINE 2u       /* skip ransac_voting_gpu.py:579-595, return the winning hypotheses */
/* element type of `vertex` (and, for pvnet_vote_v3_logits, of `seg_pred`) when it is not float32 -- what a backbone under
 * autocast emits.  The pointer is passed through the `const float*` parameter and read as the flagged type; strides stay
 * in ELEMENTS.  Elements are widened to float32 where they are read (exact), so the result equals the float32 path on
 * `tensor.float()` without the copy.  At most one flag per tensor. */
#define PVNET_F_VERTEX_F16   4u
#define PVNET_F_VERTEX_BF16  8u
#define PVNET_F_LOGITS_F16  16u
#define PVNET_F_LOGITS_BF16 32u
#define PVNET_F_APPROX      64u     /* the round-1/2 "fast" mode: matrix-pipe scoring WITHOUT the rounding-band rescoring.
                                      Counts are those of exact arithmetic to within the fp32-equivalent error of the
                                      bf16x3 products, i.e. they may differ from the reference's float32 kernels by a few
                                      votes per hypothesis where the reference's own rounding decides a pair.  ~10 % faster
                                      than the default; ignored with PVNET_F_LITERAL */

#define PVNET_F_BAND_STATS 128u     /* development aid (exact mode): count the re-evaluated cells / literal tests into the two
                                      spare words ctrl[b][4], ctrl[b][5] of the workspace (tools/exact_probe.py) */
#define PVNET_F_CONCURRENT 256u    /* hint (results do not depend on it): the caller keeps OTHER batches in flight on other streams.
                                      The exact-mode scoring kernel then runs three waves per SIMD in 136 VGPRs (a batch alone:
                                      four in 128, which would fill the SIMDs), leaving room for the small stages of the other
                                      batches, and (ABI 7) every workgroup takes a contiguous run of work items, keeping its B
                                      columns, hypotheses and vote counters while the (image, key-point) stays the same: +4 %
                                      throughput with six batches in flight, -5 % for a batch alone (profiles/r04_ab_runs.txt).
                                      The Python front end sets it when consecutive calls alternate streams
                                      (voting.concurrent_hint; concurrent=True / False decide explicitly). */

#define PVNET_F_CULL_ALL   512u    /* exact mode: score EVERY key-point with disc culling where the layout supports it (batch shapes with
                                      256-pixel work items, 1 024 hypotheses, at most 32 key-points); */
#define PVNET_F_CULL_NONE 1024u    /* ... or none.  Neither flag (the default): K3 selects per image on the device from the spread of
                                      eight candidate intersections per key-point (profiles/r06k_cull_crossover.txt).  The inlier counts
                                      are the same integers under every selection; only the time differs (tests / probes use the flags) */

/* per-(image,key-point) status bits written to out_status */
#define PVNET_S_SKIPPED   1        /* fewer than min_num foreground pixels (or none kept): zeros returned */
#define PVNET_S_SINGULAR  2        /* normal matrix singular: winning hypothesis returned (reference raises) */
#define PVNET_S_NO_INLIER 4        /* best hypothesis had zero inliers: (0,0) carried into refinement */
#define PVNET_S_OVERFLOW  8        /* kept pixels exceeded the workspace capacity and were truncated */

/* Layout of the caller-owned workspace (byte offsets).  Exposed so tests and tools can read the
 * intermediate products (compacted pixels, hypotheses, inlier counts) without extra copies. */
typedef struct PvnetVoteLayout {
    int32_t b, h, w, vn, hn;
    int32_t cap;            /* per-image capacity of the compacted pixel list (multiple of 8, incl. pad)   */
    int32_t words;          /* 64-pixel words per image in the foreground bit mask                          */
    int32_t chunk;          /* pixels per scoring work item                                                 */
    int32_t max_chunks;     /* ceil(cap / chunk)                                                            */
    int32_t hpl;            /* hypotheses per lane of the VALU scoring kernel (literal mode); fast mode: a
                               work item holds wg_g*64*hpl hypotheses = 4 waves x (wg_g*hpl/2) MFMA tiles of 32 */
    int32_t hgroups;        /* hypothesis groups per key-point = ceil(hn / (64*hpl)) rounded up to wg_g     */
    int32_t hn_pad;         /* hgroups * 64 * hpl                                                           */
    size_t off_ctrl;        /* int32 [b][8]: tn0, tn, status, item_base, nchunks, origin x, origin y, -  ; then [8] global: total items,
                               (1,7) culling statistics, (2,3) stage-timer ticks, (4,5) band statistics (PVNET_F_BAND_STATS), (6) layout
                               fingerprint; then int32 [b][vn][2] band origins, then int32 [b][vn] 1 = key-point disc-culled */
    size_t off_bits;        /* uint64 [b][words]           foreground bit mask (as the mask has it: before thinning) */
    size_t off_pix;         /* int32  [b][cap]             linear pixel index y*w+x of compacted pixel t    */
    size_t off_rec;         /* float4 [b][vn][cap]         record (x, y, ux, uy): pixel and its RAW direction for the
                                                           key-point (fast mode: zero when |u| < 1e-6, which never votes,
                                                           kernel.cu:121) -- the only per-pixel data, same in both modes */
    size_t off_hyp;         /* float2 [b][vn][hn_pad]      hypotheses                                       */
    size_t off_partial;     /* uint16 [b][vn][max_chunks][hn_pad]  per-chunk inlier counts -- EMPTY by default: the
                               scoring kernel adds its counts into `counts` with integer atomics (PVNET_SCORE_ATOMIC=1) */
    size_t off_counts;      /* int32  [b][vn][hn_pad]      inlier count of every hypothesis                 */
    size_t off_win;         /* int32  [b][vn][2]           (winner index, winner count)                     */
    size_t off_seg;         /* int32  [2][b][nseg]         ([1] = foreground count of every 4096-pixel segment), then,
                             *                              if max_num < h*w, uint16 [b][nseg][1024] cumulative histograms   */
    size_t off_items;       /* int32x4 [max items]         scoring work items (image, kp, chunk group, slice) */
    size_t off_hypb;        /* uint4  [b][vn][hn_pad][2]   fast mode: hypotheses as bf16x3 MFMA B operands         */
    size_t total_bytes;
    int32_t nseg;           /* ceil(words / 64)                                                             */
    int32_t wg_g, wg_s;     /* a scoring workgroup covers wg_g hypothesis groups x wg_s chunks (wg_g*wg_s=4) */
    int32_t reserved_;      /* 1: fast mode scores on the matrix pipe (score_mfma_kernel), 0: VALU kernel                */
    /* ABI 8 / 9 -- disc culling of the exact mode (k3_hypotheses.hip: cull_block of hypothesis_kernel; k4_score_cull.hip: score_exact_kernel_both).
     * Empty (cull = 0) unless the layout scores 8 hypothesis tiles per wave on 256-pixel work items with hn_pad = 1024 and
     * vn <= 32.  Which images of a call ARE culled is decided on the device -- an image's key-points vote from the spread of the
     * candidate intersections of the band-origin estimate, and a batch may be culled when two thirds of the PREVIOUS batch on
     * the same workspace voted for it (a workspace without history: yes); PVNET_F_CULL_ALL / PVNET_F_CULL_NONE override -- and
     * recorded as int32 [b][vn] behind the band origins in the ctrl block.  Culled or not, every count is the same integer. */
    int32_t cull;           /* 1: exact-mode calls of this layout may sort a key-point's hypotheses along a Hilbert curve and
                               score only the (pixel, hypothesis tile) pairs whose outcome the tile's disc does not fix        */
    size_t off_perm;        /* int32  [b][vn][hn_pad]      sorted position -> caller's hypothesis index (>= hn: padding)        */
    size_t off_hyps;        /* float2 [b][vn][hn_pad]      hypotheses in sorted order                                          */
    size_t off_cnts;        /* int32  [b][vn][hn_pad]      a culled key-point's counts in sorted order (`counts`: CALLER order, all)    */
    size_t off_hypc;        /* uint4  [b][vn][hn_pad/32][2] B column of every tile's centre, then float [b][vn][hn_pad/32] g    */
} PvnetVoteLayout;

/* Host-only: fills *out for a problem size.  max_num as passed to pvnet_vote_v3. */
int pvnet_vote_layout(int b, int h, int w, int vn, int hn, int max_num, PvnetVoteLayout* out);

/* Host-only: bytes of workspace pvnet_vote_v3 needs (0 on invalid arguments). 256-byte aligned base required. */
size_t pvnet_vote_workspace_bytes(int b, int h, int w, int vn, int hn, int max_num);

/* The whole layer.  Replaces ransac_voting_layer_v3 (ransac_voting_gpu.py:514-598).
 *   mask            [b,h,w] of mask_dtype, element strides mask_strides[3]
 *   vertex          [b,h,w,vn,2] float32, element strides vertex_strides[5] (any; planar in practice)
 *   hn              round_hyp_num
 *   inlier_thresh, min_num, max_num   as the reference's keyword arguments
 *   seed            counter-RNG seed (pixel pairs when idxs == NULL; Bernoulli subsample when tn0 > max_num:
 *                   a pixel is kept with probability ceil(1024 max_num / tn0) / 1024 -- the reference's max_num / tn0
 *                   rounded up to a multiple of 1/1024)
 *   image_base      global index of image 0 of this call: the RNG stream of image i is image_base + i, so a batch
 *                   sharded over GPUs draws exactly what the unsharded batch would (0 for a whole batch)
 *   idxs            NULL, or int32 [b,hn,vn,2] pixel-pair indices into each image's compacted list
 *   out_kpts        [b,vn,2] float32
 *   out_status      NULL or int32 [b,vn] PVNET_S_* bits
 * The reference's `confidence` / `max_iter` do not exist here: its loop re-uses one idxs draw (:547 vs :552) so
 * rounds after the first can never change the result (SURVEY.md finding 3). */
int pvnet_vote_v3(const void* mask, int mask_dtype, const int64_t mask_strides[3],
                  const float* vertex, const int64_t vertex_strides[5],
                  int b, int h, int w, int vn, int hn,
                  float inlier_thresh, int min_num, int max_num,
                  uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                  float* out_kpts, int32_t* out_status,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The same layer fed with the backbone's class LOGITS instead of a mask: seg_pred [b,C,h,w] float32 with element
 * strides seg_strides[4]; foreground <=> argmax_c(seg_pred) != 0 (first maximum wins ties), i.e. the
 * `mask = torch.argmax(seg_pred, 1)` of EvalWrapper.forward (tools/demo.py:52, tools/train_linemod.py:99-101) fused
 * into the first kernel -- the int64 mask is never materialised. */
int pvnet_vote_v3_logits(const float* seg_pred, const int64_t seg_strides[4], int num_classes,
                         const float* vertex, const int64_t vertex_strides[5],
                         int b, int h, int w, int vn, int hn,
                         float inlier_thresh, int min_num, int max_num,
                         uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                         float* out_kpts, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Same call, timed stage by stage with hipEvents on `stream`; synchronises the stream before returning.
 * stage_ms (host, PVNET_NUM_STAGES floats) receives the GPU time of each stage of this call. bench/profiling only. */
#define PVNET_STAGE_MASK      0   /* mask -> bit mask + segment counts + thinning histograms (HBM read of the mask) */
#define PVNET_STAGE_SUBSAMPLE 1   /* EMPTY since ABI 5 (was the Bernoulli-subsample launch): always ~0               */
#define PVNET_STAGE_COMPACT   2   /* thinning when tn0 > max_num + order-preserving compaction + vector gather       */
#define PVNET_STAGE_HYP       3   /* hypothesis generation + per-image work-item plan                        */
#define PVNET_STAGE_SCORE     4   /* inlier scoring (dominant; matrix pipe + VALU compare)                   */
#define PVNET_STAGE_REFINE    5   /* arg-max + least-squares refinement                                      */
#define PVNET_NUM_STAGES      6
int pvnet_vote_v3_profiled(const void* mask, int mask_dtype, const int64_t mask_strides[3],
                           const float* vertex, const int64_t vertex_strides[5],
                           int b, int h, int w, int vn, int hn,
                           float inlier_thresh, int min_num, int max_num,
                           uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                           float* out_kpts, int32_t* out_status,
                           void* workspace, size_t workspace_bytes, void* stream, float* stage_ms);

/* Profiling only, like pvnet_vote_v3_profiled: runs the whole path once, then re-launches ONE stage (PVNET_STAGE_*)
 * `repeats` times back to back on that workspace, bracketed by a single hipEvent pair on `stream`; synchronises.
 * avg_ms (host, TWO floats): [1] = event time / repeats (kernel + the ~1.5 us dependent-launch boundary, at the clocks a
 * back-to-back run sustains).  [0] = for PVNET_STAGE_SCORE in fast mode the kernel's own duration: a second series of
 * `repeats` launches of the same kernel in which every workgroup stamps the constant-rate device clock at its first and
 * last instruction (max end - min start per launch, averaged) -- what a kernel trace reports for it, measured live;
 * for every other stage [0] = [1].  bench.py's roofline uses [0]. */
int pvnet_vote_v3_stage_repeat(const void* mask, int mask_dtype, const int64_t mask_strides[3],
                               const float* vertex, const int64_t vertex_strides[5],
                               int b, int h, int w, int vn, int hn,
                               float inlier_thresh, int min_num, int max_num,
                               uint64_t seed, int image_base, const int32_t* idxs, uint32_t flags,
                               float* out_kpts, int32_t* out_status,
                               void* workspace, size_t workspace_bytes, void* stream,
                               int stage, int repeats, float* avg_ms);

/*
 * pvnet_vote_band_margin (development aid, tools/band_margin.py): how safe is the exact mode's rounding band?  To be called on
 * the workspace of a completed DEFAULT-mode (exact) call with that call's inlier_thresh.  Every (pixel, hypothesis) test is
 * re-evaluated on the matrix pipe exactly as the scoring kernel does (x = dt' - |cr'|, trusted there where |x| >= 1) and with
 * the reference's arithmetic (ransac_voting_kernel.cu:107-125); out_stats [b*vn][4] uint32 on the device:
 *   [0] float bits of max |x| over the tests whose matrix-pipe vote (x > 0) differs from the reference's -- must be < 1,
 *   [1] the number of such tests, [2] tests with |x| < 1 (the band as scored), [3] all tests (mod 2^32).
 */
int pvnet_vote_band_margin(float thresh, uint32_t* out_stats, int b, int h, int w, int vn, int hn, int max_num,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * The path's one exchange, issued by the library (ABI 9; pvnet_amd/csrc/pvnet_rccl.hip).  Replaces the gather of
 * torch.nn.DataParallel(EvalWrapper) (tools/train_linemod.py:183-184, tools/demo.py:174): every rank votes its own images and ONE
 * ncclAllGather of count_per_rank float32 per rank -- on the stream the votes were issued on, so stream order is the only
 * dependency -- gives every rank all key-points.  librccl is dlopen()ed on first use (`path` NULL: librccl.so, librccl.so.1,
 * /opt/rocm/lib); nothing here links against it.  Return values: 0, PVNET_E_*, or RCCL's own ncclResult_t (> 0).
 *   pvnet_rccl_load         host-only; explicit library path (e.g. the librccl.so PyTorch-ROCm ships), idempotent
 *   pvnet_rccl_unique_id    128 bytes a single rank generates and the caller's bootstrap (torch.distributed, MPI, a file) hands to all
 *   pvnet_rccl_comm_init    collective over the ranks, on the CURRENT device; *comm is the ncclComm_t
 *   pvnet_vote_allgather    all [nranks * count_per_rank] <- every rank's local [count_per_rank]; no sync, no allocation
 */
int pvnet_rccl_load(const char* path);
int pvnet_rccl_unique_id(void* id128);
int pvnet_rccl_comm_init(void** comm, int nranks, const void* id128, int rank);
int pvnet_rccl_comm_ranks(void* comm, int* nranks);
int pvnet_rccl_comm_destroy(void* comm);
int pvnet_vote_allgather(const float* local, float* all, size_t count_per_rank, void* comm, void* stream);

/* Epilogues of the reference's sibling functions.  Both run on the WORKSPACE of a preceding pvnet_vote_v3 call with
 * the same (b,h,w,vn,hn,max_num) on the same stream (they read its compacted pixel lists, hypotheses and counts).
 *
 * pvnet_vote_confidence: ransac_voting_layer_v5's second output (ransac_voting_gpu.py:846-850): out_conf[b,vn] =
 *   fraction of the image's kept pixels whose direction points at kpts[b,vn,2] within `thresh` (0.999 there).
 *   vote_flags: ignored since ABI 3 (records hold the raw direction in both scoring modes); kept for call compatibility. */
int pvnet_vote_confidence(const float* kpts, float thresh, float* out_conf, uint32_t vote_flags, int b, int h, int w,
                          int vn, int hn, int max_num, void* workspace, size_t workspace_bytes, void* stream);
/* pvnet_vote_distribution: estimate_voting_distribution_with_mean's epilogue (ransac_voting_gpu.py:389-404):
 *   out_cov[b,vn,2,2] = sum_h w_h (hyp_h - mean)(hyp_h - mean)^T / (sum_h w_h + 1e-3), w_h = inlier ratio of
 *   hypothesis h where it is within 0.1 of the key-point's best ratio, else 0; mean [b,vn,2]. */
int pvnet_vote_distribution(const float* mean, float* out_cov, int b, int h, int w, int vn, int hn, int max_num,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ransac_motion_voting (ransac_voting_gpu.py:960-981): out_pts[b,vn,2] = mean over the image's foreground pixels of
 * (vertex + (x, y)); zeros for an image without foreground.  mask / vertex as for pvnet_vote_v3 (any strides, read in
 * place, only foreground vectors are touched).  Needs its own small workspace (bit mask + per-segment float64 sums):
 * pvnet_motion_workspace_bytes(b, h, w, vn) bytes, 256-byte aligned. */
size_t pvnet_motion_workspace_bytes(int b, int h, int w, int vn);
int pvnet_motion_voting(const void* mask, int mask_dtype, const int64_t mask_strides[3], const float* vertex,
                        const int64_t vertex_strides[5], int b, int h, int w, int vn, float* out_pts, void* workspace,
                        size_t workspace_bytes, void* stream);
/* the same with a float16 / bfloat16 field read in place: flags = PVNET_F_VERTEX_F16 or PVNET_F_VERTEX_BF16 (0: float32);
 * every element is widened where it is read, so the result equals pvnet_motion_voting on the widened field bit for bit */
int pvnet_motion_voting_typed(const void* mask, int mask_dtype, const int64_t mask_strides[3], const void* vertex,
                              const int64_t vertex_strides[5], int b, int h, int w, int vn, uint32_t flags,
                              float* out_pts, void* workspace, size_t workspace_bytes, void* stream);

/* Op-level entry points with the reference extension's tensor layouts.
 * direct [tn,vn,2] f32, coords [tn,2] f32, idxs [hn,vn,2] i32 -> hypo_pts [hn,vn,2] f32 (fully written; degenerate
 * pairs give (0,0) as the reference's at::zeros + early return do, ransac_voting_kernel.cu:42-43,75). */
int pvnet_generate_hypothesis(const float* direct, const float* coords, const int32_t* idxs, float* hypo_pts,
                              int tn, int vn, int hn, void* stream);

/* inliers [hn,vn,tn] uint8, in/out: a 1 is stored where the pixel votes for the hypothesis, other bytes are left
 * untouched (ransac_voting_kernel.cu:124-125).  Float32 operation order of the reference (literal mode). */
int pvnet_voting_for_hypothesis(const float* direct, const float* coords, const float* hypo_pts, uint8_t* inliers,
                                int tn, int vn, int hn, float inlier_thresh, void* stream);

/* The vanishing-point pair of the reference's extension (src/ransac_voting.cpp:57-99 ->
 * src/ransac_voting_kernel.cu:170-266, :268-351): hypo_pts [hn,vn,3] are homogeneous points (x, y, z), the cross
 * product of the two pixels' line coordinates (z = 0: the rays are parallel; (0,0,0): they do not meet); a pixel votes
 * when |cos angle(direct, h.xy - c * h.z)| > inlier_thresh and both component products are non-negative.  Same layouts,
 * in/out convention and float32 operation order as the two ops above. */
int pvnet_generate_hypothesis_vanishing_point(const float* direct, const float* coords, const int32_t* idxs,
                                              float* hypo_pts, int tn, int vn, int hn, void* stream);
int pvnet_voting_for_hypothesis_vanishing_point(const float* direct, const float* coords, const float* hypo_pts,
                                                uint8_t* inliers, int tn, int vn, int hn, float inlier_thresh,
                                                void* stream);

/* ABI / build identification (host-only) */
int pvnet_vote_abi_version(void);
const char* pvnet_vote_build_info(void);

/* Host-only.  RELEASE build (libpvnet_vote.so): the tuning knobs are compile-time constants, the library reads no environment and
 * this call does nothing.  DEVELOPMENT build (libpvnet_vote_dev.so, -DPVNET_DEV, same ABI): the PVNET_* tuning environment
 * variables are read ONCE, at the first call into the library -- never on the launch path; this re-reads them (tests and the
 * tuning tools change them in-process).  Knobs re-shape grids and work items, never results. */
void pvnet_vote_tuning_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* PVNET_VOTE_H_ */
