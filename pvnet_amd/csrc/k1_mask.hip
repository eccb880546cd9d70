// k1_mask.hip -- K1: mask (any integer dtype / float32 / class logits, any strides) -> one bit per pixel, segment counts, thinning histograms
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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


}  // namespace

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

}  // namespace pvd
