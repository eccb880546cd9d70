// k2_compact.hip -- K2: thinning of the own segment (tn0 > max_num), order-preserving compaction, one float4 record per (pixel, key-point)
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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
                if (bi == 0) {   // the call's flags (vote_common.h): K3 sets / adds to them
                    int32_t* cf = call_flags_ptr(P);
                    const bool prev = P.ctrl[P.b * CTRL_STRIDE + 6] == P.layout_fp;   // a previous call of this layout left its votes here
                    cf[CF_BATCH_OK] = (!prev || 3 * cf[CF_VOTES_NOW] >= 2 * P.b) ? 1 : 0;
                    cf[CF_VOTES_NOW] = 0;
                    cf[CF_ANY_CULLED] = 0;
                }
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


}  // namespace

int launch_compact(const VoteParams& P, hipStream_t s, bool literal, int kg) {
    dim3 grid(P.nseg, P.b, (P.vn + kg - 1) / kg);
    const dim3 g3(P.nseg, P.b, (P.vn + 2) / 3);
    (void)grid;
#ifdef PVNET_DEV
#define PV_K2(VT)                                                                                                  \
    do {                                                                                                           \
        if (literal || P.exact) hipLaunchKernelGGL((compact_kernel<true, 3, VT>), g3, dim3(256), 0, s, P);         \
        else if (kg == 1) hipLaunchKernelGGL((compact_kernel<false, 1, VT>), grid, dim3(256), 0, s, P);            \
        else if (kg == 9) hipLaunchKernelGGL((compact_kernel<false, 9, VT>), grid, dim3(256), 0, s, P);            \
        else hipLaunchKernelGGL((compact_kernel<false, 3, VT>), g3, dim3(256), 0, s, P);                           \
    } while (0)
#else   // release: three key-points per block (the PVNET_COMPACT_KG variants 1 and 9 are development builds only)
#define PV_K2(VT)                                                                                                  \
    do {                                                                                                           \
        if (literal || P.exact) hipLaunchKernelGGL((compact_kernel<true, 3, VT>), g3, dim3(256), 0, s, P);         \
        else hipLaunchKernelGGL((compact_kernel<false, 3, VT>), g3, dim3(256), 0, s, P);                           \
    } while (0)
#endif
    if (P.vertex_type == VT_F16) PV_K2(VT_F16);
    else if (P.vertex_type == VT_BF16) PV_K2(VT_BF16);
    else PV_K2(VT_F32);
#undef PV_K2
    return 0;
}

}  // namespace pvd
