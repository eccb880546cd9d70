// epilogues.hip -- what reads a finished workspace or stands beside the layer: confidence / distribution (v5, estimate_voting_distribution_with_mean), ransac_motion_voting, the four ops of ransac_voting.cpp, the band-margin measurement, the clock-stamp collector; with their C entry points
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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


}  // namespace

int launch_ts_collect(const unsigned long long* stamps, int grid, unsigned long long* acc, int first, hipStream_t s) {
    hipLaunchKernelGGL(ts_collect_kernel, dim3(1), dim3(256), 0, s, stamps, grid, acc, first);
    return 0;
}

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


}  // namespace pvd

using namespace pvd;

extern "C" {

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

namespace pvd {

}  // namespace pvd
