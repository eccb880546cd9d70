// k5_refine.hip -- K5: arg-max over the counts (first index on ties), the winner's inliers, the 2x2 normal equations in float64 (ransac_voting_gpu.py:561-595)
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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


}  // namespace

int launch_select_refine(const VoteParams& P, hipStream_t s, bool literal) {
    dim3 grid(P.vn, P.b);
    if (literal) hipLaunchKernelGGL(select_refine_kernel<true>, grid, dim3(RT), 0, s, P);
    else hipLaunchKernelGGL(select_refine_kernel<false>, grid, dim3(RT), 0, s, P);
    return 0;
}

}  // namespace pvd
