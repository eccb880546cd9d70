// k3_hypotheses.hip -- K3: hypotheses (kernel.cu:11-49), the plan of the scoring work items, the band origin and the disc-culling selection per image, the sort + tile discs of a culled key-point
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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


// culled: the disc-culling body of the scoring launch scores this image's key-points (their items carry the mark)
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


// What every K3 block works out for itself about its image before anything else (cheap -- 8 vn intersections -- and no block then
// waits for another): per key-point the ORIGIN of the exact mode's rounding band and whether the key-point is DISC-CULLED.
//   origin   eight candidate intersections from FIXED pixel pairs spread over the foreground list (records t and t + tn / 2), their
//            component-wise median, rounded to integers.  It only scales the band -- no result depends on it -- so a bad estimate
//            (fewer than three usable candidates: the image's median pixel instead) costs re-evaluations, never correctness.
//   culling  P.cull = 1: every key-point (PVNET_F_CULL_ALL; PVNET_SCORE_CULL=1: tests and probes); P.cull = 2 (the default where the
//            layout supports it): a key-point VOTES for culling when ALL BUT ONE of its candidates lie close together -- the seventh
//            smallest of the eight Chebyshev distances from the median point <= cull_q rho tan(theta0): the field's angular noise is
//            small against the threshold angle AND few directions are outliers (a candidate is off when either pixel of its pair
//            is: with 20 % outliers two or three of eight are, the hypothesis cloud is wide, and the gathered rest costs more than
//            the dense sweep although the MEDIAN distance -- the first form of this test -- is still zero:
//            profiles/r06x_cull_crossover_outliers.txt); then most pixels are certain for most hypothesis tiles (crossover
//            measured: profiles/r06y_cull_crossover.txt).  The IMAGE's key-points are culled together when the majority votes so
//            (image_culled()): eight candidates scatter, and one selection per image keeps its items of one kind.
//            Any choice gives the same counts; a wrong one only costs time.
//            (Round 6 also tried the candidates' medians and ranks through lane shuffles instead of LDS + barriers: ds_bpermute made
//            the preamble 5 k cycles longer, quad permutes + ds_swizzle 2 k -- r07a, r07b; the LDS form stays.)
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
        }
        if (dj == dj && rank == (n / 2 > n - 2 ? n / 2 : n - 2))   // the largest distance but one (n <= 4: the median one)
            S.vote[kk] = P.cull == 1 || (P.cull == 2 && dj <= P.cull_q * band_rho(tn) * P.tau) ? 1 : 0;
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
    // (requested with the image's counts, long before it is needed: behind the preamble it was a dependent load of its own, 4 us of a
    //  culled call's hypothesis launch -- r07k)
    const bool batch_ok = P.cull == 1 || (P.cull == 2 && call_flags_ptr(P)[CF_BATCH_OK] != 0);
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
            *kp_cull_ptr(P, bk) = (P.cull && live && batch_ok && image_culled(S, P.vn)) ? 1 : 0;
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
    // (fill_params: P.cull implies the exact mode and vn <= KP_MAX) this image's key-points go to the scoring launch's disc-culling body
    // -- when the image's key-points vote for it AND the previous batch's majority did (CF_BATCH_OK, vote_common.h)
    const bool votes = !LITERAL && P.cull && kp_origin && live && image_culled(S, P.vn);
    const bool culling = votes && batch_ok;
    if (blk == nbd) {      // one extra block per image plans its scoring work items (consumed by the next launches only)
        if (votes && P.cull == 2 && threadIdx.x == 0) atomicAdd(call_flags_ptr(P) + CF_VOTES_NOW, 1);
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


}  // namespace

int launch_hypotheses(const VoteParams& P, hipStream_t s, bool literal) {
    // per image: the hypothesis blocks, the plan block and -- where key-points may be disc-culled -- one block per key-point
    dim3 grid((unsigned)(((P.hn * P.vn + 255) / 256 + 1 + (!literal && P.cull ? P.vn : 0)) * ((P.b + 7) / 8) * 8));
    if (literal) hipLaunchKernelGGL(hypothesis_kernel<true>, grid, dim3(256), 0, s, P);
    else hipLaunchKernelGGL(hypothesis_kernel<false>, grid, dim3(256), 0, s, P);
    return 0;
}

}  // namespace pvd
