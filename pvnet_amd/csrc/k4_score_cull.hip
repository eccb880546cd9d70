// k4_score_cull.hip -- K4, exact mode with disc culling: the culling body and the ONE launch that scores dense and disc-culled items (score_exact_kernel_both_*)
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"
#include "k4_exact_body.h"

namespace pvd {
namespace {

// ------------------------------------------------------------------------------------------------------------
// K4 -- disc culling (round 5; exact mode, 8 hypothesis tiles per wave, 256-pixel work items): the exact kernel's two MFMAs and
// 40 vector operations are only spent on the (pixel, hypothesis tile) pairs whose outcome geometry does not already fix.
// Hypotheses arrive sorted along a Hilbert curve (cull_block of hypothesis_kernel, k3_hypotheses.hip): a tile of 32 is a disc (centre q, radius rho_T).
//   coarse pass  per item, ONE MFMA pair per pixel tile against the 32 tile CENTRES of the item's hypothesis slice (the eight
//                pixel tiles are shared out over the four waves): x' = s' |M_i| m_i(q).  |x'| >= 1 - g (1 - mu_i) means the
//                pixel's margin has one sign on the whole disc, outside the rounding band for every hypothesis in it: the pixel
//                votes for all 32 hypotheses (x' > 0: one count per tile, s_cv) or for none.  Every other pixel is UNCERTAIN
//                for that tile and its index in the item joins the tile's list (s_list, one byte per entry).
//   fine pass    a wave walks its eight hypothesis tiles; for each it scores ceil(uncertain / 32) GATHERED pixel groups -- lane
//                `col` of the MFMA's A operand reads the row the list names, so any 32 pixels of the item form a tile -- with
//                the exact kernel's epilogue (vote_subs / vote_slow_open / vote_slow_close: x = dt' - |cr'|, cells of 16 tests,
//                flagged cells re-evaluated literally).  Lanes past a list's end read a dead row (x = -4).
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


}  // namespace

int launch_score_both(const VoteParams& P, dim3 g, hipStream_t s, bool timed, bool runs) {
    const dim3 t(256);
    if (timed) {
        if (runs) hipLaunchKernelGGL(score_exact_kernel_both_1_1, g, t, CULL_LDS_BYTES, s, P);
        else hipLaunchKernelGGL(score_exact_kernel_both_1_0, g, t, CULL_LDS_BYTES, s, P);
    } else {
        if (runs) hipLaunchKernelGGL(score_exact_kernel_both_0_1, g, t, CULL_LDS_BYTES, s, P);
        else hipLaunchKernelGGL(score_exact_kernel_both_0_0, g, t, CULL_LDS_BYTES, s, P);
    }
    return 0;
}

}  // namespace pvd
