// k4_score_valu.hip -- K4, VALU form: literal mode (the reference's float32 operation order for every pair; kernel.cu:88-126)
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"

namespace pvd {
namespace {

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


}  // namespace

int launch_score_valu(const VoteParams& P, dim3 grid, hipStream_t s, bool literal) {
    const size_t lds = (size_t)P.wg_s * P.chunk * (sizeof(float4) + sizeof(float2));
#ifdef PVNET_DEV
#define PV_VALU(H_)                                                                              \
    do {                                                                                         \
        if (literal) hipLaunchKernelGGL((score_kernel<H_, true>), grid, dim3(256), lds, s, P);   \
        else hipLaunchKernelGGL((score_kernel<H_, false>), grid, dim3(256), lds, s, P);          \
    } while (0)
#else   // release: literal mode only (the VALU form of the approximate predicate is PVNET_SCORE_MODE=0: development builds)
#define PV_VALU(H_)                                                                              \
    do {                                                                                         \
        if (!literal) return PVNET_E_UNSUPPORTED;                                                \
        hipLaunchKernelGGL((score_kernel<H_, true>), grid, dim3(256), lds, s, P);                \
    } while (0)
#endif
    switch (P.hpl) {
        case 1: PV_VALU(1); break;
        case 2: PV_VALU(2); break;
        case 4: PV_VALU(4); break;
        case 8: PV_VALU(8); break;
        default: return PVNET_E_UNSUPPORTED;
    }
#undef PV_VALU
    return 0;
}

}  // namespace pvd
