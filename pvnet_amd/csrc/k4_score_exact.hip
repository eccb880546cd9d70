// k4_score_exact.hip -- K4, exact mode (the default): the dense scoring kernels -- matrix-pipe scoring with the rounding-band epilogue, counts equal to kernel.cu:88-126
// (part of libpvnet_vote.so; the stage map is at the top of vote_host.hip, the shared definitions in vote_common.h)
#include "vote_common.h"
#include "k4_exact_body.h"

namespace pvd {
namespace {

// The register allocator fills whatever budget the occupancy target leaves (3 waves per SIMD: up to 168 VGPRs), the library needs
// the top granule of every allocation unused (PVNET_SPARE_VGPRS): amdgpu_num_vgpr -- a literal, hence one definition per
// instantiation -- caps what the code may use one granule below what PVNET_SPARE_VGPRS makes the kernel allocate (the
// backend doubles the attribute's value on targets with a unified VGPR / AGPR file, hence the / 2).
// Release builds hold the variants the library's defaults reach: 1 / 2 / 4 tiles per wave (small hypothesis counts; cells of one pixel
// tile, two accumulator pairs) and, at 8 tiles per wave, one accumulator pair with strided items (a batch alone) or contiguous runs
// (batches in flight), each also with the clock stamps of the profiling entry.  Cells of a whole work item (PVNET_EXACT_FOLD=0), two
// accumulator pairs at 8 tiles (PVNET_SCORE_ACC=2) and the stamped forms of the small shapes are development builds (-DPVNET_DEV).
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
#ifdef PVNET_DEV
PV_DEF_SCORE_EXACT4(1, 2, 0, 104) PV_DEF_SCORE_EXACT4(2, 2, 0, 104) PV_DEF_SCORE_EXACT4(4, 2, 0, 136)
PV_DEF_SCORE_EXACT4(8, 1, 0, 120) PV_DEF_SCORE_EXACT4(8, 2, 0, 160)
PV_DEF_SCORE_EXACT4(8, 1, 1, 128) PV_DEF_SCORE_EXACT4(8, 2, 1, 160)
#else
PV_DEF_SCORE_EXACT(1, 1, 0, 2, 0, 104) PV_DEF_SCORE_EXACT(2, 1, 0, 2, 0, 104) PV_DEF_SCORE_EXACT(4, 1, 0, 2, 0, 136)
PV_DEF_SCORE_EXACT(8, 1, 0, 1, 0, 120) PV_DEF_SCORE_EXACT(8, 1, 1, 1, 0, 120)
PV_DEF_SCORE_EXACT(8, 1, 0, 1, 1, 128) PV_DEF_SCORE_EXACT(8, 1, 1, 1, 1, 128)
#endif
#undef PV_DEF_SCORE_EXACT4
#undef PV_DEF_SCORE_EXACT

}  // namespace

// one_acc / runs: one accumulator pair / contiguous item runs (8 tiles per wave only); cells: P.fold1
int launch_score_exact(const VoteParams& P, dim3 g, size_t lds, hipStream_t s, bool timed, bool one_acc, bool runs) {
    const int mh = P.wg_g * P.hpl / 2;
    const dim3 t(256);
#ifdef PVNET_DEV
    const int fold = P.fold1;
#define PV_EXACT3(MH_, NACC_, RUNS_)                                                                                \
    do {                                                                                                            \
        if (timed) {                                                                                                \
            if (fold == 1) hipLaunchKernelGGL((ScoreExact<MH_, 1, true, NACC_, RUNS_>::kernel), g, t, lds, s, P);   \
            else hipLaunchKernelGGL((ScoreExact<MH_, 0, true, NACC_, RUNS_>::kernel), g, t, lds, s, P);             \
        } else {                                                                                                    \
            if (fold == 1) hipLaunchKernelGGL((ScoreExact<MH_, 1, false, NACC_, RUNS_>::kernel), g, t, lds, s, P);  \
            else hipLaunchKernelGGL((ScoreExact<MH_, 0, false, NACC_, RUNS_>::kernel), g, t, lds, s, P);            \
        }                                                                                                           \
    } while (0)
    switch (mh) {
        case 1: PV_EXACT3(1, 2, false); break;
        case 2: PV_EXACT3(2, 2, false); break;
        case 4: PV_EXACT3(4, 2, false); break;
        case 8:
            if (runs) { if (one_acc) PV_EXACT3(8, 1, true); else PV_EXACT3(8, 2, true); }
            else { if (one_acc) PV_EXACT3(8, 1, false); else PV_EXACT3(8, 2, false); }
            break;
        default: return PVNET_E_UNSUPPORTED;
    }
#undef PV_EXACT3
#else
    if (!P.fold1 || (mh == 8 && !one_acc) || (timed && mh != 8)) return PVNET_E_UNSUPPORTED;   // development-build variants
    switch (mh) {
        case 1: hipLaunchKernelGGL((ScoreExact<1, 1, false, 2, false>::kernel), g, t, lds, s, P); break;
        case 2: hipLaunchKernelGGL((ScoreExact<2, 1, false, 2, false>::kernel), g, t, lds, s, P); break;
        case 4: hipLaunchKernelGGL((ScoreExact<4, 1, false, 2, false>::kernel), g, t, lds, s, P); break;
        case 8:
            if (runs) {
                if (timed) hipLaunchKernelGGL((ScoreExact<8, 1, true, 1, true>::kernel), g, t, lds, s, P);
                else hipLaunchKernelGGL((ScoreExact<8, 1, false, 1, true>::kernel), g, t, lds, s, P);
            } else {
                if (timed) hipLaunchKernelGGL((ScoreExact<8, 1, true, 1, false>::kernel), g, t, lds, s, P);
                else hipLaunchKernelGGL((ScoreExact<8, 1, false, 1, false>::kernel), g, t, lds, s, P);
            }
            break;
        default: return PVNET_E_UNSUPPORTED;
    }
#endif
    return 0;
}

}  // namespace pvd
