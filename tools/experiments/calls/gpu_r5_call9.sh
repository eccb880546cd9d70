#!/bin/bash
# round 5, GPU call 9: rocprofv3 on the OPT-IN disc-culling path (PVNET_SCORE_CULL=1): per-kernel durations and the LDS counters of
# the gathered fine pass beside the full kernel's, to put numbers behind "a gathered step costs 1.9 x" 
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05cull; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary --streams 1 --prewarm-seconds 0.2 --regions 1"
for cull in 1 0; do
  export PVNET_SCORE_CULL=$cull
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace$cull -o trace -- $B --steps 200 --warmup 20 > $OUT/trace_bench$cull.json 2> $OUT/trace$cull.err )
  ( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $OUT/pmc_lds$cull -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_lds$cull.json 2> $OUT/pmc_lds$cull.err )
  ( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu$cull -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_valu$cull.json 2> $OUT/pmc_valu$cull.err )
done
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
grep -E "score_exact_kernel|hypothesis_" $OUT/summary_rocprof_summary.txt | cut -c1-200
find $OUT -name '*.db' -size +1M -delete
