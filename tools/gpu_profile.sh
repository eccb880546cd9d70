# rocprofv3 evidence for bench.py (run on the GPU box via gpurun).  Usage: bash tools/gpu_profile.sh <tag> [quick]
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary --streams 1 --prewarm-seconds 0.2 --regions 1"   # (the profiled runs time the headline workload only: the secondary block runs other shapes through the same kernels)
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
# per-kernel durations (kernel trace only), then the counter passes, each in its own run (never together with other trace domains)
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B --steps 200 --warmup 20 > $OUT/trace_bench.json 2> $OUT/trace.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_write.json 2> $OUT/pmc_write.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err )
if [ "$2" != "quick" ]; then
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc_lds -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err )
fi
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
# the raw rocpd databases are tens of MiB each (gpurun merges at most 64 MiB back): keep the summaries only
find $OUT -name '*.db' -size +1M -delete
ls -la $OUT
