#!/bin/bash
# round 6, GPU call 17: counters of the culling body (every key-point culled: noisy benchmark field and clean field) beside round 5's
# 25.0 M bank-conflict cycles; 2 000 knob-fuzz cases on the split build (development library); the whole GPU suite; bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r06cull; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary --streams 1 --prewarm-seconds 0.2 --regions 1"
export PVNET_SCORE_CULL=1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o trace -- $B --steps 200 --warmup 20 > $OUT/trace_bench1.json 2> $OUT/trace1.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $OUT/pmc_lds1 -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_lds1.json 2> $OUT/pmc_lds1.err )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu1 -o pmc -- $B --steps 10 --warmup 2 --score-repeats 4 > $OUT/pmc_valu1.json 2> $OUT/pmc_valu1.err )
unset PVNET_SCORE_CULL
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
grep -E "score_exact_kernel|hypothesis_" $OUT/summary_rocprof_summary.txt | cut -c1-200
find $OUT -name '*.db' -size +1M -delete
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
( timeout 2400 python tools/fuzz_parity.py 2000 2>&1 | grep -v amdgpu.ids | tail -25 ) > $O/fuzz_exact.txt; tail -3 $O/fuzz_exact.txt
python bench.py > $O/bench.json 2> $O/bench.err; python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench.json", "bench_driver_form.json"):
    d = json.loads(open("gpurun_out/r06q/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["single_stream"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["pass"], d["secondary"]["pass"])
PY
