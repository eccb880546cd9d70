#!/bin/bash
# round 6, GPU call 1 (the tree of round 5's end): what the headline measures -- the six-stream kernel trace (VERDICT r05 "Next" 3),
# the stale secondary evidence refreshed (bench_configs), and the flake bisect (one v_cndmask at a time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06a; mkdir -p $O; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary --prewarm-seconds 0.2"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace6 -o trace -- $B --streams 6 --steps 200 --warmup 20 --regions 3 > $O/trace6_bench.json 2> $O/trace6.err )
DB=$(ls $O/trace6/*/*.db $O/trace6/*.db 2>/dev/null | head -1); echo "db: $DB"
python tools/overlap_summary.py "$DB" $O/overlap.txt $O/overlap.csv | head -60
gzip -f $O/overlap.csv
( timeout 900 python tools/bench_configs.py > $O/bench_configs.txt 2>&1 ); tail -30 $O/bench_configs.txt
( timeout 900 bash tools/experiments/k2_flake/run_r6.sh 200 > $O/flake_bisect.txt 2>&1 ); head -12 $O/flake_bisect.txt
( timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err ); tail -c 1500 $O/bench_driver_form.json
find $O -name '*.db' -size +1M -delete
ls -la $O
