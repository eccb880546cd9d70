#!/bin/bash
# round 6, GPU call 42: the six-stream kernel trace of the FINAL tree (the merged scoring launch), taken apart as r06a was
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r07p; mkdir -p $O; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-secondary --prewarm-seconds 0.2"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace6 -o trace -- $B --streams 6 --steps 200 --warmup 20 --regions 3 > $O/trace6_bench.json 2> $O/trace6.err )
DB=$(ls $O/trace6/*/*.db $O/trace6/*.db 2>/dev/null | head -1); echo "db: $DB"
python tools/overlap_summary.py "$DB" $O/overlap.txt $O/overlap.csv | head -40
gzip -f $O/overlap.csv
find $O -name '*.db' -size +1M -delete
