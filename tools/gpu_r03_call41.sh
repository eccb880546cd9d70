# timeline of the six-stream steady state: what runs while no scoring kernel is resident
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c41
mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --regions 3 --steps 300 --warmup 20 --score-repeats 4 > $O/bench.json 2> $O/bench.err )
DB=$(find $O/trace -name '*.db' | head -1)
python tools/timeline_probe.py $DB 0.25 0.45 > $O/timeline.txt 2>&1
cat $O/timeline.txt
( cd /tmp && BENCH_GC=0 rocprofv3 --kernel-trace -d $O/trace_nogc -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --regions 3 --steps 300 --warmup 20 --score-repeats 4 > $O/bench_nogc.json 2> $O/bench_nogc.err )
DB2=$(find $O/trace_nogc -name '*.db' | head -1)
python tools/timeline_probe.py $DB2 0.25 0.45 > $O/timeline_nogc.txt 2>&1
cat $O/timeline_nogc.txt
find $O -name '*.db' -size +1M -delete
