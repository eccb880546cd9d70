#!/bin/bash
# round 5, GPU call 3: the GPU suite with the new disc-culling tests, the culling probe with the 1 024-thread sort kernel, and the
# distributed form at world = 1 with the collective issued from a voting stream
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -40 ) > $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt | cut -c1-600
timeout 900 python tools/cull_probe.py > $O/cull_probe.txt 2>&1; cat $O/cull_probe.txt | cut -c1-520
for rep in 0 1; do
for mode in plain dist_comm dist_vote; do
  case $mode in
    plain)     python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-secondary --regions 9 > $O/g_$mode$rep.json 2> $O/g_$mode$rep.err ;;
    dist_comm) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2950$rep python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-secondary --regions 9 > $O/g_$mode$rep.json 2> $O/g_$mode$rep.err ;;
    dist_vote) BENCH_GATHER_ON_VOTING_STREAM=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$rep python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-secondary --regions 9 > $O/g_$mode$rep.json 2> $O/g_$mode$rep.err ;;
  esac
  python - $O/g_$mode$rep.json $mode $rep <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1])); print("gather-stream A/B", sys.argv[2], "round", sys.argv[3], "six-stream %.1f k  single %.1f k  spread %.1f %%  gather_stream=%s" % (j["value"]/1e3, j["single_stream"]["value"]/1e3, 100*j["regions"]["spread"], j.get("gather_stream")))
except Exception as e: print("unreadable", sys.argv[1], e)
PY
done; done 2>&1 | tee $O/gather_stream_ab.txt
