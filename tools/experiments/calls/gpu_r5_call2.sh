#!/bin/bash
# round 5, GPU call 2: the disc-culling kernel (PVNET_SCORE_CULL=1) -- probe against the full kernel, the whole GPU suite, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/cull_probe.py ${1:-quick} > $O/cull_probe.txt 2>&1; cat $O/cull_probe.txt | tail -30
( PVNET_SCORE_CULL=1 timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -60 ) > $O/pytest_gpu_cull.txt; tail -30 $O/pytest_gpu_cull.txt | cut -c1-400
PVNET_SCORE_CULL=1 python bench.py > $O/bench_cull.json 2> $O/bench_cull.err
python - <<'PY'
import json
for f in ("bench_cull.json",):
    try:
        j=json.load(open("gpurun_out/r05b/"+f)); print(f, "value", round(j["value"]), "single", round(j["single_stream"]["value"]), "score_ms", j["roofline"]["avg_launch_ms"], "parity", j["parity"].get("pass"), j["parity"].get("counts_equal_literal"), j["parity"].get("counts_equal_reference"), "stage_ms", j["stage_ms"])
    except Exception as e: print(f, "unreadable", e)
PY
tail -5 $O/bench_cull.err
