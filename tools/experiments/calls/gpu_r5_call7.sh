#!/bin/bash
# round 5, GPU call 7: the rocprofv3 evidence set of the final kernel sources (headline workload only) + the bench line in the driver's form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp
bash tools/gpu_profile.sh r05h > $O/profile.log 2>&1; tail -3 $O/profile.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
for f in ("gpurun_out/r05h/bench_driver_form.json","gpurun_out/prof_r05h/bench.json"):
    try:
        j=json.load(open(f)); print(f, "value", round(j["value"]), "single", round(j["single_stream"]["value"]), "score_ms", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"], "parity", j["parity"].get("pass"), "secondary", j.get("secondary",{}).get("pass"))
    except Exception as e: print(f, "unreadable", e)
PY
head -12 gpurun_out/prof_r05h/summary_rocprof_summary.txt
