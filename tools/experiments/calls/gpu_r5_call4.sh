#!/bin/bash
# round 5, GPU call 4: the whole GPU suite on the final kernel sources (default layout, then with PVNET_SCORE_CULL=1), the rocprofv3
# evidence set of the default path (tools/gpu_profile.sh r05f) and the bench line in the driver's form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 2>&1 | tail -40 ) > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt | cut -c1-400
( PVNET_SCORE_CULL=1 timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 2>&1 | tail -40 ) > $O/pytest_gpu_cull.txt; tail -8 $O/pytest_gpu_cull.txt | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/gpu_profile.sh r05f > $O/profile.log 2>&1; tail -3 $O/profile.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
for f in ("gpurun_out/r05d/bench_driver_form.json","gpurun_out/prof_r05f/bench.json"):
    try:
        j=json.load(open(f)); print(f, "value", round(j["value"]), "single", round(j["single_stream"]["value"]), "score_ms", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "busy", j["roofline"]["mfma_busy_frac"], "parity", j["parity"].get("pass"), "secondary", j.get("secondary",{}).get("pass"))
    except Exception as e: print(f, "unreadable", e)
PY
