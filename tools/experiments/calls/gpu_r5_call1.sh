#!/bin/bash
# round 5, GPU call 1: the tightened GPU suite (incl. 200 fuzz cases), the bench line with its new `secondary` block, the flake's
# discriminating experiment and the small-shape layout sweep (item 6).  Everything lands in gpurun_out/r05a/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt | tail -5
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
for f in ("bench.json","bench_driver_form.json"):
    try:
        j=json.load(open("gpurun_out/r05a/"+f)); print(f, "value", round(j["value"]), "single", round(j["single_stream"]["value"]), "score_ms", j["roofline"]["avg_launch_ms"], "parity", j["parity"]["pass"], "secondary", {k:(round(v["us_per_call"],1), round(v["score_us"],1), v["pass"]) for k,v in j["secondary"]["entries"].items()} if "entries" in j.get("secondary",{}) else j.get("secondary"))
    except Exception as e: print(f, "unreadable", e)
PY
bash tools/experiments/k2_flake/run_r5.sh 300 > $O/flake_r5.txt 2>&1; cat $O/flake_r5.txt
timeout 600 python tools/shape_sweep.py quick > $O/shape_sweep.txt 2>&1; cat $O/shape_sweep.txt
