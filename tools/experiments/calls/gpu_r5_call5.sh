#!/bin/bash
# round 5, GPU call 5: deferred re-evaluation of the flagged cells (the scoring kernel lists them, select_refine decides them) --
# the GPU suite, then a same-box A/B against the library built before the change, then the secondary configurations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 2>&1 | tail -40 ) > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt | cut -c1-500
python tools/ab.py --rounds 2 before=_ab/lib_r05_before_dr.so after=pvnet_amd/libpvnet_vote.so --no-secondary --steps 300 --warmup 30 2>&1 | tee $O/ab_dr.txt
python tools/exact_probe.py --quick 2>&1 | tail -12 | tee $O/exact_probe.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05e/bench_driver_form.json")); print("driver form: value", round(j["value"]), "single", round(j["single_stream"]["value"]), "score_ms", j["roofline"]["avg_launch_ms"], "parity", j["parity"].get("pass"), "stage_ms", {k: round(v*1e3,1) for k,v in j["stage_ms"].items()}, "secondary", {k:(round(v["us_per_call"],1), round(v["score_us"],1), v["pass"]) for k,v in j["secondary"]["entries"].items()})
PY
