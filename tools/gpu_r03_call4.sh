# round 3, GPU call 4: exact mode after the LDS / band changes -- tests, probe, whole GPU suite, smoke, bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c4
mkdir -p $O
timeout 900 python -m pytest tests/test_exact_mode.py -x -q -m gpu > $O/pytest_exact.txt 2>&1; echo "exact rc=$?"; tail -5 $O/pytest_exact.txt
timeout 600 python tools/exact_probe.py > $O/exact_probe.txt 2>&1; cat $O/exact_probe.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_exact_mode.py > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -30 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03c4/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "regions", "single_stream", "approx_mode", "literal_mode", "parity", "stage_ms"):
    print(k, json.dumps(j.get(k))[:700])
print("roofline", json.dumps({k: v for k, v in j["roofline"].items() if k not in ("note", "timing", "definition")}))
PY
