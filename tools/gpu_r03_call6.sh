# round 3, GPU call 6: whole GPU suite on the exact-mode tree, probe, smoke, bench (driver form)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c6
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -15 $O/pytest_gpu.txt
timeout 600 python tools/exact_probe.py --quick > $O/exact_probe.txt 2>&1; cat $O/exact_probe.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03c6/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_stream", "approx_mode", "literal_mode", "stage_ms"):
    print(k, json.dumps(j.get(k))[:400])
r = j["regions"]; print("regions", r["runs"], r["min"], r["max"], r["spread"], r["gpu"])
p = j["parity"]; print("parity", {k: p[k] for k in p if k not in ("reference", "mode")})
print("roofline", json.dumps({k: v for k, v in j["roofline"].items() if k not in ("note", "timing", "definition")}))
PY
