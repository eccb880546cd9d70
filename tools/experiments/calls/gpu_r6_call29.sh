#!/bin/bash
# round 6, GPU call 29: final evidence set -- rocprofv3 summary + counters (tools/gpu_profile.sh), crossover sweeps of the final selection
# (seventh-smallest candidate distance, batch gate from the previous call), the whole GPU suite, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/gpu_profile.sh r07d > gpurun_out/prof_r07d.log 2>&1
O=$GRAFT_REPO_ROOT/gpurun_out/r07d; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
( timeout 2400 python tools/cull_crossover.py 2>&1 | grep -v amdgpu.ids ) > $O/cull_crossover.txt
( timeout 1500 python tools/cull_crossover.py outliers 2>&1 | grep -v amdgpu.ids ) > $O/cull_crossover_outliers.txt
cut -c1-200 $O/cull_crossover.txt | head -24
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err
python - <<'PY'
import json
for f in ("gpurun_out/prof_r07d/bench.json", "gpurun_out/r07d/bench_driver_form.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["single_stream"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["pass"], d["secondary"]["pass"],
          {k: round(v.get("us_per_call", 0), 1) for k, v in d["secondary"]["entries"].items()})
PY
