#!/bin/bash
# round 6, GPU call 32: the evidence set once more (boxes of the pool differ by +-4 %: r07d landed on a slow one)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/gpu_profile.sh r07g quick > gpurun_out/prof_r07g.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/prof_r07g/bench_driver_form.json 2>> gpurun_out/prof_r07g/bench.err
head -6 gpurun_out/prof_r07g/summary_rocprof_summary.txt | cut -c1-170
python - <<'PY'
import json
for f in ("gpurun_out/prof_r07g/bench.json", "gpurun_out/prof_r07g/bench_driver_form.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["single_stream"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["pass"], d["secondary"]["pass"])
PY
