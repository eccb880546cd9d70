#!/bin/bash
# round 6, GPU call 38 (re-run as r07q after comment-only edits: the committed profile must carry the sources' hash): the evidence set of the final sources (reductions, early gate load): rocprofv3 summary + counters, suite, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/gpu_profile.sh r07q > gpurun_out/prof_r07q.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/prof_r07q/bench_driver_form.json 2>> gpurun_out/prof_r07q/bench.err
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/prof_r07q/pytest_gpu.txt; cat gpurun_out/prof_r07q/pytest_gpu.txt
head -12 gpurun_out/prof_r07q/summary_rocprof_summary.txt | cut -c1-170
python - <<'PY'
import json
for f in ("gpurun_out/prof_r07q/bench.json", "gpurun_out/prof_r07q/bench_driver_form.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["single_stream"]["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["pass"], d["secondary"]["pass"],
          {k: round(v.get("us_per_call", 0), 1) for k, v in d["secondary"]["entries"].items()})
PY
