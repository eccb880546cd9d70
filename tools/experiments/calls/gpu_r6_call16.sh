#!/bin/bash
# round 6, GPU call 16: evidence set of the split build -- rocprofv3 summary + counters (one stream), end-to-end with the stand-in backbone,
# the reference's call configurations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/gpu_profile.sh r06p > gpurun_out/prof_r06p.log 2>&1
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; mkdir -p $O
( timeout 900 python tools/e2e_amd.py 2>&1 | grep -v amdgpu.ids ) > $O/e2e.txt; cat $O/e2e.txt
( timeout 900 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids ) > $O/bench_configs.txt; tail -32 $O/bench_configs.txt
