#!/bin/bash
# round 6, GPU call 11: the merged scoring launch -- whole GPU suite, crossover sweep (selection threshold), item mapping of the culling body
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
( timeout 600 python tools/experiments/stage_ab.py clean --rounds 2 M4=pvnet_amd/libpvnet_vote.so M4runs=pvnet_amd/libpvnet_vote.so,PVNET_SCORE_RUNS=1 M4all=pvnet_amd/libpvnet_vote.so,PVNET_SCORE_CULL=1 2>&1 | grep -v amdgpu.ids ) > $O/stage_ab_runs.txt; cat $O/stage_ab_runs.txt
( timeout 1500 python tools/cull_crossover.py 2>&1 | grep -v amdgpu.ids > $O/cull_crossover.txt ); cat $O/cull_crossover.txt
