#!/bin/bash
# round 5, GPU call 12: the final tree -- both GPU suites, then the evidence set r05h (tools/experiments/calls/gpu_r5_call7.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
( PVNET_SCORE_CULL=1 timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 ) > $O/pytest_gpu_cull.txt; tail -2 $O/pytest_gpu_cull.txt
bash tools/experiments/calls/gpu_r5_call7.sh
