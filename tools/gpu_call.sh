#!/bin/bash
# One GPU-box call of the development loop (replaces the per-call scripts of round 3):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_call.sh <tag> <command> [; <command> ...]'
# runs the commands from the repo root with TMPDIR=/tmp and tees everything into gpurun_out/<tag>/log.txt (merged back).
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp OUT=$O
bash -c "$*" 2>&1 | tee "$O/log.txt"
find "$O" -name '*.db' -size +1M -delete
exit ${PIPESTATUS[0]}
