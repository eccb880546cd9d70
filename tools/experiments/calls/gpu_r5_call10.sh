#!/bin/bash
# round 5, GPU call 10: the mask kernel with interleaved words and one coalesced bit-word store per segment -- suite, then same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05j; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12 ) > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
python tools/ab.py --rounds 2 before=_ab/lib_r05_k1old.so v1=_ab/lib_r05_k1v1.so after=pvnet_amd/libpvnet_vote.so --no-secondary --steps 300 --warmup 30 2>&1 | tee $O/ab_k1.txt
python - <<'PY' 2>&1 | tee gpurun_out/r05j/stage_times.txt
import os, sys, numpy as np, torch, subprocess
for lib in ("_ab/lib_r05_k1old.so", "_ab/lib_r05_k1v1.so", "pvnet_amd/libpvnet_vote.so", "_ab/lib_r05_k1old.so", "_ab/lib_r05_k1v1.so", "pvnet_amd/libpvnet_vote.so"):
    out = subprocess.run([sys.executable, "tools/experiments/k1_hist_probe.py"], env=dict(os.environ, PVNET_VOTE_LIB=os.path.abspath(lib)), capture_output=True, text=True).stdout
    print(lib, [l for l in out.splitlines() if l.startswith("max_num 30000")][-1][:150])
PY
