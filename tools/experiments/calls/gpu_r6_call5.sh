#!/bin/bash
# round 6, GPU call 5: per-image culling selection (majority of the key-points' votes), next-item record prefetch, coarse-pass MFMAs
# issued together, list lengths in four 16-byte reads
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_disc_culling.py tests/test_exact_mode.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_cull.txt; tail -5 $O/pytest_cull.txt
( for f in clean noisy; do PVNET_SCORE_CULL=1 timeout 300 python tools/phase_probe.py $f; done 2>&1 | grep -v amdgpu.ids ) > $O/phase_probe.txt; cat $O/phase_probe.txt
( timeout 900 python tools/cull_crossover.py quick 2>&1 | grep -v amdgpu.ids > $O/cull_crossover_quick.txt ); cat $O/cull_crossover_quick.txt
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids > $O/bench.json ); head -c 1500 $O/bench.json
