#!/bin/bash
# round 6, GPU call 4: the K3 sort on registers / DPP / permlane swaps; where the culling kernel's fixed cost goes (phase probe)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_disc_culling.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_cull.txt; tail -5 $O/pytest_cull.txt
( timeout 300 python tools/experiments/k3_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/k3_probe.txt; cat $O/k3_probe.txt
( for f in clean noisy; do PVNET_SCORE_CULL=1 timeout 300 python tools/phase_probe.py $f; PVNET_SCORE_CULL=0 timeout 300 python tools/phase_probe.py $f; done 2>&1 | grep -v amdgpu.ids ) > $O/phase_probe.txt; cat $O/phase_probe.txt
( timeout 900 python tools/cull_crossover.py quick 2>&1 | grep -v amdgpu.ids > $O/cull_crossover_quick.txt ); cat $O/cull_crossover_quick.txt
