#!/bin/bash
# round 6, GPU call 3: sorted-order count buffer restored (the perm-scattered atomics cost the strided kernel 90 us), K3 phases
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_disc_culling.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_cull.txt; tail -5 $O/pytest_cull.txt
( timeout 300 python tools/experiments/k3_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/k3_probe.txt; cat $O/k3_probe.txt
( timeout 900 python tools/cull_crossover.py quick > $O/cull_crossover_quick.txt 2>&1 ); cat $O/cull_crossover_quick.txt
( timeout 600 python tools/cull_probe.py quick 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/cull_probe_quick.txt ); cat $O/cull_probe_quick.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt
