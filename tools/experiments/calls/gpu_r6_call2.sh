#!/bin/bash
# round 6, GPU call 2: per-key-point disc culling (K3 selects; merged K3 launch; counts through perm; new LDS tile layout) -- parity
# first, then where culling pays
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06b; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_disc_culling.py -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_cull.txt; tail -15 $O/pytest_cull.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt
( timeout 900 python tools/cull_crossover.py quick > $O/cull_crossover_quick.txt 2>&1 ); cat $O/cull_crossover_quick.txt
( timeout 600 python tools/cull_probe.py quick > $O/cull_probe_quick.txt 2>&1 ); cat $O/cull_probe_quick.txt
