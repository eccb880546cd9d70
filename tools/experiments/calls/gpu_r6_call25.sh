#!/bin/bash
# round 6, GPU call 25: the culling selection on the seventh-smallest candidate distance (threshold 1.0): crossover sweeps again,
# the culling tests, same-box A/B of the headline against the previous selection
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_disc_culling.py tests/test_exact_mode.py -m gpu -q -x 2>&1 | tail -3 ) | tee $O/pytest.txt
( timeout 1500 python tools/cull_crossover.py outliers 2>&1 | grep -v amdgpu.ids ) > $O/cull_crossover_outliers.txt; cut -c1-200 $O/cull_crossover_outliers.txt
( timeout 2400 python tools/cull_crossover.py 2>&1 | grep -v amdgpu.ids ) > $O/cull_crossover.txt; cut -c1-200 $O/cull_crossover.txt
