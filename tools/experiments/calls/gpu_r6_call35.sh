#!/bin/bash
# round 6, GPU call 35: 8 000 further knob-fuzz cases on the final tree (development build: culling / layout knobs flipped at random),
# the whole GPU suite and the smoke entry for the record
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r07j; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
( python __graft_entry__.py smoke 2>&1 | tail -1 ) > $O/smoke.txt; cat $O/smoke.txt
( timeout 3300 python tools/fuzz_parity.py 10000 2000 2>&1 | grep -v amdgpu.ids | tail -30 ) > $O/fuzz_exact.txt; tail -4 $O/fuzz_exact.txt
