set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c9
mkdir -p $O
timeout 900 python -m pytest tests/test_exact_mode.py tests/test_fast_mode_parity.py tests/test_flake_canary.py -x -q -m gpu -rx > $O/pytest_exact.txt 2>&1; echo "exact rc=$?"; tail -8 $O/pytest_exact.txt
timeout 600 python tools/exact_probe.py --quick > $O/exact_probe.txt 2>&1; cat $O/exact_probe.txt
timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "threads or baseline" > $O/pytest_threads.txt 2>&1; tail -3 $O/pytest_threads.txt
