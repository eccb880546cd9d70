# round 3, GPU call 2: first run of the exact scoring mode -- its own tests, a timing probe, the whole GPU suite, bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c2
mkdir -p $O
timeout 900 python -m pytest tests/test_exact_mode.py -x -q -m gpu > $O/pytest_exact.txt 2>&1; echo "exact rc=$?"; tail -25 $O/pytest_exact.txt
timeout 600 python tools/exact_probe.py > $O/exact_probe.txt 2>&1; cat $O/exact_probe.txt
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_exact_mode.py > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -40 $O/pytest_gpu.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
