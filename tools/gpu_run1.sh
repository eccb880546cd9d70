set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4
nproc
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
timeout 120 tools/ubench_valu.bin > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 5 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench1.log
