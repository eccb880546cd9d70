set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log; tail -15 gpurun_out/r02a/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_driver_form.json 2> gpurun_out/r02a/bench_driver_form.err; tail -c 1500 gpurun_out/r02a/bench_driver_form.json
for x in 1 0 1 0; do PVNET_SCORE_XCD=$x python bench.py --no-cpu-baseline --no-parity > gpurun_out/r02a/bench_xcd${x}_$RANDOM.json 2>> gpurun_out/r02a/bench_xcd.err; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r02a/bench_xcd*.json')):
    d = json.load(open(f)); print(f, round(d['value']), d['single_stream']['ms_per_step'], d['roofline']['avg_launch_ms'], d['stage_ms'])
PY
bash tools/gpu_profile.sh r02a quick > gpurun_out/r02a/profile.log 2>&1; tail -5 gpurun_out/r02a/profile.log
python tools/bench_configs.py > gpurun_out/r02a/bench_configs.txt 2>&1; cat gpurun_out/r02a/bench_configs.txt
