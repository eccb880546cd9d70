cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
PVNET_SCORE_ATOMIC=1 python -m pytest tests/test_hip_parity.py tests/test_fast_mode_parity.py tests/test_reference_kernels.py -m gpu -q 2>&1 | tail -4
for x in 0 1 0 1; do PVNET_SCORE_ATOMIC=$x python bench.py --no-cpu-baseline --no-parity > gpurun_out/r02h/bench_atomic${x}_$RANDOM.json 2>> gpurun_out/r02h/bench.err; done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r02h/bench_atomic*.json')):
    d = json.load(open(f)); print(f, round(d['value']), 'single', round(d['single_stream']['ms_per_step']*1e3,1), 'score', round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['avg_launch_ms_back_to_back_events']*1e3,1), {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})
PY
python tools/bench_configs.py 2>&1 | grep -i "call site"
