set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log; tail -40 gpurun_out/r02b/pytest.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02b/bench.json')); print(d['value'], d['roofline'])"
