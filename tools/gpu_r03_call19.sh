# round 3, GPU call 19: lazy literal gate + rows staged in registers -- whole GPU suite, then A/B against the previous head on this box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c19
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
B="bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 60"
for rep in 1 2; do
  (cd _ab/prev && timeout 300 python $B 2>/dev/null | tail -1 > ../../$O/prev_$rep.json)
  timeout 300 python $B 2>/dev/null | tail -1 > $O/new_$rep.json
done
(cd _ab/prev && timeout 300 python tools/exact_probe.py --quick 2>/dev/null > ../../$O/probe_prev.txt); timeout 300 python tools/exact_probe.py --quick 2>/dev/null > $O/probe_new.txt
cat $O/probe_prev.txt $O/probe_new.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c19/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f  approx %8.0f  score %.1f us (b2b %.1f) spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["roofline"]["avg_launch_ms_back_to_back_events"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
