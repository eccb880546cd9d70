# one accumulator pair in the exact scoring kernel (PVNET_SCORE_ACC=1): parity, then A/B against the shipped two pairs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c23
mkdir -p $O
PVNET_SCORE_ACC=1 timeout 900 python -m pytest tests/test_exact_mode.py tests/test_fast_mode_parity.py -q -m gpu -x > $O/pytest_acc1.txt 2>&1; echo "acc1 rc=$?"; tail -3 $O/pytest_acc1.txt
timeout 200 tools/ubench_exact.bin 2>&1 | tail -3 > $O/ubench_tail.txt; cat $O/ubench_tail.txt
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 60"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/$tag.json; }
run acc2_a PVNET_SCORE_ACC=2
run acc1_a PVNET_SCORE_ACC=1
run acc2_b PVNET_SCORE_ACC=2
run acc1_b PVNET_SCORE_ACC=1
run acc1_k1w4 PVNET_SCORE_ACC=1 PVNET_NOOP=1
PVNET_SCORE_ACC=1 timeout 200 python tools/exact_probe.py --quick 2>/dev/null | head -2 > $O/probe_acc1.txt; timeout 200 python tools/exact_probe.py --quick 2>/dev/null | head -2 > $O/probe_acc2.txt
cat $O/probe_acc1.txt $O/probe_acc2.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c23/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f  approx %8.0f  score %.1f us (b2b %.1f) spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["roofline"]["avg_launch_ms_back_to_back_events"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
