# cells of two pixel tiles in the one-accumulator exact kernel (FOLD = 2): parity, fuzz, probe, A/B against the previous head
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c36
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 40"
run() { tag=$1; dir=$2; shift; shift; (cd $dir && env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/$tag.json); }
run tree_a . PVNET_NOOP=1
run prio3_a _ab/prio PVNET_NOOP=1
run tree_b . PVNET_NOOP=1
run prio3_b _ab/prio PVNET_NOOP=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c36/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f (%.4f ms) approx %8.0f  score %.1f us  spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["single_stream"]["ms_per_step"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
