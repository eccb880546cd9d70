# small stages with 256-thread workgroups (fit beside the one-accumulator scoring kernel): _ab/alt against the tree, ACC = 1 / 2
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c27
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 40"
run() { tag=$1; dir=$2; shift; shift; (cd $dir && env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/$tag.json); }
run tree_acc2 . PVNET_SCORE_ACC=2
run tree_acc1 . PVNET_SCORE_ACC=1
run k8rt256_acc1 _ab/k8_rt256 PVNET_SCORE_ACC=1
run k8rt256_acc2 _ab/k8_rt256 PVNET_SCORE_ACC=2
run k8rt512_acc1 _ab/k8_rt512 PVNET_SCORE_ACC=1
run tree_acc2b . PVNET_SCORE_ACC=2
run tree_acc1b . PVNET_SCORE_ACC=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c27/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f (%.4f ms) approx %8.0f  score %.1f us  spread %.3f  stages %s" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["single_stream"]["ms_per_step"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["regions"]["spread"], {k: round(v * 1e3, 1) for k, v in j["stage_ms"].items()}))
    except Exception as e:
        print(f, "ERR", e)
PY
