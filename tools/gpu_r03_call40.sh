# grid size / stream count once more, with the co-resident small stages (final tree)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c40
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 20"
run() { tag=$1; st=$2; shift; shift; env "$@" timeout 300 $B --streams $st 2>/dev/null | tail -1 > $O/$tag.json; }
run w8_s6 6 PVNET_NOOP=1
run w6_s6 6 PVNET_SCORE_WGS_PER_CU=6
run w12_s6 6 PVNET_SCORE_WGS_PER_CU=12
run w16_s6 6 PVNET_SCORE_WGS_PER_CU=16
run w8_s8 8 PVNET_NOOP=1
run w8_s4 4 PVNET_NOOP=1
run w8_s12 12 PVNET_NOOP=1
run w8_s6b 6 PVNET_NOOP=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c40/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f  approx %8.0f  score %.1f us  spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
