# residency / grid experiments for the exact scoring kernel in the 6-stream bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c13
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 40"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/$tag.json; }
run base PVNET_NOOP=1
run lds64 PVNET_SCORE_LDS_KB=64
run lds50 PVNET_SCORE_LDS_KB=50
run wgs3 PVNET_SCORE_WGS_PER_CU=3
run wgs6 PVNET_SCORE_WGS_PER_CU=6
run wgs12 PVNET_SCORE_WGS_PER_CU=12
run wgs0 PVNET_SCORE_WGS_PER_CU=0
run base2 PVNET_NOOP=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c13/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f  approx %8.0f  score %.1f us  spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
