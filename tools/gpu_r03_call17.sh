# overlap experiments, six-stream bench: hypothesis tiles per wave (register room for the other streams' small stages), streams
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c17
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-parity --regions 7 --steps 50 --warmup 5 --score-repeats 40"
run() { tag=$1; st=$2; shift; shift; env "$@" timeout 300 $B --streams $st 2>/dev/null | tail -1 > $O/$tag.json; }
run base 6 PVNET_NOOP=1
run hpl4 6 PVNET_SCORE_HPL=4
run hpl4_s8 8 PVNET_SCORE_HPL=4
run s4 4 PVNET_NOOP=1
run s8 8 PVNET_NOOP=1
run s12 12 PVNET_NOOP=1
run lds64_s8 8 PVNET_SCORE_LDS_KB=64
run hpl4_lds 6 PVNET_SCORE_HPL=4 PVNET_SCORE_LDS_KB=48
run base2 6 PVNET_NOOP=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c17/*.json")):
    try:
        j = json.loads(open(f).read())
        print("%-10s value %8.0f  step %.4f ms  single %8.0f  approx %8.0f  score %.1f us  spread %.3f" % (
            f.split("/")[-1][:-5], j["value"], j["ms_per_step"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j["roofline"]["avg_launch_ms"] * 1e3, j["regions"]["spread"]))
    except Exception as e:
        print(f, "ERR", e)
PY
