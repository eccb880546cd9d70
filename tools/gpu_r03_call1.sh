# round 3, GPU call 1: epilogue micro-benchmark for the exact mode + same-box A/B of the r01 head against this tree
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c1
O=gpurun_out/r03c1
rocminfo | grep -E "Marketing|gfx" | head -2
timeout 300 tools/ubench_exact.bin > $O/ubench_exact.txt 2>&1; cat $O/ubench_exact.txt
for i in 1 2 3; do
  ( cd _ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > ../../$O/ab_r01_$i.json )
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/ab_head_$i.json
done
python - <<'PY'
import json, glob
for tag in ("r01", "head"):
    for f in sorted(glob.glob(f"gpurun_out/r03c1/ab_{tag}_*.json")):
        try:
            j = json.loads(open(f).read())
            print(tag, f, round(j["value"]), j["ms_per_step"], (j.get("single_stream") or {}).get("value"))
        except Exception as e:
            print(tag, f, "ERR", e)
PY
