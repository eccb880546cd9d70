# what the distributed form costs at world = 1: plain against torch.distributed.run, K = 20 and K = 200
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c33
mkdir -p $O
for K in 20 200; do
  timeout 600 python bench.py --gpus 1 --steps $K --warmup 5 --no-cpu-baseline --no-parity --regions 9 > $O/plain_$K.json 2> $O/plain_$K.err
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps $K --warmup 5 --no-cpu-baseline --no-parity --regions 9 > $O/dist_$K.json 2> $O/dist_$K.err
done
python - <<'PY'
import json
for f in ("plain_20", "dist_20", "plain_200", "dist_200"):
    try:
        j = json.loads(open(f"gpurun_out/r03c33/{f}.json").read().strip().splitlines()[-1])
        print(f, "value %.0f  step %.4f ms  spread %.3f  single %.0f  approx %.0f  gather_ms %s" % (
            j["value"], j["ms_per_step"], j["regions"]["spread"], j["single_stream"]["value"], j["approx_mode"]["value"], j.get("gather_ms", j.get("config", {}).get("gather_ms"))))
    except Exception as e:
        print(f, "ERR", e)
PY
