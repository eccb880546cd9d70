# bisecting the distributed form's overhead at world = 1
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c47
mkdir -p $O
A="bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-parity --regions 7"
timeout 300 python $A 2>/dev/null | tail -1 > $O/plain.json
OMP_NUM_THREADS=1 timeout 300 python $A 2>/dev/null | tail -1 > $O/plain_omp1.json
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 timeout 300 python $A 2>/dev/null | tail -1 > $O/env_dist.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $A 2>/dev/null | tail -1 > $O/torchrun.json
python - <<'PY'
import json
for f in ("plain", "plain_omp1", "env_dist", "torchrun"):
    try:
        j = json.loads(open(f"gpurun_out/r03c47/{f}.json").read())
        print(f, "value %.0f  step %.4f ms  spread %.3f  single %.0f  approx %.0f  ranks %s" % (
            j["value"], j["ms_per_step"], j["regions"]["spread"], j["single_stream"]["value"], j["approx_mode"]["value"], j.get("rccl_ranks_seen")))
    except Exception as e:
        print(f, "ERR", e)
PY
