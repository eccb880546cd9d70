# the driver's own forms on the final tree: bench with --steps 20 --warmup 5, and the torch.distributed.run launch at N = 1
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c29
mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err; echo "rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/launch_form.json 2> $O/launch_form.err; echo "rc=$?"
python - <<'PY'
import json
for f in ("driver_form", "launch_form"):
    try:
        j = json.loads(open(f"gpurun_out/r03c29/{f}.json").read().strip().splitlines()[-1])
        print(f, "value %.0f  step %.4f ms  spread %.3f  single %.0f  approx %.0f  parity %s  ranks %s  sclk %s" % (
            j["value"], j["ms_per_step"], j["regions"]["spread"], j["single_stream"]["value"], j["approx_mode"]["value"],
            j.get("parity", {}).get("pass"), j.get("rccl_ranks_seen"), j["regions"]["gpu"]["sclk_mhz"]["mean"]))
        print("  roofline frac %.4f avg_launch_ms %.4f traffic %s mfma_busy %s" % (j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["roofline"]["traffic"], j["roofline"]["mfma_busy_frac"]))
    except Exception as e:
        print(f, "ERR", e)
PY
