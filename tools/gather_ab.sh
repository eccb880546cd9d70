#!/bin/bash
# Same-box A/B (VERDICT r05 "Next" 4): what the distributed form costs at world = 1 -- bench.py plain, under torch.distributed.run with the
# library's own ncclAllGather on the voting streams (--gather rccl), and with torch.distributed's all-gather (--gather torch: rounds 1-5).
#   bash tools/gather_ab.sh [rounds]      -> one line per run: six-stream rate, gather mode
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=${1:-2}
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:22s} six-stream {d['value'] / 1e3:7.1f} k   ms/step {d['ms_per_step']:.4f}   gather_ms {d.get('gather_ms')}   ranks seen {d.get('rccl_ranks_seen')}   spread {d['regions']['spread'] * 100:.1f} %")
PY
}
for r in $(seq 1 "$R"); do
  python bench.py --no-cpu-baseline --no-parity --no-secondary --regions 7 > /tmp/g_plain.json 2>/dev/null; line "round $r plain" /tmp/g_plain.json
  for g in rccl torch; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + r)) bench.py --gpus 1 --gather $g \
      --no-cpu-baseline --no-parity --no-secondary --regions 7 > /tmp/g_$g.json 2>/dev/null; line "round $r torchrun $g" /tmp/g_$g.json
  done
done
