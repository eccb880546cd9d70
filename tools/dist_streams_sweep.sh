#!/bin/bash
# bench.py's value against the number of streams in the DISTRIBUTED form at world = 1 (an initialised RCCL communicator changes
# the stream-to-hardware-queue mapping), then the plain form for comparison; same box, interleaved rounds:
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_call.sh <tag> "bash tools/dist_streams_sweep.sh"'
for r in 0 1; do for s in 5 6 7 8; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520+s+10*r)) bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity --regions 5 --score-repeats 20 --streams $s 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dist round $r streams %2d  value %7.1f k  spread %.1f %%' % ($s, d['value']/1e3, d['regions']['spread']*100), flush=True)"
done; done
python bench.py --no-cpu-baseline --no-parity --regions 5 --score-repeats 20 --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('plain streams 6 value %7.1f k' % (d['value']/1e3))"
