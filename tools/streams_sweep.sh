#!/bin/bash
# value of bench.py against the number of HIP streams the batches are issued on (same box, interleaved rounds):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_call.sh <tag> "bash tools/streams_sweep.sh 4 5 6 7 8"'
for r in 0 1; do
  for s in "$@"; do
    python bench.py --no-cpu-baseline --no-parity --regions 5 --score-repeats 20 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r streams %2d  value %7.1f k  spread %.1f %%' % ($s, d['value']/1e3, d['regions']['spread']*100), flush=True)"
  done
done
