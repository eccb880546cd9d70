#!/bin/bash
# Round 6: which v_cndmask carries the `s_nop 3` effect?  (bisect_cndmask.py wrote r6_co/*.co in the build container)
# usage (MI355X): bash tools/experiments/k2_flake/run_r6.sh [launches per variant]      -> stdout
cd "$(dirname "$0")"
N=${1:-200}
hipcc -O2 -w co_runner.cpp -o /tmp/co_runner.bin || exit 1
for f in r6_co/base.co r6_co/all.co r6_co/half_lo.co r6_co/half_hi.co r6_co/blk_*.co r6_co/one_*.co r6_co/pre_*.co; do
  printf "%-28s %s\n" "$(basename $f .co)" "$(timeout 120 /tmp/co_runner.bin $f $N 2>&1 | grep -i 'bad runs' | tail -1)"
done
