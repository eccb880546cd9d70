#!/bin/bash
# Round 6, level 2 of the bisect: ranges of block LBB0_25's v_cndmask sites (bisect_cndmask.py)      usage (MI355X): bash run_r6b.sh [launches]
cd "$(dirname "$0")"
N=${1:-300}
hipcc -O2 -w co_runner.cpp -o /tmp/co_runner.bin || exit 1
for f in r6_co/base.co r6_co/all.co r6_co/blk_LBB0_25.co r6_co/allbut_LBB0_25.co r6_co/rng_*.co r6_co/alt*_LBB0_25.co; do
  printf "%-28s %s\n" "$(basename $f .co)" "$(timeout 120 /tmp/co_runner.bin $f $N 2>&1 | grep -i 'bad runs' | tail -1)"
done
