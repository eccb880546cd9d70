#!/bin/bash
# Round 4: which neighbour does the compaction flake need?    Run on the MI355X from the repo root.
cd tools/experiments/k2_flake
echo "== environment"
/opt/rocm/bin/hipcc --version 2>/dev/null | head -2
cat /opt/rocm/.info/version 2>/dev/null
rocm-smi --showfwinfo 2>/dev/null | grep -i "MEC\|SDMA\|SMC\|RLC\|CP\|VBIOS\|firmware" | head -14
rocm-smi --showdriverversion 2>/dev/null | grep -i version
uname -r
echo "== original reproducer (24 VGPRs used = allocated): tight / spare"
./k2_repro_tight.bin 300 | tail -4 | head -1
./k2_repro_spare.bin 300 | tail -4 | head -1
echo "== round-4 variants of the same kernel (k2_repro_r4.hip: 28 used of 32 allocated unless noted)"
for v in tight spare tight_2percu tight_1percu dual dual_1percu nt128 nt128_lds40k nt64; do
  echo "-- $v"; ./k2_r4_$v.bin 300 | grep "threads per\|bad runs"
done
