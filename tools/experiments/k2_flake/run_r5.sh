#!/bin/bash
# Round 5: one more discriminating experiment on the wrong-VGPR flake, then file it.
#  (a) the SAME failing binary under HSA_XNACK=0 / 1 (page-fault retry mode changes how waves are launched / replayed), and the
#      source rebuilt under amdgpu_waves_per_eu(1,1) -- does a changed occupancy hint / kernel descriptor move it?
#  (b) is rocgdb on the box (to read a failing wave's state)?  The failure is only visible on the host after the launch, so
#      without a device-side trap there is nothing to stop at: reported as available / not available.
# usage (MI355X): bash tools/experiments/k2_flake/run_r5.sh [runs]     -> stdout
cd "$(dirname "$0")"
N=${1:-300}
hipcc -O3 --offload-arch=gfx950 -w k2_repro.hip -o /tmp/k2_tight.bin
hipcc -O3 --offload-arch=gfx950 -w k2_repro.hip -o /tmp/k2_spare.bin -DV_SPARE
hipcc -O3 --offload-arch=gfx950 -w k2_repro.hip -o /tmp/k2_wpe1.bin -DV_WPE1
hipcc -O3 --offload-arch=gfx950 -w k2_repro.hip -S --cuda-device-only -o /tmp/k2_wpe1.s -DV_WPE1
echo "descriptor of the waves_per_eu(1,1) build:"; grep -E "amdhsa_next_free_vgpr|amdhsa_accum_offset|amdhsa_granulated|; Occupancy|; NumVgprs" /tmp/k2_wpe1.s | sort | uniq -c
for x in unset 0 1; do
  for b in tight spare wpe1; do
    if [ $x = unset ]; then r=$(/tmp/k2_$b.bin $N 2>&1 | grep "bad runs"); else r=$(HSA_XNACK=$x /tmp/k2_$b.bin $N 2>&1 | grep "bad runs"); fi
    echo "HSA_XNACK=$x  $b: $r"
  done
done
echo "xnack as the runtime sees it:"; (rocminfo 2>/dev/null | grep -i -m2 xnack) || true
echo -n "rocgdb: "; (which rocgdb && rocgdb --version | head -1) || echo "not on this box"
