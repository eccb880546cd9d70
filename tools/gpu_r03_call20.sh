# energy accounting of the exact and the approximate mode (tools/energy_probe.py), six streams and one
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c20
mkdir -p $O
PROBE_SECONDS=4 timeout 300 python tools/energy_probe.py 6 > $O/energy_exact.txt 2>&1
PROBE_SECONDS=4 timeout 300 python tools/energy_probe.py 6 --approx > $O/energy_approx.txt 2>&1
cat $O/energy_exact.txt $O/energy_approx.txt
