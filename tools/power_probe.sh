# Development aid: sample power / clocks with rocm-smi while the path (or the scoring kernel alone) runs on 6 streams.
# bash tools/power_probe.sh   (on the GPU box)
cd $GRAFT_REPO_ROOT
sample() {  # $1 = label, runs while the background python lives
  for i in 1 2 3 4 5 6 7 8; do
    sleep 0.7
    echo "[$1] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk|fclk' | sed 's/GPU\[0\]\s*:\s*//' | tr '\n' ';')"
  done
}
echo "idle: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk' | sed 's/GPU\[0\]\s*:\s*//' | tr '\n' ';')"
PROBE_SHORT=1 python tools/overlap_probe.py 6 40000 > /tmp/op.txt 2>&1 &
PID=$!
sleep 6   # input generation + warm-up
sample "all stages"
wait $PID
cat /tmp/op.txt | grep -v amdgpu | head -3
