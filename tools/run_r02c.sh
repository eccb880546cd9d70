cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
./tools/ubench_wallclock.bin 2>&1 | tee gpurun_out/r02c/wallclock.txt
python -m pytest tests/test_fast_mode_parity.py tests/test_reference_callers.py -m gpu -q 2>&1 | tail -5
python - <<'PY'
import torch, numpy as np
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
for r in (1, 10, 100):
    print(r, voting.stage_repeat_ms(m, v, 1024, inlier_thresh=0.99, stage="score", repeats=r, both=True))
PY
