cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
python -m pytest tests/test_evaluation.py tests/test_library_cpu.py -q 2>&1 | tail -8
python - <<'PY'
import torch, numpy as np, time
from pvnet_amd import synth, voting, evaluation as E
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
for r in (1, 10, 100, 100):
    print(r, voting.stage_repeat_ms(m, v, 1024, inlier_thresh=0.99, stage="score", repeats=r, both=True))
for pn in (5000, 20000, 50000):
    a = torch.rand((pn, 3), device=dev); b = torch.rand((pn, 3), device=dev)
    E.nearest_point_idx(a, b); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): E.nearest_point_idx(a, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"nearest_point_idx {pn} x {pn} (3-D): {dt*1e6:.1f} us  = {pn*pn/dt/1e12:.2f} T pairs/s")
PY
