#!/bin/bash
# round 5, GPU call 6: deferred re-evaluation, second form (cells decoded into LDS, four lanes per (cell, tile) unit)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 2>&1 | tail -40 ) > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt | cut -c1-500
python - <<'PY' 2>&1 | tee gpurun_out/r05f/stage_times.txt
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
for lib in ("_ab/lib_r05_before_dr.so", "pvnet_amd/libpvnet_vote.so"):
    pass
for fold in ("-1", "0"):
    for thr in (0.99, 0.999):
        os.environ["PVNET_EXACT_FOLD"] = fold; voting.reload_tuning()
        ts = []
        for i in range(8):
            _, t = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thr, seed=i, stage_times=True, concurrent=False)
            ts.append(t)
        med = {k: round(float(np.median([x[k] for x in ts])) * 1e3, 1) for k in ts[0]}
        print("PVNET_EXACT_FOLD", fold, "thresh", thr, med, "sum", round(sum(med.values()), 1))
PY
python tools/ab.py --rounds 2 before=_ab/lib_r05_before_dr.so after=pvnet_amd/libpvnet_vote.so after_item_cells=pvnet_amd/libpvnet_vote.so,PVNET_EXACT_FOLD=0 --no-secondary --steps 300 --warmup 30 2>&1 | tee $O/ab_dr.txt
