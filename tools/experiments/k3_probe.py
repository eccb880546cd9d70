"""Development aid (round 6): shader-clock stamps of the phases of a culled key-point's K3 block (cull_block, k3_hypotheses.hip) --
needs a library built with -DPVNET_K3_PROBE (PVNET_VOTE_LIB points at it); every key-point culled (PVNET_F_CULL_ALL).
    python tools/experiments/variant_lib.py _ab/lib_k3probe.so k3_hypotheses.hip:-DPVNET_K3_PROBE ; python tools/experiments/k3_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
os.environ.setdefault("PVNET_VOTE_LIB", os.path.abspath("_ab/lib_k3probe.so"))
from pvnet_amd import synth, voting  # noqa: E402

voting.set_cull_selection("all")

dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
names = ("preamble", "hypotheses", "keys", "sort", "outputs")
for rep in range(3):
    _, d, st = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=rep, return_debug=True, stage_times=True)
    torch.cuda.synchronize()
    L, ws = d["layout"], d["workspace"]
    max_items = L.b * L.vn * (L.hgroups // L.wg_g) * ((L.max_chunks + L.wg_s - 1) // L.wg_s)
    items = ws[L.off_items:L.off_items + 16 * max_items].view(torch.int32).view(max_items, 4).cpu().numpy()
    nbk = L.b * L.vn
    st_ = np.stack([items[max_items - 2 - 2 * bk:max_items - 2 * bk].reshape(-1)[:5] for bk in range(nbk)]).astype(np.float64)
    dl = np.diff(np.concatenate([np.zeros((nbk, 1)), st_], 1), axis=1)
    print(f"call {rep}: hypotheses stage {st['hypotheses'] * 1e3:.1f} us; cull blocks, shader-clock cycles per phase (median / max over "
          f"{nbk} blocks): " + "  ".join(f"{n} {np.median(dl[:, i]):.0f}/{dl[:, i].max():.0f}" for i, n in enumerate(names)) +
          f"  | total {np.median(st_[:, 4]):.0f}/{st_[:, 4].max():.0f}")
