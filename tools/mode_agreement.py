"""How often does the default (fast) scoring pick a different winner than the literal float32 order, and which of
the two agrees with float64 arithmetic?   python tools/mode_agreement.py   (MI355X; oracle64 on the disagreements)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ransac_voting_oracle as O  # noqa: E402  (a tool, not the product)
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
B = 64
mask, planar, _ = synth.make_batch(B, radius=40, noise=True, background="normal")
vnp = synth.planar_to_vertex_view(planar)
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
fast, df = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=7, return_debug=True)
lit, dl = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=7, literal=True, return_debug=True)
wf, wl = df["win"][:, :, 0].cpu().numpy(), dl["win"][:, :, 0].cpu().numpy()
dc = (df["counts"] - dl["counts"]).abs()
diff = (fast - lit).norm(dim=-1).cpu().numpy()
same = wf == wl
print(f"{B} images x 9 key-points, 1024 hypotheses: winners equal in {same.sum()} of {same.size} "
      f"({100 * same.mean():.2f} %); counts differ on {int((dc > 0).sum())} of {dc.numel()} hypotheses (max {int(dc.max())})")
print(f"key-point distance fast vs literal: max {diff[same].max():.2e} px where winners agree, "
      f"{diff[~same].max() if (~same).any() else 0:.2e} px where they differ")
bad = np.argwhere(~same)
if len(bad):
    imgs = sorted(set(int(b) for b, _ in bad))
    o64, d64 = O.ransac_voting_layer_v3(mask[imgs], vnp[imgs], 1024, inlier_thresh=0.99, seed=7, image_offset=0,
                                        idxs=np.stack([O.draw_idxs(7, i, 1024, 9, int(df["tn"][i])) for i in imgs]),
                                        return_debug=True)
    agree_fast = agree_lit = 0
    for b, k in bad:
        w64 = d64[imgs.index(int(b))]["win_idx"][k]
        agree_fast += int(w64 == wf[b, k])
        agree_lit += int(w64 == wl[b, k])
    print(f"on the {len(bad)} disagreements the float64 oracle sides with fast {agree_fast}x, with literal {agree_lit}x")
