"""Development aid (GPU): fast mode vs literal mode over many hypothesis counts (every MFMA tile count, padded
slices) -- counts within 2, identical key-points wherever the winners agree.   python tools/hn_sweep.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
bad = 0
for hn in [1, 2, 7, 31, 32, 33, 64, 100, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1025, 2000, 3000]:
    for b, vn, r in [(1, 9, 20), (3, 2, 33), (5, 9, 12)]:
        mask, planar, kp = synth.make_batch(b, first_index=hn, h=150, w=200, vn=vn, radius=r, noise=True, background="normal")
        m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
        of, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=3, return_debug=True)
        cf, wf = df["counts"].clone(), df["win"].clone()
        ol, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=3, literal=True, return_debug=True)
        dc = (cf - dl["counts"]).abs().max().item()
        same = (wf[:, :, 0] == dl["win"][:, :, 0])
        d = (of - ol).norm(dim=-1)
        if dc > 2 or (d[same].max().item() if same.any() else 0) > 1e-3:
            bad += 1
            print("BAD", hn, b, vn, r, dc, d.max().item(), same.float().mean().item())
print("hn sweep done, bad =", bad)
