"""Development aid (round 5): what does the thinning histogram cost the mask kernel?  With max_num >= h*w thinning is impossible and K1
skips it (P.cum == NULL): same inputs, stage times with the default max_num = 30 000 against max_num = h*w."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
sets = []
for s in range(4):   # four input sets: the masks do not sit in the Infinity Cache
    mask, planar, _ = synth.make_batch(32, first_index=32 * s, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
for max_num in (30000, 480 * 640, 30000, 480 * 640):
    ts = []
    for i in range(24):
        m, v = sets[i % 4]
        _, t = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i, max_num=max_num, stage_times=True, concurrent=False)
        if i >= 4: ts.append(t)
    med = {k: round(float(np.median([x[k] for x in ts])) * 1e3, 1) for k in ts[0]}
    print("max_num", max_num, med, flush=True)
