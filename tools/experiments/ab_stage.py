import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
tag = os.environ.get("PVNET_VOTE_LIB", "new").split("/")[-1]
for st, name in (("mask_bits", "mask"), ("subsample", "subsample"), ("compact", "compact")):
    r = [voting.stage_repeat_ms(m, v, 1024, inlier_thresh=0.99, seed=1, stage=st, repeats=300) * 1e3 for _ in range(4)]
    print(tag, name, " ".join("%.2f" % x for x in r), "us")
# whole call, one stream
for i in range(20): voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i)
torch.cuda.synchronize()
import time
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(300): voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i)
    torch.cuda.synchronize()
    print(tag, "whole call one stream %.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
