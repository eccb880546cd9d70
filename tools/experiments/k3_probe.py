import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["PVNET_VOTE_LIB"] = os.path.abspath("_ab/lib_k3probe.so"); os.environ["PVNET_SCORE_CULL"] = "1"
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
for i in range(4):
    voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i); torch.cuda.synchronize()
