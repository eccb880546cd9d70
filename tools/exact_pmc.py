"""run the scoring stage of approx and exact mode a few dozen times (to be wrapped by rocprofv3 --pmc ...)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, first_index=0, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
thresh = float(sys.argv[1]) if len(sys.argv) > 1 else 0.99
for approx in (True, False):
    ms = voting.stage_repeat_ms(m, v, 1024, inlier_thresh=thresh, stage="score", repeats=30, approx=approx)
    print("approx" if approx else "exact", ms * 1e3, "us")
