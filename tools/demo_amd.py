"""tools/demo.py of the reference, minus what this image lacks (backbone weights, cv2, torchvision):
demo fixture -> ground-truth vector field (demo.py:58-71) -> HIP RANSAC voting (demo.py:55) -> host PnP (demo.py:179)
-> pose error against data/demo/cat_pose.npy.   python tools/demo_amd.py   (needs an MI355X)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvnet_amd import pnp, synth, voting  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "demo_cat.npz"))
h, w = (int(x) for x in g["shape"])
mask = np.unpackbits(g["mask_bits"])[: h * w].reshape(h, w)
planar = synth.field_from_keypoints(mask.astype(bool), g["points_2d"])
dev = torch.device("cuda:0")
m = torch.from_numpy(mask[None].astype(np.int64)).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar[None]).to(dev))
kpts, conf = voting.ransac_voting_layer_v5(m, v, 512, inlier_thresh=0.99, max_num=30000)
mean, cov = voting.estimate_voting_distribution_with_mean(m, v, kpts)
kp = kpts[0].cpu().numpy()
pose = pnp.pnp(g["points_3d"], kp, g["K"])
pose_u = pnp.uncertainty_pnp_v2(kp, cov[0].cpu().numpy(), g["points_3d"], g["K"])
target = g["pose"].astype(np.float64)
print("max |kpt - projected GT| px :", np.abs(kp - g["points_2d"]).max())
print("confidence                  :", conf[0].cpu().numpy().round(3))
print("pnp            (cm, deg, 2d-proj px):", *pnp.cm_degree_error(pose, target), pnp.projection_2d_error(pose, target, g["bb8_3d"], g["K"]))
print("uncertainty pnp(cm, deg, 2d-proj px):", *pnp.cm_degree_error(pose_u, target), pnp.projection_2d_error(pose_u, target, g["bb8_3d"], g["K"]))
