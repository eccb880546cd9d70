"""Regenerates the golden fixtures under tests/golden/.  Run from the repo root IN THE BUILD CONTAINER
(needs /root/reference for the demo fixture):  python tests/golden/make_golden.py

G1  demo_cat.npz   -- the reference's demo fixture (data/demo/): foreground mask (bit-packed), the nine 3-D
                      key-points, the pose and the LINEMOD intrinsics, plus the analytically projected 2-D
                      key-points = the known answer of voting on the ground-truth field (tools/demo.py:74-103,
                      lib/utils/base_utils.py:239-256).  The projection is restated here (3 lines of numpy)
                      because importing lib/utils/base_utils.py needs cv2/plyfile, absent from this image.
G3  noisy_oracle.npz -- float64-oracle outputs (key-points, winner indices, winner counts) on seeded noisy
                      synthetic images with the counter-based RNG; guards the oracle itself against drift.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def make_demo():
    from PIL import Image
    d = os.path.join(REF, "data", "demo")
    mask = np.array(Image.open(os.path.join(d, "cat_mask.png"))).astype(np.int32)[..., 0]  # demo.py:79
    mask[mask != 0] = 1  # demo.py:80
    pts3d = np.loadtxt(os.path.join(d, "cat_points_3d.txt"))
    bb8 = np.loadtxt(os.path.join(d, "cat_bb8_3d.txt"))
    pose = np.load(os.path.join(d, "cat_pose.npy"))
    K = np.array([[572.4114, 0., 325.2611], [0., 573.57043, 242.04899], [0., 0., 1.]])  # base_utils.py:241-243
    p = pts3d @ pose[:, :3].T + pose[:, 3:].T  # base_utils.py:253
    p = p @ K.T  # :254
    pts2d = p[:, :2] / p[:, 2:]  # :255
    np.savez_compressed(os.path.join(OUT, "demo_cat.npz"), mask_bits=np.packbits(mask.astype(np.uint8)),
                        shape=np.array(mask.shape), points_3d=pts3d, bb8_3d=bb8, pose=pose, K=K, points_2d=pts2d)
    print("demo_cat: tn =", int(mask.sum()), "\n", pts2d)


def make_noisy():
    from oracle import ransac_voting_oracle as O
    from pvnet_amd import synth
    cases = []
    for idx, (radius, hn, thresh) in enumerate([(20, 128, 0.99), (30, 256, 0.99), (25, 128, 0.999)]):
        mask, planar, kpts = synth.make_batch(2, first_index=100 + 10 * idx, h=240, w=320, radius=radius,
                                              background="normal", noise=True)
        vertex = synth.planar_to_vertex_view(planar)
        out, dbg = O.ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=thresh, seed=1234 + idx,
                                            return_debug=True)
        cases.append(dict(radius=radius, hn=hn, thresh=thresh, seed=1234 + idx, first_index=100 + 10 * idx,
                          out=out, win_idx=np.stack([d["win_idx"] for d in dbg]),
                          win_cnt=np.stack([d["win_cnt"] for d in dbg]), tn=np.array([d["tn"] for d in dbg])))
        print("noisy case", idx, "tn", cases[-1]["tn"], "max|out-kpt|", np.abs(out - kpts).max())
    flat = {}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat[f"c{i}_{k}"] = np.asarray(v)
    flat["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "noisy_oracle.npz"), **flat)


if __name__ == "__main__":
    if os.path.isdir(REF):
        make_demo()
    else:
        print("no /root/reference: demo fixture not regenerated")
    make_noisy()
