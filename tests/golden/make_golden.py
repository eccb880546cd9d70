"""Regenerates the golden fixtures under tests/golden/.  Run from the repo root IN THE BUILD CONTAINER
(needs /root/reference for the demo fixture):  python tests/golden/make_golden.py

G1  demo_cat.npz   -- the reference's demo fixture (data/demo/): foreground mask (bit-packed), the nine 3-D
                      key-points, the pose and the LINEMOD intrinsics, plus the analytically projected 2-D
                      key-points = the known answer of voting on the ground-truth field (tools/demo.py:74-103,
                      lib/utils/base_utils.py:239-256).  The projection is restated here (3 lines of numpy)
                      because importing lib/utils/base_utils.py needs cv2/plyfile, absent from this image.
G6  ref_driver_v3.npz -- outputs of the REFERENCE'S OWN ransac_voting_layer_v3 (ransac_voting_gpu.py:514-598), executed
                      here on CPU tensors by oracle/ref_driver.py (the extension's two kernels stubbed by the C
                      restatement, which tests/test_reference_kernels.py holds bit-equal to the reference's device code),
                      together with what the run drew: the idxs of every live image, the pixels it kept, the rounds its
                      confidence loop made.  Inputs are regenerated from the recorded synth parameters.
G7  ref_driver_siblings.npz -- the same for the reference's sibling functions of the "next" rows of SURVEY section 8(f):
                      ransac_voting_layer_v5 (:763-858), estimate_voting_distribution_with_mean (:333-406),
                      ransac_motion_voting (:960-981) and the Python-level generate_hypothesis (:983-1034).
G8  reference_callers.json -- how the reference's OWN tools/demo.py and tools/train_linemod.py call the voting layer:
                      both scripts imported UNCHANGED through tools/refshim.py (their five voting-layer imports bind this
                      repository's HIP functions), every EvalWrapper's forward() run with recorders in place of the layer
                      (tools/reference_callers_probe.py): function called, arguments, dtype / shape / strides of the tensors.
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


REF_DRIVER_CASES = [
    # synth parameters, call parameters (those of ransac_voting_layer_v3), edits applied to the mask
    dict(b=2, first_index=900, h=120, w=160, radius=15, noise=True, hn=64, kw=dict(inlier_thresh=0.99), edit="none"),
    dict(b=3, first_index=910, h=96, w=128, radius=12, noise=True, hn=128, kw=dict(inlier_thresh=0.999),
         edit="image1_three_pixels"),                       # fewer than min_num=5 -> zeros (:531-534)
    dict(b=1, first_index=920, h=120, w=160, radius=22, noise=True, hn=64, kw=dict(inlier_thresh=0.99, max_num=400),
         edit="none"),                                      # tn0 > max_num -> uniform_ < max_num/tn0 thinning (:537-540)
    dict(b=1, first_index=930, h=96, w=128, radius=12, noise=True, hn=32,
         kw=dict(inlier_thresh=0.99, confidence=1.0, max_iter=3), edit="none"),  # the loop runs max_iter+1 rounds
    dict(b=1, first_index=940, h=96, w=128, radius=14, noise=False, hn=32, kw=dict(inlier_thresh=0.99), edit="none"),
]


def ref_driver_inputs(case):
    """mask [b,h,w] int64, vertex view [b,h,w,vn,2] of one G6 case (shared with the tests)."""
    from pvnet_amd import synth
    mask, planar, kpts = synth.make_batch(case["b"], first_index=case["first_index"], h=case["h"], w=case["w"],
                                          radius=case["radius"], background="normal", noise=case["noise"])
    if case["edit"] == "image1_three_pixels":
        mask[1] = 0
        mask[1, 7, 10:13] = 1
    return mask, synth.planar_to_vertex_view(planar), kpts


def make_ref_driver():
    from oracle import ref_driver
    flat = {"ncases": np.array(len(REF_DRIVER_CASES))}
    for ci, case in enumerate(REF_DRIVER_CASES):
        mask, vertex, kpts = ref_driver_inputs(case)
        out, cap = ref_driver.run_v3(mask, vertex, case["hn"], torch_seed=100 + ci, **case["kw"])
        b, h, w = mask.shape
        min_num = case["kw"].get("min_num", 5)
        live = [bi for bi in range(b) if int((mask[bi].astype(np.uint8) != 0).sum()) >= min_num]
        assert len(live) == len(cap.idxs)
        keep = np.zeros((b, h, w), bool)
        for j, bi in enumerate(live):
            xy = cap.coords[j].astype(np.int64)
            keep[bi, xy[:, 1], xy[:, 0]] = True
        flat[f"c{ci}_out"] = out
        flat[f"c{ci}_live"] = np.array(live)
        flat[f"c{ci}_tn"] = np.array(cap.tn)
        flat[f"c{ci}_rounds"] = np.array(cap.rounds)
        flat[f"c{ci}_keep_bits"] = np.packbits(keep)
        for j in range(len(live)):
            flat[f"c{ci}_idxs{j}"] = cap.idxs[j].astype(np.int32)
        print(f"ref driver case {ci}: live {live} tn {cap.tn} rounds {cap.rounds} max|out-kpt| "
              f"{np.abs(out[live] - kpts[live]).max():.3f} px")
    np.savez_compressed(os.path.join(OUT, "ref_driver_v3.npz"), **flat)


SIBLING_CASE = dict(b=2, first_index=950, h=96, w=128, radius=13, noise=True, edit="none")


def keep_from_capture(cap, live, shape):
    keep = np.zeros(shape, bool)
    for j, bi in enumerate(live):
        xy = cap.coords[j].astype(np.int64)
        keep[bi, xy[:, 1], xy[:, 0]] = True
    return keep


def make_ref_siblings():
    from oracle import ref_driver
    mask, vertex, kpts = ref_driver_inputs(SIBLING_CASE)
    b = mask.shape[0]
    flat = {}
    # v5: default max_num=100 -> every image is thinned; returns (points, confidence at 0.999)
    (pts, conf), cap = ref_driver.run("ransac_voting_layer_v5", mask, vertex, 64, torch_seed=7, inlier_thresh=0.99)
    flat.update(v5_pts=pts, v5_conf=conf, v5_keep_bits=np.packbits(keep_from_capture(cap, range(b), mask.shape)),
                v5_idxs=np.stack(cap.idxs).astype(np.int32), v5_tn=np.array(cap.tn))
    # distribution about the true key-points: 4 rounds of 64 hypotheses
    mean = kpts.astype(np.float32)
    (mean_out, cov), cap = ref_driver.run("estimate_voting_distribution_with_mean", mask, vertex, mean, torch_seed=8,
                                          round_hyp_num=64, min_hyp_num=256, inlier_thresh=0.99)
    rounds = len(cap.idxs) // b
    idxs = np.stack([np.concatenate(cap.idxs[bi * rounds:(bi + 1) * rounds], 0) for bi in range(b)]).astype(np.int32)
    flat.update(dist_mean=mean, dist_cov=cov, dist_idxs=idxs, dist_rounds=np.array(rounds))
    # motion voting: no extension involved at all
    (mv,), _ = ref_driver.run("ransac_motion_voting", mask, vertex)
    flat.update(motion=mv)
    # Python-level generate_hypothesis: all hypotheses and their inlier counts
    (hyp, counts), cap = ref_driver.run("generate_hypothesis", mask, vertex, 48, torch_seed=9, inlier_thresh=0.99)
    flat.update(gh_hyp=hyp, gh_counts=counts, gh_idxs=np.stack(cap.idxs).astype(np.int32))
    np.savez_compressed(os.path.join(OUT, "ref_driver_siblings.npz"), **flat)
    print("ref siblings: v5 tn", cap.tn, "conf", conf.round(3).tolist(), "| dist rounds", rounds,
          "| cov[0,0]", cov[0, 0].round(4).tolist())


def make_reference_callers():
    import json
    import subprocess
    txt = subprocess.check_output([sys.executable, "-B", os.path.join(ROOT, "tools", "reference_callers_probe.py"), REF],
                                  cwd="/tmp", stderr=subprocess.DEVNULL)
    d = json.loads(txt)
    d.pop("reference_root")
    with open(os.path.join(OUT, "reference_callers.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote reference_callers.json:", list(d["calls"]))


if __name__ == "__main__":
    if os.path.isdir(REF):
        make_reference_callers()
        make_demo()
        make_ref_driver()
        make_ref_siblings()
    else:
        print("no /root/reference: demo and reference-driver fixtures not regenerated")
    make_noisy()
