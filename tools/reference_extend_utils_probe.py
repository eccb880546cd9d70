"""Runs the reference's OWN lib/utils/extend_utils/extend_utils.py and lib/utils/evaluation_utils.py UNCHANGED on this
repository's native libraries (the `_extend_utils` stand-in in the overlay tree; cv2 through tools/refshim.py) and prints
what they return for seeded synthetic inputs -- the downstream half of "the reference's callers run unchanged".

    python -B tools/reference_extend_utils_probe.py /path/to/pvnet [--gpu]        -> one JSON document on stdout
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refshim  # noqa: E402


def main(argv):
    ref = os.path.abspath(argv[0])
    gpu = "--gpu" in argv
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    refshim.install(ref)
    refshim.pin_overlay(ref)
    os.chdir(ref)  # lib/utils/config.py opens configs/ relatively
    import importlib
    import numpy as np
    eu = importlib.import_module("lib.utils.extend_utils.extend_utils")       # the reference's file
    stand_in = importlib.import_module("lib.utils.extend_utils._extend_utils")  # this repository's
    assert os.path.abspath(eu.__file__).startswith(ref + os.sep), eu.__file__
    assert os.path.abspath(stand_in.__file__).startswith(ROOT + os.sep), stand_in.__file__
    ev = importlib.import_module("lib.utils.evaluation_utils")
    assert os.path.abspath(ev.__file__).startswith(ref + os.sep), ev.__file__
    from pvnet_amd import pnp as P
    rng = np.random.default_rng(5)
    out = {"extend_utils": eu.__file__, "stand_in": stand_in.__file__}
    # farthest-point sampling as lib/utils/data_utils.py:144 calls it
    model = rng.normal(size=(2000, 3)).astype(np.float32) * np.array([0.05, 0.03, 0.08], np.float32)
    out["fps_points"] = eu.farthest_point_sampling(model, 8, True).tolist()
    # poses
    X3 = rng.uniform(-0.08, 0.08, size=(9, 3))
    K = P.LINEMOD_K
    aa = np.array([0.4, -0.7, 0.2])
    pose = np.concatenate([P.rodrigues(aa), np.array([[0.02], [-0.03], [0.9]])], 1)
    x2 = P.project(X3, pose, K) + rng.normal(scale=0.4, size=(9, 2))
    out["pose_true"] = pose.tolist()
    out["pnp"] = ev.pnp(X3, x2, K).tolist()                                     # evaluation_utils.py:19-52
    sig = rng.uniform(0.3, 2.0, size=9)
    W = np.stack([1 / sig, np.zeros(9), 1 / sig], 1)                            # (wxx, wxy, wyy)
    out["uncertainty_pnp"] = eu.uncertainty_pnp(x2, W, X3, K).tolist()          # extend_utils.py:63-114
    cov = np.stack([np.diag([s * s, s * s]) for s in sig])
    out["uncertainty_pnp_v2"] = eu.uncertainty_pnp_v2(x2, cov, X3, K).tolist()  # :116-165
    out["inputs"] = {"X3": X3.tolist(), "x2": x2.tolist(), "W": W.tolist(), "cov": cov.tolist(), "model_seed": 5}
    if gpu:
        ref_pts = rng.normal(size=(3000, 3)).astype(np.float32)
        que = rng.normal(size=(2500, 3)).astype(np.float32)
        out["nn_idx"] = eu.find_nearest_point_idx(ref_pts, que).tolist()        # :39-60
        out["nn_dist_mean"] = float(ev.find_nearest_point_distance(ref_pts, que).mean())
    os.write(real_stdout, json.dumps(out).encode())
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
