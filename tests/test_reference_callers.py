"""Row N1: the reference's OWN callers -- tools/demo.py:46-55 and tools/train_linemod.py:94-131 -- run
UNCHANGED on top of the HIP voting layer.

* CPU, build container (has /root/reference, no GPU): both scripts are imported byte for byte through the launcher's
  shims (tools/refshim.py); the five voting-layer names they import are checked to BE this repository's functions, and
  the way every wrapper's forward() calls them (function, arguments, tensor dtype / shape / strides) must equal the
  committed fixture G8 (tests/golden/reference_callers.json, made by the same probe).
* GPU: where the reference checkout exists next to a GPU the wrappers are executed for real on the demo fixture's
  ground-truth field; on the GPU box (no /root/reference) the recorded calls of G8 are replayed on the HIP layer --
  same functions, same arguments, tensors built by the very permute / view the wrappers apply -- and the key-points
  must be the fixture's analytic projections (G1), the pose the fixture's pose.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_callers.json")
PROBE = os.path.join(ROOT, "tools", "reference_callers_probe.py")
TOL_PX = 1e-3


def _probe(*extra):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    txt = subprocess.check_output([sys.executable, "-B", PROBE, REF, *extra], cwd="/tmp", env=env,
                                  stderr=subprocess.DEVNULL, timeout=600)
    return json.loads(txt)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tools")), reason="needs the reference checkout (build container)")
def test_reference_scripts_import_unchanged_and_call_the_hip_layer_as_recorded():
    got = _probe()
    want = json.load(open(FIXTURE))
    assert all(got["bound_to_hip_layer"].values()), got["bound_to_hip_layer"]  # `is` identity with pvnet_amd.voting.*
    assert got["calls"] == want["calls"]
    assert got["input"] == want["input"]
    # the call sites of the reference, spelled out (tools/demo.py:55, tools/train_linemod.py:104-106,117,129-130)
    c = want["calls"]
    assert c["demo.EvalWrapper"]["calls"][0]["args"] == [512] and \
        c["demo.EvalWrapper"]["calls"][0]["kwargs"] == {"inlier_thresh": 0.99}
    assert c["train.EvalWrapper"]["calls"][0]["kwargs"] == {"inlier_thresh": 0.99, "max_num": 100}
    assert c["train.EvalWrapper[use_uncertainty]"]["calls"][0]["function"] == "ransac_voting_layer_v5"
    assert [x["function"] for x in c["train.UncertaintyEvalWrapper"]["calls"]] == \
        ["ransac_voting_layer_v3", "estimate_voting_distribution_with_mean"]
    assert c["train.MotionEvalWrapper"]["calls"][0]["function"] == "ransac_motion_voting"
    mask_spec, vertex_spec = c["demo.EvalWrapper"]["calls"][0]["tensor_args"]
    assert mask_spec["dtype"] == "torch.int64" and not vertex_spec["contiguous"]  # argmax mask, permuted planar view


def test_probe_leaves_the_reference_tree_clean():
    """importing from the reference checkout must never write into it (no __pycache__)"""
    if not os.path.isdir(REF):
        pytest.skip("no reference checkout")
    import time
    t0 = time.time()
    _probe()
    fresh = []
    for d, _, fs in os.walk(REF):
        fresh += [os.path.join(d, f) for f in fs if f.endswith(".pyc") and os.path.getmtime(os.path.join(d, f)) >= t0 - 1]
    assert not fresh, fresh


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "utils", "extend_utils")), reason="needs the reference checkout")
def test_reference_extend_utils_and_evaluation_utils_run_unchanged_on_the_native_libraries():
    """the downstream callers: the reference's own lib/utils/extend_utils/extend_utils.py (cffi front end) and
    lib/utils/evaluation_utils.py, byte for byte, on this repository's stand-in for the cffi-built `_extend_utils`
    (lib/utils/extend_utils/_extend_utils.py: uncertainty_pnp / farthest_point_sampling in libpvnet_pnp.so) -- key-point
    selection as data_utils.py:144 does it, `pnp`, `uncertainty_pnp`, `uncertainty_pnp_v2`."""
    from oracle import fps_oracle
    from pvnet_amd import pnp as P
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    txt = subprocess.check_output([sys.executable, "-B", os.path.join(ROOT, "tools", "reference_extend_utils_probe.py"), REF],
                                  cwd="/tmp", env=env, stderr=subprocess.DEVNULL, timeout=600)
    got = json.loads(txt)
    assert got["extend_utils"].startswith(REF) and got["stand_in"].startswith(ROOT)
    rng = np.random.default_rng(5)                                   # the probe's inputs, regenerated
    model = rng.normal(size=(2000, 3)).astype(np.float32) * np.array([0.05, 0.03, 0.08], np.float32)
    idx = fps_oracle.farthest_point_sampling_init_center(model, 8)
    np.testing.assert_array_equal(np.asarray(got["fps_points"], np.float32), model[idx])
    X3, x2 = np.asarray(got["inputs"]["X3"]), np.asarray(got["inputs"]["x2"])
    W, cov = np.asarray(got["inputs"]["W"]), np.asarray(got["inputs"]["cov"])
    K, true = P.LINEMOD_K, np.asarray(got["pose_true"])
    for name, mine in (("pnp", P.pnp(X3, x2, K)), ("uncertainty_pnp", P.uncertainty_pnp(x2, W, X3, K)),
                       ("uncertainty_pnp_v2", P.uncertainty_pnp_v2(x2, cov, X3, K))):
        theirs = np.asarray(got[name])
        assert np.abs(theirs - mine).max() < 1e-6, name             # same minimum of the same cost, whatever the start
        tr, rot = P.cm_degree_error(theirs, true)
        assert tr < 3.0 and rot < 3.0, (name, tr, rot)              # 0.4 px of key-point noise at 0.9 m


def test_farthest_point_sampling_equals_oracle_and_the_references_own_source():
    """libpvnet_pnp.so `farthest_point_sampling_init_center` (the reference's C symbol, include/pvnet_pnp.h) == numpy float32
    restatement == the reference's own farthest_point_sampling.cpp compiled where it lies (oracle/_ref, when built)."""
    import ctypes as C
    from oracle import fps_oracle
    import importlib
    stand_in = importlib.import_module("lib.utils.extend_utils._extend_utils")
    rng = np.random.default_rng(11)
    for pn, sn in ((500, 8), (64, 64), (3000, 33), (10, 4)):
        pts = rng.normal(size=(pn, 3)).astype(np.float32)
        pts[pn // 2] = pts[0]                                       # a duplicate point: ties and zero distances
        idxs = np.zeros(sn, np.int32)
        stand_in.lib.farthest_point_sampling_init_center(stand_in.ffi.cast("float*", pts.ctypes.data),
                                                         stand_in.ffi.cast("int*", idxs.ctypes.data), pn, sn)
        np.testing.assert_array_equal(idxs, fps_oracle.farthest_point_sampling_init_center(pts, sn))
        if fps_oracle.reference_available():
            np.testing.assert_array_equal(idxs, fps_oracle.reference_init_center(pts, sn))
        if sn < pn // 2:
            assert len(set(idxs.tolist())) == sn
    # the random-start variant: sn distinct indices, spread out (every selected pair farther apart than the cloud's median)
    pts = rng.normal(size=(400, 3)).astype(np.float32)
    idxs = np.zeros(6, np.int32)
    stand_in.lib.farthest_point_sampling(stand_in.ffi.cast("float*", pts.ctypes.data), stand_in.ffi.cast("int*", idxs.ctypes.data), 400, 6)
    assert len(set(idxs.tolist())) == 6 and (idxs >= 0).all() and (idxs < 400).all()
    sel = pts[idxs]
    d = np.linalg.norm(sel[:, None] - sel[None], axis=-1)[np.triu_indices(6, 1)]
    assert d.min() > np.median(np.linalg.norm(pts[:, None] - pts[None], axis=-1))
    with pytest.raises(NotImplementedError):
        stand_in.lib.mesh_binary_rasterization(None, None, 0, 0, 0)


# ------------------------------------------------------------------------------------------------------------ GPU
def _demo_inputs(demo_fixture, dev):
    from pvnet_amd import synth
    mask = demo_fixture["mask"]
    planar = synth.field_from_keypoints(mask.astype(bool), demo_fixture["points_2d"])
    m = torch.from_numpy(mask.astype(np.int64)).to(dev)
    seg_pred = torch.stack([1.0 - m.float(), m.float()])[None].contiguous()  # logits whose arg-max is the mask
    vertex_pred = torch.from_numpy(planar[None]).to(dev)                       # [1, 2vn, h, w] as the backbone emits it
    return seg_pred, vertex_pred


def _as_the_wrappers_do(seg_pred, vertex_pred):
    """the tensor preparation every EvalWrapper.forward of the reference performs (tools/demo.py:48-52)"""
    vertex_pred = vertex_pred.permute(0, 2, 3, 1)
    b, h, w, vn_2 = vertex_pred.shape
    return torch.argmax(seg_pred, 1), vertex_pred.view(b, h, w, vn_2 // 2, 2)


@pytest.mark.gpu
def test_reference_wrappers_on_the_hip_layer(demo_fixture):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    from pvnet_amd import pnp as P
    from pvnet_amd import voting
    pts = demo_fixture["points_2d"]
    if os.path.isdir(os.path.join(REF, "tools")):  # a GPU next to the reference checkout: the real thing
        r = _probe("--device", "cuda")
        outs = {k: [np.asarray(o) for o in v["outputs"]] for k, v in r["calls"].items()}
    else:  # GPU box: replay what the unchanged scripts were recorded to do (fixture G8)
        spec = json.load(open(FIXTURE))["calls"]
        seg_pred, vertex_pred = _demo_inputs(demo_fixture, dev)
        mask, vertex = _as_the_wrappers_do(seg_pred, vertex_pred)
        b, h, w, vn, _ = vertex.shape
        outs = {}
        for label, rec in spec.items():
            torch.manual_seed(7)
            result = None
            for call in rec["calls"]:
                m_spec, v_spec = call["tensor_args"][:2]
                # the tensors handed over are of the recorded kind: int64 arg-max mask, strided [b,h,w,vn,2] view
                assert str(mask.dtype) == m_spec["dtype"] and mask.is_contiguous() == m_spec["contiguous"]
                assert str(vertex.dtype) == v_spec["dtype"] and vertex.is_contiguous() == v_spec["contiguous"]
                assert tuple(vertex.stride())[1:] == (w, 1, 2 * h * w, h * w)  # (dim 0 has size 1 here: any stride)
                assert np.argsort(v_spec["stride"][1:]).tolist() == np.argsort(list(vertex.stride())[1:]).tolist()
                fn = getattr(voting, call["function"])
                args = [mask, vertex]
                if call["function"] == "estimate_voting_distribution_with_mean":
                    args.append(result)  # the mean of the preceding v3 call (train_linemod.py:129-130)
                result = fn(*args, *call["args"], **call["kwargs"])
            r = result if isinstance(result, tuple) else (result,)
            outs[label] = [x.detach().cpu().numpy() for x in r]
    # demo.py:55 / train_linemod.py:106: key-points of the ground-truth field = the analytic projections
    for label in ("demo.EvalWrapper", "train.EvalWrapper"):
        kp = outs[label][0]
        assert kp.shape == (1, 9, 2) and np.abs(kp[0] - pts).max() < (TOL_PX if label == "demo.EvalWrapper" else 5e-2)
    # train_linemod.py:104: v5 = key-points + per-key-point confidence (clean field: every kept pixel agrees)
    kp5, conf = outs["train.EvalWrapper[use_uncertainty]"]
    assert kp5.shape == (1, 9, 2) and conf.shape == (1, 9) and np.abs(kp5[0] - pts).max() < 5e-2 and conf.min() > 0.9
    # train_linemod.py:117: mean over the foreground of (vertex + pixel): a unit-vector field gives points near the object
    mo = outs["train.MotionEvalWrapper"][0]
    assert mo.shape == (1, 9, 2) and np.isfinite(mo).all()
    # train_linemod.py:129-130: v3 mean + hypothesis covariance (clean field: hypotheses collapse onto the key-point)
    mean, var = outs["train.UncertaintyEvalWrapper"]
    assert mean.shape == (1, 9, 2) and var.shape == (1, 9, 2, 2) and np.abs(mean[0] - pts).max() < TOL_PX
    assert np.isfinite(var).all() and np.abs(var).max() < 1.0
    # and the pose the demo goes on to compute from these key-points (demo.py:166-171) is the fixture's
    pose = P.pnp(demo_fixture["points_3d"], outs["demo.EvalWrapper"][0][0].astype(np.float64), demo_fixture["K"])
    tr_cm, rot_deg = P.cm_degree_error(pose, demo_fixture["pose"].astype(np.float64))
    assert tr_cm < 0.05 and rot_deg < 0.1


@pytest.mark.gpu
def test_nearest_neighbour_through_the_cffi_stand_in():
    """extend_utils.py:39-60 as the reference writes it -- host arrays, `ffi.cast`, `lib.findNearestPointIdxLauncher` -- on the
    stand-in for `_extend_utils`: the HIP brute-force search behind the reference's own launcher symbol."""
    import importlib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    m = importlib.import_module("lib.utils.extend_utils._extend_utils")
    lib, ffi = m.lib, m.ffi
    rng = np.random.default_rng(3)
    for dim in (3, 2):
        ref_pts = np.ascontiguousarray(rng.normal(size=(1, 1500, dim)), np.float32)
        que_pts = np.ascontiguousarray(rng.normal(size=(1, 1200, dim)), np.float32)
        idxs = np.zeros([1, 1200], np.int32)
        lib.findNearestPointIdxLauncher(ffi.cast('float *', ref_pts.ctypes.data), ffi.cast('float *', que_pts.ctypes.data),
                                        ffi.cast('int *', idxs.ctypes.data), 1, 1500, 1200, dim, 0)
        d = ((que_pts[0][:, None, :] - ref_pts[0][None, :, :]) ** 2).sum(-1)
        np.testing.assert_array_equal(idxs[0], d.argmin(1))
