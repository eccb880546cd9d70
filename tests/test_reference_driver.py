"""The oracle and the HIP path against fixture G6 = outputs of the reference's OWN ransac_voting_layer_v3
(ransac_voting_gpu.py:514-598) executed on CPU by oracle/ref_driver.py (tests/golden/make_golden.py).

CPU tests: the numpy oracle, fed the idxs and the kept pixels the reference run drew, reproduces the reference's
key-points; GPU test: so does the HIP path through the C ABI."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ransac_voting_oracle as O
from oracle import ref_driver

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
MG = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MG)
FIX = np.load(os.path.join(HERE, "golden", "ref_driver_v3.npz"))

# The reference solves the refinement's 2x2 normal equations in float32 with un-centred sums (:586-594): its own
# output carries ~1e-4..2e-3 px of rounding noise depending on where the key-point lies (oracle).
TOL_REF_PX = 1e-3


def case_data(ci):
    case = MG.REF_DRIVER_CASES[ci]
    mask, vertex, _ = MG.ref_driver_inputs(case)
    b, h, w = mask.shape
    live = FIX[f"c{ci}_live"]
    keep = np.unpackbits(FIX[f"c{ci}_keep_bits"])[:b * h * w].reshape(b, h, w).astype(bool)
    idxs = np.zeros((b, case["hn"], 9, 2), np.int32)
    for j, bi in enumerate(live):
        idxs[bi] = FIX[f"c{ci}_idxs{j}"]
    return case, mask, vertex, keep, idxs, live, FIX[f"c{ci}_out"]


@pytest.mark.parametrize("ci", range(len(MG.REF_DRIVER_CASES)))
def test_oracle_reproduces_the_executed_reference_driver(ci):
    case, mask, vertex, keep, idxs, live, ref_out = case_data(ci)
    assert int(FIX["ncases"]) == len(MG.REF_DRIVER_CASES)
    kw = dict(case["kw"])
    thresh = kw.pop("inlier_thresh")
    for dtype, lsq in ((np.float32, np.float32), (np.float32, np.float64), (np.float64, np.float64)):
        out, dbg = O.ransac_voting_layer_v3(mask, vertex, case["hn"], inlier_thresh=thresh, idxs=idxs, keep=keep,
                                            dtype=dtype, lsq_dtype=lsq, return_debug=True, **kw)
        assert np.abs(out - ref_out).max() < TOL_REF_PX, (dtype, lsq)
    dead = [bi for bi in range(mask.shape[0]) if bi not in set(live.tolist())]
    assert all((ref_out[bi] == 0).all() for bi in dead)          # fewer than min_num pixels: zeros (:531-534)
    assert [d["tn"] for d in dbg if d.get("tn", 0) > 0] == FIX[f"c{ci}_tn"].tolist()
    if "max_num" in case["kw"]:                                  # thinned to about max_num (:537-540)
        tn0 = int((mask[0] != 0).sum())
        assert case["kw"]["max_num"] * 0.8 < FIX[f"c{ci}_tn"][0] < tn0


def test_reference_confidence_loop_is_degenerate():
    """confidence=1.0 can never be exceeded, so the reference's while-loop runs max_iter+1 rounds (:574-575) -- every
    one with the idxs drawn once before the loop (:547): the capture saw 4 identical calls, and the result equals the
    single-round oracle.  This is why round_hyp_num is the only hypothesis budget that matters."""
    ci = 3
    case, mask, vertex, keep, idxs, live, ref_out = case_data(ci)
    assert FIX[f"c{ci}_rounds"].tolist() == [case["kw"]["max_iter"] + 1]
    one = O.ransac_voting_layer_v3(mask, vertex, case["hn"], inlier_thresh=0.99, idxs=idxs)
    many = O.ransac_voting_layer_v3(mask, vertex, case["hn"], inlier_thresh=0.99, idxs=idxs, confidence=1.0, max_iter=3,
                                    emulate_rounds=True)
    assert np.array_equal(one, many) and np.abs(one - ref_out).max() < TOL_REF_PX


@pytest.mark.skipif(not ref_driver.available(), reason="needs the reference tree (build container only)")
def test_fixture_is_what_the_reference_driver_returns_today():
    case, mask, vertex, keep, idxs, live, ref_out = case_data(0)
    out, cap = ref_driver.run_v3(mask, vertex, case["hn"], torch_seed=100, **case["kw"])
    assert np.array_equal(out, ref_out) and all(np.array_equal(a, idxs[bi]) for a, bi in zip(cap.idxs, live))


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(MG.REF_DRIVER_CASES)))
@pytest.mark.parametrize("literal", [False, True])
def test_hip_path_reproduces_the_executed_reference_driver(ci, literal):
    import torch
    from pvnet_amd import voting
    case, mask, vertex, keep, idxs, live, ref_out = case_data(ci)
    kw = dict(case["kw"])
    kw.pop("max_num", None)   # the reference's own thinning decisions are applied to the mask instead of ours
    dev = torch.device("cuda:0")
    m = torch.from_numpy(mask * keep).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(vertex)).to(dev)
    out = voting.ransac_voting_layer_v3(m, v, case["hn"], idxs=torch.from_numpy(idxs).to(dev), literal=literal, **kw)
    assert np.abs(out.cpu().numpy() - ref_out).max() < TOL_REF_PX


# ------------------------------------------------------------------------------------------------ G7: siblings
SIB = np.load(os.path.join(HERE, "golden", "ref_driver_siblings.npz"))


def sibling_inputs():
    mask, vertex, kpts = MG.ref_driver_inputs(MG.SIBLING_CASE)
    b, h, w = mask.shape
    keep5 = np.unpackbits(SIB["v5_keep_bits"])[:b * h * w].reshape(b, h, w).astype(bool)
    return mask, vertex, kpts, keep5


def test_oracle_reproduces_reference_v5_distribution_motion_and_hypothesis_counts():
    mask, vertex, kpts, keep5 = sibling_inputs()
    b = mask.shape[0]
    # v5 (:763-858): thinned to ~max_num=100 pixels by the reference's own draw; points and confidence at 0.999
    assert (SIB["v5_tn"] < 160).all() and (SIB["v5_tn"] > 50).all()
    pts = O.ransac_voting_layer_v3(mask * keep5, vertex, 64, inlier_thresh=0.99, idxs=SIB["v5_idxs"], dtype=np.float32)
    assert np.abs(pts - SIB["v5_pts"]).max() < TOL_REF_PX
    conf = O.vote_confidence(mask * keep5, vertex, SIB["v5_pts"], 0.999, dtype=np.float32)
    assert np.abs(conf - SIB["v5_conf"]).max() < 1e-6
    # distribution about a given mean (:333-406): 4 rounds of 64 independent pairs = one draw of 256
    assert int(SIB["dist_rounds"]) == 4 and SIB["dist_idxs"].shape == (b, 256, 9, 2)
    cov = O.estimate_voting_distribution_with_mean(mask, vertex, SIB["dist_mean"], 256, inlier_thresh=0.99,
                                                   idxs=SIB["dist_idxs"], dtype=np.float32)
    assert np.abs(cov - SIB["dist_cov"]).max() < 1e-3 * max(1.0, np.abs(SIB["dist_cov"]).max())
    # motion voting (:960-981)
    assert np.abs(O.ransac_motion_voting(mask, vertex) - SIB["motion"]).max() < 1e-3
    # Python-level generate_hypothesis (:983-1034): every hypothesis and its inlier count
    _, dbg = O.ransac_voting_layer_v3(mask, vertex, 48, inlier_thresh=0.99, idxs=SIB["gh_idxs"], dtype=np.float32,
                                      return_debug=True)
    for bi in range(b):
        assert dbg[bi]["hyp"].astype(np.float32).tobytes() == SIB["gh_hyp"][bi].tobytes()
        np.testing.assert_array_equal(dbg[bi]["counts"], SIB["gh_counts"][bi])


# pixels of the ~100 kept ones whose vote may differ when the confidence is evaluated 1e-3 px away from the reference's point
CONF_EDGE_PIXELS = 2.0


@pytest.mark.gpu
def test_hip_siblings_reproduce_the_executed_reference():
    import torch
    from pvnet_amd import voting
    mask, vertex, kpts, keep5 = sibling_inputs()
    b = mask.shape[0]
    dev = torch.device("cuda:0")
    v = torch.from_numpy(np.ascontiguousarray(vertex)).to(dev)
    m = torch.from_numpy(mask).to(dev)
    m5 = torch.from_numpy(mask * keep5).to(dev)
    for literal in (False, True):
        pts, conf = voting.ransac_voting_layer_v5(m5, v, 64, inlier_thresh=0.99, max_num=30000, literal=literal,
                                                  idxs=torch.from_numpy(SIB["v5_idxs"]).to(dev))
        assert np.abs(pts.cpu().numpy() - SIB["v5_pts"]).max() < TOL_REF_PX
        # the confidence (:846-850) evaluated at the REFERENCE'S OWN refined points on this call's pixel list: the same integers
        # over the same tn -- equal, in both modes.  (At OUR refined point, <= 1e-3 px away, a pixel or two of the ~100 may sit
        # within that distance of the 0.999 cone's edge: that value is only held to the pixel count's granularity.)
        _, dbg5 = voting.ransac_voting_layer_v3(m5, v, 64, inlier_thresh=0.99, max_num=30000, literal=literal,
                                                idxs=torch.from_numpy(SIB["v5_idxs"]).to(dev), return_debug=True)
        conf_ref = voting.vote_confidence(dbg5, torch.from_numpy(SIB["v5_pts"]).to(dev), 0.999)
        assert np.abs(conf_ref.cpu().numpy() - SIB["v5_conf"]).max() <= 1e-6
        if np.array_equal(pts.cpu().numpy(), SIB["v5_pts"]):
            np.testing.assert_allclose(conf.cpu().numpy(), SIB["v5_conf"], rtol=0, atol=1e-6)
        else:
            assert np.abs(conf.cpu().numpy() - SIB["v5_conf"]).max() <= CONF_EDGE_PIXELS / SIB["v5_tn"].min() + 1e-6
        mean = torch.from_numpy(SIB["dist_mean"]).to(dev)
        _, cov = voting.estimate_voting_distribution_with_mean(m, v, mean, round_hyp_num=64, min_hyp_num=256,
                                                               inlier_thresh=0.99, literal=literal,
                                                               idxs=torch.from_numpy(SIB["dist_idxs"]).to(dev))
        # both modes hold the reference's integers (exact mode since ABI 6): the same tolerance, the same equality
        assert np.abs(cov.cpu().numpy() - SIB["dist_cov"]).max() < 1e-3 * max(1.0, np.abs(SIB["dist_cov"]).max())
        hyp, counts = voting.generate_hypothesis_counts(m, v, 48, inlier_thresh=0.99, literal=literal,
                                                        idxs=torch.from_numpy(SIB["gh_idxs"]).to(dev))
        assert hyp.cpu().numpy().tobytes() == SIB["gh_hyp"].tobytes()
        np.testing.assert_array_equal(counts.cpu().numpy(), SIB["gh_counts"])
    assert np.abs(voting.ransac_motion_voting(m, v).cpu().numpy() - SIB["motion"]).max() < 1e-3
