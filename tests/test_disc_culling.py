"""GPU tests of the exact mode's DISC CULLING (rounds 5-6; k3_hypotheses.hip: cull_block of hypothesis_kernel; k4_score_cull.hip:
score_exact_kernel_both.  PVNET_F_CULL_ALL / PVNET_SCORE_CULL=1 (development build) cull every key-point; by default K3 selects per
image from the spread of the band-origin candidates, gated per batch by the previous call on the workspace): a culled key-point's hypotheses are sorted along a Hilbert curve, every tile of 32 is described by a disc,
and a pixel whose margin at the disc's centre exceeds the disc's radius (+ the rounding band) votes for all 32 hypotheses of the
tile or for none -- only the other pixels are gathered into MFMA tiles.  The claim under test: EVERY inlier count, every winner
and every key-point is the one the full exact kernel (and therefore literal mode, i.e. the reference's own arithmetic:
ransac_voting_kernel.cu:88-126) returns -- `torch.equal`, no tolerance -- while the work really shrinks."""
import os

import numpy as np
import pytest
import torch

from oracle import refkernels
from pvnet_amd import synth, voting

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def batch(n, first, h, w, radius, noise=True, background="normal", **kw):
    mask, planar, kpts = synth.make_batch(n, first_index=first, h=h, w=w, radius=radius, noise=noise, background=background, **kw)
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    return m, v, kpts


@pytest.fixture
def cull(monkeypatch):
    """every key-point culled (True) / none (False) / the library's own selection ("auto") for the calls of one test; the library
    default (auto) comes back afterwards"""
    def on(flag=True):
        if flag == "auto":
            monkeypatch.delenv("PVNET_SCORE_CULL", raising=False)
        else:
            monkeypatch.setenv("PVNET_SCORE_CULL", "1" if flag else "0")
        voting.reload_tuning()
    yield on
    monkeypatch.delenv("PVNET_SCORE_CULL", raising=False)
    voting.reload_tuning()


def snapshot(out, d):
    return out.clone(), d["counts"].clone(), d["win"].clone(), d["hyp"].clone()


@pytest.mark.parametrize("thresh", [0.9, 0.99, 0.999])
@pytest.mark.parametrize("conc", [False, True])
def test_culled_counts_equal_literal_and_full_kernel_at_the_bench_shape(cull, thresh, conc):
    m, v, _ = batch(4, 40, 480, 640, 40)
    lit = snapshot(*voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=3, literal=True, return_debug=True))
    cull(False)
    full = snapshot(*voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=3, return_debug=True, concurrent=conc))
    cull(True)
    out, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=3, return_debug=True, concurrent=conc,
                                           band_stats=True)
    assert d["cull"] and bool(d["cull_bits"].all()) and d["layout"].cull == 1 and d["mode"] == "exact"
    assert d["hyp"].cpu().numpy().tobytes() == lit[3].cpu().numpy().tobytes()   # caller order, the same draws
    assert torch.equal(d["counts"], lit[1]) and torch.equal(d["counts"], full[1])
    assert torch.equal(d["win"], lit[2]) and torch.equal(d["win"], full[2])
    assert torch.equal(out, full[0])                       # same inlier sets, same float64 sums: bit-identical key-points
    ex, total = d["cull_stats"]
    assert 0 < ex < total                                   # and part of the full kernel's steps was really not executed
    if thresh == 0.9:
        assert ex < 0.6 * total


def test_selection_flags_on_the_release_library():
    """PVNET_F_CULL_ALL / PVNET_F_CULL_NONE (voting.set_cull_selection) on the RELEASE library -- no environment knob: the marks K3 records
    follow the flag, the library's own selection culls a clean field and leaves the noisy benchmark field alone, and every selection
    gives the same integers"""
    if any(os.environ.get(k) for k in voting.TUNING_KNOBS):
        pytest.skip("a PVNET_* knob is set in the environment: the front end loads the development build")
    voting.reload_tuning()
    assert b"release build" in voting.load_library().pvnet_vote_build_info()
    mn, vn_, _ = batch(8, 300, 480, 640, 40)                                    # noisy: sigma 0.05, 10 % outliers
    mc, vc, _ = batch(8, 300, 480, 640, 40, noise=False, background="zeros")    # the ground-truth field
    L = voting.vote_layout(8, 480, 640, 9, 1024, 30000)
    try:
        res = {}
        for sel in (None, "all", "none"):
            voting.set_cull_selection(sel)
            for name, (m, v) in (("noisy", (mn, vn_)), ("clean", (mc, vc))):
                ws = torch.zeros(L.total_bytes, dtype=torch.uint8, device=dev())   # no previous call on this workspace
                out, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=5, return_debug=True, workspace=ws)
                res[sel, name] = (out.clone(), d["counts"].clone(), d["win"].clone(), d["cull_bits"].clone())
    finally:
        voting.set_cull_selection(None)
    for name in ("noisy", "clean"):
        assert bool(res["all", name][3].all()) and not bool(res["none", name][3].any())
        for sel in (None, "all"):
            assert torch.equal(res[sel, name][1], res["none", name][1]) and torch.equal(res[sel, name][2], res["none", name][2])
            assert torch.equal(res[sel, name][0], res["none", name][0])
    assert bool(res[None, "clean"][3].all()) and not bool(res[None, "noisy"][3].any())   # what the spread test decides on these fields


def test_a_batch_follows_the_previous_batch_on_its_workspace():
    """whether a call's images may be disc-culled at all is decided from the PREVIOUS call on the same workspace (CF_BATCH_OK,
    vote_common.h): two thirds of its images must have voted for culling.  A batch of 2 clean + 6 noisy images: the first call on a fresh
    workspace culls the two clean ones, the second none; a clean batch keeps being culled; a clean batch behind a noisy one is scored
    densely once, then culled.  The key-points, counts and winners never change."""
    if any(os.environ.get(k) for k in voting.TUNING_KNOBS):
        pytest.skip("a PVNET_* knob is set in the environment")
    voting.reload_tuning()
    mn, vn_, _ = batch(8, 300, 480, 640, 40)
    mc, vc, _ = batch(8, 300, 480, 640, 40, noise=False, background="zeros")
    mm = torch.cat([mc[:2], mn[2:]])
    vm = torch.cat([vc[:2].contiguous(), vn_[2:].contiguous()])   # (a contiguous field here: strides are the caller's business)
    L = voting.vote_layout(8, 480, 640, 9, 1024, 30000)
    ws = torch.zeros(L.total_bytes, dtype=torch.uint8, device=dev())

    def call(m, v):
        out, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=9, return_debug=True, workspace=ws)
        return out.clone(), d["counts"].clone(), d["win"].clone(), d["cull_bits"].clone()
    a1, a2 = call(mm, vm), call(mm, vm)
    assert a1[3][:2].all() and not a1[3][2:].any()          # fresh workspace: every image that votes for it is culled
    assert not a2[3].any()                                   # 2 of 8 voted: the next batch is scored densely
    for x, y in zip(a1[:3], a2[:3]):
        assert torch.equal(x, y)
    c1, c2, c3 = call(mc, vc), call(mc, vc), call(mc, vc)    # behind a batch that did not vote for culling: dense once, then culled
    assert not c1[3].any() and c2[3].all() and c3[3].all()
    for x, y in zip(c1[:3], c3[:3]):
        assert torch.equal(x, y)
    n1 = call(mn, vn_)
    assert not n1[3].any()


@pytest.mark.skipif(not refkernels.available("off"), reason="oracle/_ref (the reference's kernels compiled for gfx950) not built")
def test_culled_counts_equal_the_references_own_kernel(cull):
    """the reference's voting_for_hypothesis_kernel itself (oracle/_ref) on the path's compacted pixels and hypotheses"""
    m, v, _ = batch(2, 60, 480, 640, 40)
    cull(True)
    _, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=11, return_debug=True)
    assert d["cull"]
    for bi in range(2):
        tn = int(d["tn"][bi])
        rec = d["rec"][bi, :, :tn]
        coords = rec[0, :, 0:2].contiguous()
        direct = rec[:, :, 2:4].permute(1, 0, 2).contiguous()
        hyp = d["hyp"][bi].permute(1, 0, 2).contiguous()
        inl = refkernels.voting_for_hypothesis(direct, coords, hyp, 0.99)
        assert torch.equal(d["counts"][bi].T, inl.sum(2, dtype=torch.int32))


def test_clean_field_is_almost_entirely_certain(cull):
    """every hypothesis of a clean field is the key-point itself: tiles of radius ~0, (nearly) every pixel certain -- the fine
    pass has next to nothing left, and the winner still collects every pixel"""
    m, v, kpts = batch(3, 70, 480, 640, 40, noise=False, background="zeros")
    cull(True)
    out, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=2, return_debug=True, band_stats=True)
    ex, total = d["cull_stats"]
    assert d["cull"] and ex < 0.05 * total
    assert (d["win"][:, :, 1] == d["tn"][:3, None]).all()
    assert np.abs(out.cpu().numpy() - kpts).max() < 1e-3
    lit_out, dl = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=2, literal=True, return_debug=True)
    assert torch.equal(d["counts"], dl["counts"])


@pytest.mark.parametrize("hn,b", [(777, 3), (1024, 1), (1000, 2), (1023, 32), (769, 1)])
def test_padding_hypotheses_and_small_batches(cull, hn, b):
    """hn < 1024: padding columns inside the last tile and whole padding tiles (they sort behind every real hypothesis and are
    never scored); b = 1: PVNET_SCORE_CULL=1 gives a small batch 256-pixel items too"""
    m, v, _ = batch(b, 80 + hn, 240, 320, 30)
    lit = snapshot(*voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=5, literal=True, return_debug=True))
    cull(True)
    out, d = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=5, return_debug=True)
    assert d["cull"] and bool(d["cull_bits"].all())
    assert torch.equal(d["counts"], lit[1]) and torch.equal(d["win"], lit[2])
    assert d["hyp"].cpu().numpy().tobytes() == lit[3].cpu().numpy().tobytes()
    assert float((out - lit[0]).abs().max()) < 1e-3


def test_layouts_the_culling_kernel_does_not_cover_run_the_full_kernel(cull):
    cull(True)
    m, v, _ = batch(2, 90, 240, 320, 24)
    for hn in (256, 2048, 8192):   # 2 hypothesis tiles per wave / more than the one slice of 1 024 hypotheses a K3 block sorts
        _, lit = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=6, literal=True, return_debug=True)
        cl = lit["counts"].clone()
        _, d = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=6, return_debug=True)
        assert not d["cull"] and d["mode"] == "exact" and torch.equal(d["counts"], cl)


def test_unnormalised_dead_and_far_inputs(cull):
    """|u| scaled per pixel over many orders of magnitude, zero / sub-gate / NaN / Inf directions (dead rows: certain non-votes),
    near-parallel fields whose hypotheses lie 1e6 px away (tiles with huge discs: everything uncertain, scored in full)"""
    rng = np.random.default_rng(17)
    mask, planar, _ = synth.make_batch(2, first_index=95, h=240, w=320, radius=30, noise=True, background="normal")
    fac = np.exp(rng.normal(0.0, 3.0, size=(2, 1, 240, 320))).astype(np.float32)
    fac[rng.random(fac.shape) < 0.02] = 0.0
    fac[rng.random(fac.shape) < 0.02] = 1e-7
    planar = (planar.reshape(2, 9, 2, 240, 320) * fac[:, :, None]).reshape(2, 18, 240, 320).astype(np.float32)
    ys, xs = np.nonzero(mask[0])
    planar[0, 0, ys[::11], xs[::11]] = np.nan
    planar[0, 3, ys[::13], xs[::13]] = np.inf
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    _, lit = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=8, literal=True, return_debug=True)
    cl, wl = lit["counts"].clone(), lit["win"].clone()
    cull(True)
    _, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=8, return_debug=True)
    assert d["cull"] and torch.equal(d["counts"], cl) and torch.equal(d["win"], wl)
    # far key-points: all directions of a key-point nearly parallel
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w]
    fg = ((xx - 64) ** 2 + (yy - 48) ** 2) <= 20 ** 2
    far = synth.field_from_keypoints(fg, np.array([[2.0e6, 48.0], [-3.0e8, 1.0e8]]))
    m2 = torch.from_numpy(fg[None].astype(np.uint8)).to(dev())
    v2 = synth.planar_to_vertex_view(torch.from_numpy(far[None].astype(np.float32)).to(dev()))
    cull(False)
    _, lit2 = voting.ransac_voting_layer_v3(m2, v2, 1024, inlier_thresh=0.99, seed=9, literal=True, return_debug=True)
    c2 = lit2["counts"].clone()
    cull(True)
    _, d2 = voting.ransac_voting_layer_v3(m2, v2, 1024, inlier_thresh=0.99, seed=9, return_debug=True)
    assert d2["cull"] and torch.equal(d2["counts"], c2)


def test_siblings_read_caller_order_counts_and_the_band_margin_still_holds(cull):
    """the sibling epilogues (distribution, hypothesis counts) read `counts` / `hyp` in CALLER order -- K5 returns the culled
    counts to it -- and the measured safety margin of the rounding band is evaluated on the sorted operands"""
    m, v, _ = batch(2, 99, 240, 320, 30)
    cull(False)
    hyp0, cnt0 = voting.generate_hypothesis_counts(m, v, 1024, inlier_thresh=0.99, seed=4)
    mean = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=4)
    _, cov0 = voting.estimate_voting_distribution_with_mean(m, v, mean, round_hyp_num=256, min_hyp_num=1024, seed=4)
    cov0 = cov0.clone()
    cull(True)
    hyp1, cnt1 = voting.generate_hypothesis_counts(m, v, 1024, inlier_thresh=0.99, seed=4)
    assert torch.equal(hyp0, hyp1) and torch.equal(cnt0, cnt1)
    _, cov1 = voting.estimate_voting_distribution_with_mean(m, v, mean, round_hyp_num=256, min_hyp_num=1024, seed=4)
    assert torch.equal(cov0, cov1)
    _, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=4, return_debug=True)
    bm = voting.band_margin(d, 0.99)
    assert bm["tests"] > 1e7 and bm["worst"] < 0.5


def test_the_library_selects_clean_key_points_and_leaves_noisy_ones_to_the_full_kernel(cull):
    """the default: K3 decides per (image, key-point) from the spread of the band-origin candidates -- a clean field's key-points are
    culled, the noisy benchmark field's are not, and a batch that holds both gets both kernels in one call; every count equals
    literal mode's either way"""
    cull("auto")

    def fresh():   # a workspace without a previous call: whether a batch may be culled at all follows the previous call on its workspace
        return torch.zeros(voting.vote_layout(32, 480, 640, 9, 1024, 30000).total_bytes, dtype=torch.uint8, device=dev())
    mc, vc, _ = batch(32, 300, 480, 640, 40, noise=False, background="zeros")
    _, d = voting.ransac_voting_layer_v3(mc, vc, 1024, inlier_thresh=0.99, seed=1, return_debug=True, band_stats=True, workspace=fresh())
    assert d["layout"].cull == 1 and bool(d["cull_bits"].all())
    ex, total = d["cull_stats"]
    assert total > 0 and ex < 0.05 * total
    cc = d["counts"].clone()
    _, dl = voting.ransac_voting_layer_v3(mc, vc, 1024, inlier_thresh=0.99, seed=1, literal=True, return_debug=True)
    assert torch.equal(cc, dl["counts"])
    mn, vn_, _ = batch(32, 300, 480, 640, 40)
    _, d = voting.ransac_voting_layer_v3(mn, vn_, 1024, inlier_thresh=0.99, seed=1, return_debug=True, band_stats=True, workspace=fresh())
    assert d["layout"].cull == 1 and float(d["cull_bits"].float().mean()) < 0.1   # (a lucky draw of candidates may cull a key-point)
    # a mixed batch: even images clean, odd images noisy
    mm, vm = mn.clone(), vn_.clone()
    mm[0::2], vm[0::2] = mc[0::2], vc[0::2]
    out, d = voting.ransac_voting_layer_v3(mm, vm, 1024, inlier_thresh=0.99, seed=1, return_debug=True, band_stats=True, workspace=fresh())
    bits = d["cull_bits"].clone()
    assert bool(bits[0::2].all()) and float(bits[1::2].float().mean()) < 0.1
    cm, wm = d["counts"].clone(), d["win"].clone()
    out = out.clone()
    lo, dl = voting.ransac_voting_layer_v3(mm, vm, 1024, inlier_thresh=0.99, seed=1, literal=True, return_debug=True)
    assert torch.equal(cm, dl["counts"]) and torch.equal(wm, dl["win"])
    assert float((out - lo).abs().max()) < 1e-3
    cull(False)
    of, df = voting.ransac_voting_layer_v3(mm, vm, 1024, inlier_thresh=0.99, seed=1, return_debug=True)
    assert df["layout"].cull == 0 and torch.equal(df["counts"], cm) and torch.equal(of, out)


def hilbert_index(x, y, bits):
    """numpy restatement of hilbert_index() (k3_hypotheses.hip): position of integer cells (x, y) on the 2^bits x 2^bits Hilbert curve"""
    x, y = x.astype(np.uint32).copy(), y.astype(np.uint32).copy()
    d = np.zeros_like(x)
    s = np.uint32(1 << (bits - 1))
    while s > 0:
        rx, ry = ((x & s) != 0).astype(np.uint32), ((y & s) != 0).astype(np.uint32)
        d += s * s * ((np.uint32(3) * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x, y = np.where(flip, ~x, x), np.where(flip, ~y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s = np.uint32(s >> 1)
    return d


def test_the_sorted_order_is_a_permutation_along_the_hilbert_curve(cull):
    """the K3 block of a culled key-point: `perm` must be a permutation of the hypothesis indices (padding behind), `hyps` the
    hypotheses in that order, and the order the one of the sort keys -- (Hilbert position of the hypothesis' 1/8-pixel cell about
    the band origin, caller index): the 256-thread register / DPP / permlane sorting network against numpy's sort of the same keys"""
    m, v, _ = batch(3, 120, 480, 640, 40)
    cull(True)
    hn = 1000
    _, d = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=13, return_debug=True)
    assert bool(d["cull_bits"].all())
    perm, hyps, hyp = d["perm"].cpu().numpy(), d["hyps"].cpu().numpy(), d["hyp"].cpu().numpy()
    org = d["band_origin"].cpu().numpy().astype(np.float32)
    for bi in range(3):
        for k in range(9):
            p = perm[bi, k]
            assert sorted(p.tolist()) == list(range(1024))
            assert (p[:hn] < hn).all() and (p[hn:] >= hn).all()          # padding sorts behind every real hypothesis
            assert hyps[bi, k, :hn].tobytes() == hyp[bi, k][p[:hn]].tobytes()
            cells = np.float32(2048.0)
            fx = np.clip((hyp[bi, k, :, 0] - org[bi, k, 0]) * np.float32(8.0) + np.float32(0.5) * cells, 0, cells - 1)
            fy = np.clip((hyp[bi, k, :, 1] - org[bi, k, 1]) * np.float32(8.0) + np.float32(0.5) * cells, 0, cells - 1)
            key = (hilbert_index(fx.astype(np.uint32), fy.astype(np.uint32), 11).astype(np.uint64) << np.uint64(10)) | np.arange(hn, dtype=np.uint64)
            assert (p[:hn] == np.argsort(key, kind="stable")).all()
