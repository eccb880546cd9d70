"""GPU tests of the DEFAULT scoring mode ("exact mode", ABI 6): matrix-pipe scoring whose integer products -- the inlier
count of every hypothesis and the winners -- EQUAL those of the reference's own device code
(lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu:88-126 compiled for gfx950 from the reference tree, oracle/_ref)
and of the library's literal mode (which other tests pin bit-for-bit to that device code).

No tolerance, no agreement mask: `torch.equal` on the whole [b, vn, hn] count tensor."""
import numpy as np
import pytest
import torch

from oracle import refkernels
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth, voting

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cull_selection")]   # (tests/conftest.py: both selections of the disc culling)


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def batch(n, first, h, w, radius, noise=True, background="normal", **kw):
    mask, planar, kpts = synth.make_batch(n, first_index=first, h=h, w=w, radius=radius, noise=noise,
                                          background=background, **kw)
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    return m, v, kpts


def both_modes(m, v, hn, thresh, seed=7, **kw):
    out_l, lit = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, literal=True,
                                               return_debug=True, **kw)
    lit = {k: (x.clone() if torch.is_tensor(x) else x) for k, x in lit.items() if k != "workspace"}
    out_e, ex = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, return_debug=True,
                                              band_stats=True, **kw)
    assert ex["mode"] == "exact" and lit["mode"] == "literal"
    return out_l, lit, out_e, ex


def assert_same_integers(lit, ex, b):
    assert torch.equal(ex["tn"][:b], lit["tn"][:b])
    assert ex["hyp"].cpu().numpy().tobytes() == lit["hyp"].cpu().numpy().tobytes()
    bad = (ex["counts"] != lit["counts"])
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} counts differ, max |diff| " \
                                f"{int((ex['counts'] - lit['counts']).abs().max())}"
    assert torch.equal(ex["win"], lit["win"])


@pytest.fixture(params=[0, 1], ids=["cell=item", "cell=tile"])
def fold(request, monkeypatch):
    monkeypatch.setenv("PVNET_EXACT_FOLD", str(request.param))
    voting.reload_tuning()
    yield request.param
    monkeypatch.delenv("PVNET_EXACT_FOLD")
    voting.reload_tuning()


@pytest.mark.skipif(not refkernels.available("off"), reason="oracle/_ref not built (needs the reference tree)")
@pytest.mark.parametrize("thresh", [0.9, 0.99, 0.999])
@pytest.mark.parametrize("radius,hn", [(40, 1024), (97, 1024), (22, 200)])
def test_counts_equal_reference_device_code(thresh, radius, hn):
    """the VERDICT's bar: default-mode counts `torch.equal` to the counts of the reference's own voting kernel, at the
    benchmark shape (480x640, tn ~ 5 000) and with masks just under max_num (tn ~ 29 500), three thresholds."""
    b = 2 if radius == 97 else 3
    m, v, _ = batch(b, 900 + radius, 480, 640, radius)
    seed = 13
    out, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, return_debug=True,
                                             band_stats=True)
    cells, tests = dbg["band_stats"]
    for bi in range(b):
        tn = int(dbg["tn"][bi])
        rec = dbg["rec"][bi, :, :tn]
        coords = rec[0, :, 0:2].contiguous()
        direct = rec[:, :, 2:4].permute(1, 0, 2).contiguous()
        idxs = torch.from_numpy(O.draw_idxs(seed, bi, hn, 9, tn)).to(dev())
        hyp_ref = refkernels.generate_hypothesis(direct, coords, idxs)
        assert dbg["hyp"][bi].permute(1, 0, 2).contiguous().cpu().numpy().tobytes() == hyp_ref.cpu().numpy().tobytes()
        counts_ref = torch.zeros((hn, 9), dtype=torch.int32, device=dev())
        step = max(1, (1 << 28) // (9 * tn))                      # the reference's [hn, vn, tn] byte tensor, in slices
        for h0 in range(0, hn, step):
            inl = refkernels.voting_for_hypothesis(direct, coords, hyp_ref[h0:h0 + step].contiguous(), thresh)
            counts_ref[h0:h0 + step] = inl.sum(2, dtype=torch.int32)
        ours = dbg["counts"][bi].T.contiguous()
        assert torch.equal(ours, counts_ref), \
            f"image {bi}: {int((ours != counts_ref).sum())} counts differ (max {int((ours - counts_ref).abs().max())})"
        first = (counts_ref == counts_ref.max(0).values[None]).int().argmax(0)
        assert torch.equal(dbg["win"][bi, :, 0].long(), first)
        assert torch.equal(dbg["win"][bi, :, 1], counts_ref.max(0).values)
    total = sum(int(dbg["tn"][bi]) for bi in range(b)) * 9 * hn
    assert tests < 0.2 * total, f"{tests} of {total} tests re-evaluated literally: the band is not doing its job"


@pytest.mark.parametrize("thresh", [0.5, 0.9, 0.99, 0.999, 0.9999])
def test_counts_equal_literal_mode(thresh, fold):
    m, v, _ = batch(4, 950, 300, 400, 30)
    _, lit, _, ex = both_modes(m, v, 512, thresh)
    assert_same_integers(lit, ex, 4)


def test_benchmark_inputs_equal_literal_and_keypoints_close():
    """the inputs bench.py times (BASELINE configs[2]): all 32 x 9 x 1024 counts equal literal mode's, same winners,
    key-points within 1e-3 px (the refinement sums run over the same inlier set; only float64 summation order differs)."""
    m, v, _ = batch(32, 0, 480, 640, 40)
    out_l, lit, out_e, ex = both_modes(m, v, 1024, 0.99, seed=20240)
    assert_same_integers(lit, ex, 32)
    assert float((out_l - out_e).norm(dim=2).max()) < 1e-3
    cells, tests = ex["band_stats"]
    total = int(ex["tn"][:32].sum()) * 9 * 1024
    assert tests < 0.01 * total


def test_clean_field_and_thinned_masks(fold):
    m, v, kp = batch(3, 40, 480, 640, 60, noise=False, background="zeros")
    out_l, lit, out_e, ex = both_modes(m, v, 256, 0.999)
    assert_same_integers(lit, ex, 3)
    assert float((out_e.cpu() - torch.from_numpy(kp[:3]).float()).norm(dim=2).max()) < 1e-2
    # tn0 > max_num: the thinned pixel list is the same in both modes
    _, lit, _, ex = both_modes(m, v, 128, 0.99, max_num=2000)
    assert int(ex["tn0"][0]) > 2500 > int(ex["tn"][0]) > 1500  # thinned to ~max_num (+ < tn0 / 1024)
    assert_same_integers(lit, ex, 3)


def test_field_scales_and_degenerate_directions(fold):
    """un-normalised fields (the vote is scale-invariant, the reference's gates are not), zero / tiny / huge / NaN / Inf
    directions, hypotheses exactly on a pixel and at (0, 0) (degenerate pairs), far hypotheses (near-parallel pairs)."""
    m, v, _ = batch(2, 330, 240, 320, 26)
    v = v.clone()
    fg = m[0].nonzero()
    ys, xs = fg[:, 0], fg[:, 1]
    v[0, ys[0:40], xs[0:40]] = 0.0                       # zero directions: never vote (kernel.cu:121)
    v[0, ys[40:60], xs[40:60]] *= 1e-7                   # |u| below the 1e-6 gate
    v[0, ys[60:70], xs[60:70]] *= 1.0000001e-6           # ... and right at it
    v[0, ys[70:90], xs[70:90]] *= 1e12                   # large but finite: decided like unit vectors
    v[0, ys[90:100], xs[90:100]] *= 3e19                 # nx * nx overflows in the reference: it never votes there
    v[0, ys[100:105], xs[100:105], :, 0] = float("nan")
    v[0, ys[105:110], xs[105:110], :, 1] = float("inf")
    v[0, ys[110:130], xs[110:130]] = v[0, ys[111:131], xs[111:131]]  # parallel neighbours: hypotheses far away or (0, 0)
    hn = 256
    tn0 = int(m[0].sum())
    idxs = torch.from_numpy(np.random.default_rng(5).integers(0, tn0, (2, hn, 9, 2), dtype=np.int32)).to(dev())
    idxs[:, 0:8, :, 1] = idxs[:, 0:8, :, 0]              # the same pixel twice: hypothesis (0, 0)
    idxs[0, 8:40, :, 0] = torch.arange(110, 142, device=dev(), dtype=torch.int32)[:, None] % 129
    idxs[0, 8:40, :, 1] = idxs[0, 8:40, :, 0] + 1        # neighbouring, mostly parallel pixels
    for thresh in (0.99, 0.999):
        _, lit, _, ex = both_modes(m, v, hn, thresh, idxs=idxs)
        assert_same_integers(lit, ex, 2)
    for scale in (2.0 ** -10, 3.7, 2.0 ** 30):
        _, lit, _, ex = both_modes(m, v * scale, hn, 0.99, idxs=idxs)
        assert_same_integers(lit, ex, 2)


def test_hypothesis_on_pixels_and_tiny_objects(fold):
    """every hypothesis sits EXACTLY on a foreground pixel (norm2 < 1e-6 for that pixel, kernel.cu:121) -- axis-aligned
    unit vectors make the intersections exact integers; objects of 5 .. 40 pixels; odd hypothesis counts (padded slices)."""
    h, w = 64, 96
    m = torch.zeros((3, h, w), dtype=torch.int64, device=dev())
    v = torch.zeros((3, h, w, 9, 2), dtype=torch.float32, device=dev())
    m[0, 20:26, 30:37] = 1
    m[1, 10, 5:10] = 1                                   # exactly min_num pixels
    m[2, 40:44, 50:60] = 1
    ang = torch.tensor([0.0, np.pi / 2, np.pi, 3 * np.pi / 2], device=dev())
    pick = torch.randint(0, 4, (3, h, w, 9), device=dev(), generator=torch.Generator(device=dev()).manual_seed(3))
    v[..., 0] = torch.cos(ang)[pick].round()
    v[..., 1] = torch.sin(ang)[pick].round()
    for hn in (128, 200, 333):
        _, lit, _, ex = both_modes(m, v, hn, 0.99, seed=hn)
        assert_same_integers(lit, ex, 3)
        assert int(lit["counts"].max()) > 0


def test_approx_mode_is_still_available_and_close():
    m, v, _ = batch(2, 820, 240, 320, 24)
    _, lit = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=5, literal=True, return_debug=True)
    lit_counts = lit["counts"].clone()
    _, ap = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=5, approx=True, return_debug=True)
    assert ap["mode"] == "approx"
    diff = (ap["counts"] - lit_counts).abs()
    tn = int(ap["tn"][:2].sum())
    assert diff.sum().item() <= 2e-6 * 256 * 9 * tn + 2 and diff.max().item() <= 3


@pytest.mark.gpu
def test_concurrent_hint_is_result_neutral_and_follows_the_streams():
    """PVNET_F_CONCURRENT picks the one-accumulator scoring variant: counts, winners and key-points stay bit-identical; the
    Python front end sets it when consecutive calls alternate streams and clears it when they stay on one."""
    m, v, _ = batch(4, 321, 480, 640, 40)
    ref, dref = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=5, return_debug=True, concurrent=False)
    assert dref["concurrent"] is False
    for thresh in (0.9, 0.99, 0.999):
        a, da = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=5, return_debug=True, concurrent=False)
        b, db = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=5, return_debug=True, concurrent=True)
        assert db["concurrent"] is True and da["concurrent"] is False
        assert torch.equal(da["counts"], db["counts"]) and torch.equal(a, b)
    s1, s2 = torch.cuda.Stream(dev()), torch.cuda.Stream(dev())
    torch.cuda.synchronize()
    seen = []
    for st in (s1, s1, s2, s1, s1):
        with torch.cuda.stream(st):
            out, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=5, return_debug=True)
            seen.append(d["concurrent"])
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    assert seen[1:] == [False, True, True, False]


@pytest.mark.gpu
@pytest.mark.parametrize("hpl,chunk,hn", [(8, 384, 1024), (2, 224, 128), (8, 480, 1024)])
def test_items_of_more_than_21_pixel_tiles_count_like_literal(hpl, chunk, hn, monkeypatch):
    """cells of one pixel tile list a flagged (hypothesis, half-wave) as a tile mask above 11 index bits -- 21 tiles.
    Layouts with larger work items (24, 28, 30 tiles here) fall back to the cell = work item form instead of losing flags."""
    m, v, _ = batch(2, 77, 480, 640, 45)
    _, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.999, seed=4, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    monkeypatch.setenv("PVNET_SCORE_HPL", str(hpl))
    monkeypatch.setenv("PVNET_SCORE_CHUNK", str(chunk))
    voting.reload_tuning()
    try:
        for conc in (False, True):
            _, de = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.999, seed=4, return_debug=True, concurrent=conc)
            L = de["layout"]
            assert L.wg_s * L.chunk // 32 > 21
            assert torch.equal(de["counts"], cl), (hpl, chunk, hn, conc)
    finally:
        monkeypatch.delenv("PVNET_SCORE_HPL")
        monkeypatch.delenv("PVNET_SCORE_CHUNK")
        voting.reload_tuning()


@pytest.mark.gpu
def test_default_mode_without_the_matrix_pipe_buffers_is_scored_literally(monkeypatch):
    """PVNET_SCORE_MODE=0 leaves no B-operand buffer; the default mode must then give the reference's counts by
    literal scoring (and say so), not the approximate VALU predicate's."""
    m, v, _ = batch(2, 91, 240, 320, 24)
    _, dl = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=8, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    monkeypatch.setenv("PVNET_SCORE_MODE", "0")
    voting.reload_tuning()
    try:
        _, d = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=8, return_debug=True)
        assert d["mode"] == "literal" and torch.equal(d["counts"], cl)
        _, da = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=8, approx=True, return_debug=True)
        assert da["mode"] == "approx"
    finally:
        monkeypatch.delenv("PVNET_SCORE_MODE")
        voting.reload_tuning()
    _, d = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=8, return_debug=True)
    assert d["mode"] == "exact" and torch.equal(d["counts"], cl)


@pytest.mark.gpu
@pytest.mark.parametrize("thresh", [0.99999, 0.999999])
def test_thresholds_next_to_one(thresh):
    """the band is computed on both sides of the threshold angle (band_constant): at 0.99999 / 0.999999 the
    first-order form was 1.5 % / 20 % short on the vote side.  Counts equal literal's; the measured margin stays below 1."""
    m, v, _ = batch(4, 611, 480, 640, 40, noise=False)   # a clean field: most tests sit right at such a threshold
    _, dl = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=3, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    _, de = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=3, return_debug=True)
    assert de["mode"] == "exact" and torch.equal(de["counts"], cl)
    r = voting.band_margin(de, thresh)
    assert r["tests"] > 1e8 and r["worst"] < 0.5, r


@pytest.mark.gpu
def test_band_margin_is_measured_not_assumed():
    """on the benchmark field and on large objects, thresholds 0.9 / 0.99 / 0.999, every test is evaluated on
    the matrix pipe as the scoring kernel does and with the reference's arithmetic: the largest |x| of a test on which the two
    disagree must stay far below 1 (where the kernel stops trusting x) -- tools/band_margin.py, profiles/r04_band_margin.txt."""
    worst, tests = 0.0, 0
    for b, radius in ((8, 40), (2, 97)):
        m, v, _ = batch(b, 4242, 480, 640, radius)
        for thresh in (0.9, 0.99, 0.999):
            _, d = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=1, return_debug=True)
            r = voting.band_margin(d, thresh)
            assert r["band"] > 0 and r["tests"] == int(d["tn"][:b].sum()) * 9 * 1024
            worst, tests = max(worst, r["worst"]), tests + r["tests"]
    assert tests > 2e9 and worst < 0.5, (tests, worst)


@pytest.mark.gpu
def test_band_origin_is_the_keypoint_where_that_helps_and_never_matters_for_the_result():
    """Round 4: the exact mode's band is centred on an estimate of the key-point (hypothesis kernel) unless the candidate
    intersections scatter (near-parallel fields): the origin moves the number of re-evaluated cells, never a count."""
    mask, planar, kpts = synth.make_batch(3, first_index=700, h=480, w=640, radius=40, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    _, dl = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.999, seed=2, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    _, de = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.999, seed=2, return_debug=True, band_stats=True)
    assert torch.equal(de["counts"], cl)
    org = de["band_origin"].cpu().numpy().astype(np.float64)[:, :, :2]
    err = np.abs(org - kpts[:, :, :2]).max(axis=2)            # [image, key-point]
    assert np.median(err) < 6.0 and (err < 20.0).mean() >= 0.8, err   # a median of eight noisy intersections: a few pixels off
    cells_kp = de["band_stats"][0]
    # a field whose lines are nearly parallel: intersections 1e4 .. 1e6 px away, scattered -> the median pixel is kept
    squeezed = planar.copy()
    squeezed[:, 1::2] *= 1e-4
    v2 = synth.planar_to_vertex_view(torch.from_numpy(squeezed).to(dev()))
    _, dl2 = voting.ransac_voting_layer_v3(m, v2, 512, inlier_thresh=0.999, seed=2, literal=True, return_debug=True)
    cl2 = dl2["counts"].clone()
    _, de2 = voting.ransac_voting_layer_v3(m, v2, 512, inlier_thresh=0.999, seed=2, return_debug=True, band_stats=True)
    assert torch.equal(de2["counts"], cl2)
    tests = int(de2["tn"][:3].sum()) * 9 * 512
    assert de2["band_stats"][1] < 1e-3 * tests, de2["band_stats"]   # (a far-away origin would re-evaluate most of them)
    assert cells_kp > 0


@pytest.mark.gpu
def test_more_than_32_keypoints_use_the_image_origin():
    """the key-point origin is worked out for up to 32 key-points per image; beyond that the median pixel serves all of them"""
    rng = np.random.default_rng(5)
    h, w, vn = 120, 160, 40
    mask = np.zeros((1, h, w), np.int64)
    mask[0] = synth.disk_mask(h, w, 80, 60, 22)
    kp = np.stack([rng.uniform(40, 120, vn), rng.uniform(30, 90, vn)], 1)
    planar = synth.add_noise(synth.field_from_keypoints(mask[0].astype(bool), kp), mask[0].astype(bool), rng)[None]
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    _, dl = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=1, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    _, de = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=1, return_debug=True)
    assert torch.equal(de["counts"], cl)
    org = de["band_origin"][0].cpu().numpy()
    assert (org == org[0]).all()   # one origin for all 40


@pytest.mark.parametrize("vn", [32, 33, 40])
def test_many_key_points(vn):
    """the band origin (and the culling selection) is estimated per key-point for up to 32 of them -- eight candidate lanes each, all 256
    threads of a K3 block at vn = 32; beyond that every key-point takes the image's median pixel as origin and nothing is culled.
    Either way: the reference's integers."""
    m, v, _ = batch(3, 910, 120, 160, 15, vn=vn)
    _, lit, _, ex = both_modes(m, v, 256, 0.99)
    assert_same_integers(lit, ex, 3)
    mc, vc, _ = batch(3, 910, 120, 160, 15, vn=vn, noise=False, background="zeros")
    ol, lit, oe, ex = both_modes(mc, vc, 256, 0.99)
    assert_same_integers(lit, ex, 3)
    assert float((ol - oe).abs().max()) < 1e-3
