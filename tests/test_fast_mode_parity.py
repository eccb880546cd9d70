"""GPU parity tests of the two matrix-pipe scoring modes against literal mode (the reference's float32 order).

DEFAULT = exact mode, the mode bench.py times: every inlier count and every winner EQUAL literal mode's -- asserted with
`torch.equal`, no tolerance, no "if the winner differs" branch -- on the benchmark inputs, across
thresholds, randomised shapes, field scales, far hypotheses, NaN / Inf directions and every tiling.
approx=True (PVNET_F_APPROX, the round-1/2 "fast" mode): counts within 2 votes per hypothesis and 2e-6 of the pair tests of
literal's, never farther from float64 arithmetic than the reference's own float32 order is."""
import numpy as np
import pytest
import torch

import bench
from oracle import cref
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth, voting

pytestmark = pytest.mark.gpu

TOL_PX = 1e-3


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def to_dev(mask, planar):
    return (torch.from_numpy(np.ascontiguousarray(mask)).to(dev()),
            synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev())))


# ------------------------------------------------------------------------------------------------ (a) bench inputs
def test_timed_mode_on_the_exact_bench_inputs():
    """bench.py's own inputs (first two input sets: first_index 0 and 32, radius 40, noisy field, N(0,1) background), its
    seeds (SEED0 + step), 1024 hypotheses, thresh 0.99: 2 x 288 key-points x 1024 counts, all EQUAL literal mode's, the
    first 8 images also EQUAL the reference's own voting kernel (oracle/_ref); no mask, no alternative branch."""
    sets = bench.make_inputs(0, 2, 40, True, dev())
    r = bench.parity_check(sets, 0, o64_images=bench.BATCH)
    print("bench parity:", r)
    n = r["keypoints_checked"]
    assert n == 2 * 32 * 9
    assert r["counts_equal_literal"] == n and r["max_count_diff_vs_literal"] == 0
    assert r["reference_keypoints_checked"] in (0, 72)
    assert r["counts_equal_reference"] in (None, r["reference_keypoints_checked"])
    assert r["literal_winners_equal_c_oracle"] == n       # literal mode IS the reference's arithmetic
    assert r["winners_equal"] and r["winners_equal_literal"] == n and r["winners_equal_c_oracle"] == n
    assert r["max_px_vs_c_oracle"] <= TOL_PX               # all 576 key-points, C oracle (f32 votes, f64 LSQ)
    assert r["max_px_vs_literal"] <= TOL_PX
    assert r["max_px_vs_oracle64"] <= TOL_PX               # all 288 key-points of set 0, float64 oracle
    assert r["pass"] is True


# ------------------------------------------------------------------------------------------------ (b) count bounds
@pytest.mark.parametrize("thresh", [0.9, 0.99, 0.999])
def test_approx_counts_against_exact_arithmetic_and_literal(thresh):
    """inlier counts of one draw in three arithmetics: float64 (oracle64, exact for this purpose), the reference's
    float32 order (literal mode) and the timed fast mode.  Bars: fast vs float64 <= 2 per hypothesis and <= 2e-6 of the
    pair tests; fast never farther from float64 than the reference's own arithmetic is (whose cos-based float32 test
    loses resolution as thresh -> 1: ~6e-7 of its decisions differ from exact at 0.99, several e-6 at 0.999); fast vs
    literal <= 2 per hypothesis."""
    mask, planar, _ = synth.make_batch(4, first_index=3000, h=240, w=320, radius=22, noise=True, background="normal")
    vnp = synth.planar_to_vertex_view(planar)
    m, v = to_dev(mask, planar)
    hn = 512
    _, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=31, literal=True, return_debug=True)
    dl = {k: (x.clone() if torch.is_tensor(x) else x) for k, x in dl.items() if k != "workspace"}
    cl, hyp = dl["counts"].clone(), dl["hyp"].clone()
    _, de = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=31, return_debug=True)
    assert torch.equal(de["counts"], cl) and torch.equal(de["win"], dl["win"])  # exact mode: the reference's integers
    _, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=31, approx=True, return_debug=True)
    assert torch.equal(df["hyp"], hyp)  # hypothesis generation is the literal order in every mode
    cf = df["counts"].cpu().numpy()
    cl = cl.cpu().numpy()
    hyp = hyp.cpu().numpy()
    tests = err_f = err_l = 0
    for bi in range(4):
        coords, direct = O.compact(O.foreground(mask[bi]), vnp[bi])
        c64 = O.voting_counts(direct, coords, hyp[bi].transpose(1, 0, 2).astype(np.float64), thresh, np.float64)  # [hn,vn]
        tests += hn * 9 * coords.shape[0]
        df_ = np.abs(cf[bi].T - c64)
        dl_ = np.abs(cl[bi].T - c64)
        assert df_.max() <= 2
        err_f += int(df_.sum())
        err_l += int(dl_.sum())
    print(f"thresh {thresh}: {tests} pair tests; decisions differing from float64: fast {err_f} ({err_f / tests:.1e}), "
          f"literal (reference order) {err_l} ({err_l / tests:.1e})")
    assert err_f <= max(2, 2e-6 * tests), (err_f, tests)
    assert err_f <= err_l + 2
    assert int(np.abs(cf - cl).max()) <= 2


@pytest.mark.parametrize("case", range(12))
def test_randomised_shapes_both_modes_vs_literal_counts(case):
    """the randomised sweep of test_hip_parity (same shapes and seeds), fast mode against literal counts"""
    rng = np.random.default_rng(1000 + case)
    h, w = int(rng.integers(24, 200)), int(rng.integers(24, 260))
    vn = int(rng.integers(1, 13))
    hn = int(rng.choice([16, 40, 64, 129, 300, 777]))
    b = int(rng.integers(1, 5))
    radius = int(rng.integers(4, max(5, min(h, w) // 3)))
    thresh = float(rng.choice([0.9, 0.99, 0.999]))
    max_num = int(rng.choice([30000, 200, 60]))
    mask, planar, _ = synth.make_batch(b, first_index=2000 + 7 * case, h=h, w=w, vn=vn, radius=radius, noise=True,
                                       background="normal", mask_dtype=np.uint8)
    m, v = to_dev(mask, planar)
    seed = 50 + case
    lit, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, literal=True,
                                            return_debug=True)
    cl, wl, lit = dl["counts"].clone(), dl["win"].clone(), lit.clone()
    tn = int(dl["tn"].sum())
    ex, de = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed,
                                           return_debug=True)
    assert torch.equal(de["counts"], cl) and torch.equal(de["win"], wl), "exact mode: the reference's integers"
    ok = torch.isfinite(lit).all(-1) & (lit.abs() < 1e5).all(-1)
    if ok.any():  # same inlier sets: the refined points differ by float64 summation order only
        assert float((ex - lit)[ok].abs().max()) < 1e-3 * max(1.0, float(lit[ok].abs().max()) / 100)
    fast, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, approx=True,
                                             return_debug=True)
    d = (df["counts"] - cl).abs()
    assert int(d.max()) <= 2
    assert int(d.sum()) <= max(3, 2e-5 * hn * vn * tn)  # literal = the reference's own float32 noise (see above)
    same = (df["win"][:, :, 0] == wl[:, :, 0])
    # where the winner is the same hypothesis the refined points agree to the tolerance (same inliers up to edge votes)
    good = torch.isfinite(lit).all(-1) & (lit.abs() < 1e5).all(-1) & same
    if good.any():
        scale = max(1.0, float(lit[good].abs().max()) / 100)
        assert float((fast - lit)[good].abs().max()) < 2e-3 * scale
    # a different winner is only ever a tie-break between hypotheses whose counts the modes see within 2 votes
    if (~same).any():
        bi, ki = torch.nonzero(~same, as_tuple=True)
        gap = (cl[bi, ki, df["win"][bi, ki, 0].long()] - cl[bi, ki, wl[bi, ki, 0].long()]).abs()
        assert int(gap.max()) <= 2


# ------------------------------------------------------------------------------------------------ (c) robustness
def _field_case(b=2, h=120, w=160, radius=18, first=4000):
    mask, planar, kpts = synth.make_batch(b, first_index=first, h=h, w=w, radius=radius, noise=True,
                                          background="normal")
    return mask, planar, kpts


@pytest.mark.parametrize("scale", [2.0 ** -4, 2.0 ** 10, 2.0 ** 40])
def test_power_of_two_field_scale_changes_nothing(scale):
    """the predicate is scale-invariant in |u| and the per-record scale is an exact exponent shift: a field multiplied
    by a power of two gives bit-identical inlier counts, in both modes, for every hypothesis that the scaling leaves
    bit-identical (hypothesis generation has absolute 1e-6 determinant gates upstream, kernel.cu:42: a scaled field
    moves pairs across them, nothing else changes -- at 2^-10 every pair of unit vectors would fall below the gate)"""
    mask, planar, _ = _field_case()
    m, v = to_dev(mask, planar)
    ms, vs = to_dev(mask, (planar * np.float32(scale)).astype(np.float32))
    for mode in (dict(), dict(approx=True), dict(literal=True)):
        _, d1 = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=5, return_debug=True, **mode)
        h1, c1 = d1["hyp"].clone(), d1["counts"].clone()
        _, d2 = voting.ransac_voting_layer_v3(ms, vs, 256, inlier_thresh=0.99, seed=5, return_debug=True, **mode)
        same = (d2["hyp"] == h1).all(-1) & (h1 != 0).any(-1)
        assert float(same.float().mean()) > 0.5, float(same.float().mean())
        assert torch.equal(d2["counts"][same], c1[same]), (scale, mode)  # a power-of-two scale is exact in every arithmetic
        assert int(c1[same].max()) > 100


def test_far_hypotheses_and_unnormalised_fields_vote_like_the_reference():
    """near-parallel direction pairs put hypotheses 1e6 .. 1e9 px away; with |u| = 1024 the old fixed 2^90 record scale
    overflowed float32 there (inf - inf = NaN = no vote) while the reference still votes.  Fast counts must stay within
    two votes of literal for every hypothesis, far ones included."""
    h, w, vn = 96, 128, 2
    ys, xs = np.mgrid[0:h, 0:w]
    fg = ((xs - 64) ** 2 + (ys - 48) ** 2) <= 20 ** 2
    kp = np.array([[2.0e6, 48.0], [-3.0e8, 1.0e8]])  # far key-points: all directions of a key-point nearly parallel
    planar = synth.field_from_keypoints(fg, kp)
    mask = fg[None].astype(np.uint8)
    for mul in (1.0, 1024.0):
        m, v = to_dev(mask, (planar[None] * np.float32(mul)).astype(np.float32))
        _, dl = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=9, literal=True, return_debug=True)
        cl, hl = dl["counts"].clone(), dl["hyp"].clone()
        _, de = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=9, return_debug=True)
        assert torch.equal(de["counts"], cl), "exact mode: the reference's integers, far hypotheses included"
        _, df = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=9, approx=True, return_debug=True)
        far = hl.abs().amax(-1) > 1e5
        assert int(far.sum()) > 100  # the case really exercises far hypotheses
        assert torch.isfinite(df["hyp"]).all()
        d = (df["counts"] - cl).abs()
        assert int(d.max()) <= 2, (mul, int(d.max()))
        assert int(cl[far].max()) > 1000  # and they do collect votes in the reference's arithmetic


def test_nan_and_inf_directions_never_vote_and_never_spread():
    mask, planar, _ = _field_case(b=1)
    fgy, fgx = np.nonzero(mask[0])
    bad = planar.copy()
    sel = np.arange(0, len(fgy), 7)
    bad[0, 0, fgy[sel], fgx[sel]] = np.nan          # key-point 0, x component
    bad[0, 3, fgy[sel], fgx[sel]] = np.inf          # key-point 1, y component
    bad[0, 4, fgy[sel[::2]], fgx[sel[::2]]] = -np.inf
    m, v = to_dev(mask, bad)
    rng = np.random.default_rng(3)
    tn = len(fgy)
    good_idx = np.setdiff1d(np.arange(tn), sel)      # draw hypotheses from clean pixels only: finite hypotheses
    idxs = torch.from_numpy(good_idx[rng.integers(0, len(good_idx), (1, 256, 9, 2))].astype(np.int32)).to(dev())
    lit, dl = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, idxs=idxs, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    lit = lit.clone()
    ex, de = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, idxs=idxs, return_debug=True)
    assert torch.equal(de["counts"], cl) and torch.isfinite(ex).all()
    fast, df = voting.ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, idxs=idxs, approx=True, return_debug=True)
    assert torch.isfinite(fast).all() and torch.isfinite(lit).all()
    d = (df["counts"] - cl).abs()
    assert int(d.max()) <= 2
    # the poisoned pixels are exactly the ones that cannot vote: counts of the affected key-points drop by at most them
    mc, planar_c = mask, planar
    mcd, vcd = to_dev(mc, planar_c)
    _, dc = voting.ransac_voting_layer_v3(mcd, vcd, 256, inlier_thresh=0.99, idxs=idxs, approx=True, return_debug=True)
    drop = dc["counts"] - df["counts"]
    assert int(drop[:, 0].max()) <= len(sel) + 2 and int(drop[:, 0].min()) >= -2
    assert int(drop[:, 3:].abs().max()) <= 2          # key-points without poisoned vectors are untouched
    assert float((fast - lit).abs().max()) < 5e-2


def test_v5_confidence_and_debug_directions_with_out_of_range_threshold():
    """thresh outside (0,1) forces literal scoring inside the library; the Python mirror must know (records,
    debug directions and the v5 confidence read the workspace accordingly), and return_* keywords passed through the
    sibling wrappers must not break their return shapes."""
    mask, planar, _ = _field_case(b=2)
    m, v = to_dev(mask, planar)
    vnp = synth.planar_to_vertex_view(planar)
    out, dbg = voting.ransac_voting_layer_v3(m, v, 64, inlier_thresh=1.0, seed=2, return_debug=True)
    assert dbg["literal"] is True  # effective mode
    for bi in range(2):
        coords, direct = O.compact(O.foreground(mask[bi]), vnp[bi])
        tn = coords.shape[0]
        np.testing.assert_array_equal(voting.debug_dir(dbg)[bi, :, :tn].cpu().numpy().transpose(1, 0, 2), direct)
    pts, conf = voting.ransac_voting_layer_v5(m, v, 64, inlier_thresh=0.99, max_num=30000, seed=2, return_status=True,
                                              stage_times=False)
    pts_l, conf_l = voting.ransac_voting_layer_v5(m, v, 64, inlier_thresh=0.99, max_num=30000, seed=2, literal=True)
    assert pts.shape == (2, 9, 2) and conf.shape == (2, 9)
    # the confidence epilogue counts with the literal float32 test on the raw directions whatever mode scored
    ref = O.vote_confidence(mask, vnp, pts.cpu().numpy(), 0.999, np.float32)
    np.testing.assert_allclose(conf.cpu().numpy(), ref, atol=1e-6)
    assert float(conf.min()) > 0.01
    assert float((pts - pts_l).abs().max()) < 5e-2 and float((conf - conf_l).abs().max()) < 2e-2


@pytest.mark.parametrize("hpl,chunk", [(2, 64), (2, 128), (4, 64), (4, 128), (8, 64), (8, 128)])
def test_every_matrix_pipe_tiling_counts_like_literal(hpl, chunk, monkeypatch):
    """stress MH = 1/2/4/8 hypothesis tiles per wave and 2..8 pixel tiles per item (the hand-placed vote
    epilogue reads MFMA results from inline asm): counts of every tiling equal the default tiling's bit for bit and stay
    within two votes of literal."""
    mask, planar, _ = synth.make_batch(3, first_index=5000, h=200, w=280, radius=31, noise=True, background="normal")
    m, v = to_dev(mask, planar)
    _, dl = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, literal=True, return_debug=True)
    cl = dl["counts"].clone()
    _, d0 = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, approx=True, return_debug=True)
    c0 = d0["counts"].clone()
    monkeypatch.setenv("PVNET_SCORE_HPL", str(hpl))
    monkeypatch.setenv("PVNET_SCORE_CHUNK", str(chunk))
    voting.reload_tuning()
    try:
        for fold in ("0", "1"):  # exact mode, both cell sizes: the reference's integers in every tiling
            monkeypatch.setenv("PVNET_EXACT_FOLD", fold)
            voting.reload_tuning()
            _, de = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, return_debug=True)
            assert torch.equal(de["counts"], cl), (hpl, chunk, fold)
        monkeypatch.delenv("PVNET_EXACT_FOLD")
        voting.reload_tuning()
        _, d = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, approx=True, return_debug=True)
        c = d["counts"].clone()
        mh = d["layout"].wg_g * d["layout"].hpl // 2
    finally:
        monkeypatch.delenv("PVNET_SCORE_HPL")
        monkeypatch.delenv("PVNET_SCORE_CHUNK")
        monkeypatch.delenv("PVNET_EXACT_FOLD", raising=False)
        voting.reload_tuning()
    assert mh in (1, 2, 4, 8)
    assert torch.equal(c, c0)
    assert int((c - cl).abs().max()) <= 2


def test_vote_plan_matches_the_plain_call():
    mask, planar, _ = _field_case(b=1, h=480, w=640, radius=27)
    m, v = to_dev(mask.astype(np.int64), planar)
    plan = voting.VotePlan(m, v, 512, inlier_thresh=0.99)
    for seed in (1, 2):
        a = plan(m, v, seed=seed).clone()
        b = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=seed)
        assert torch.equal(a, b)
    ws = torch.empty(plan.layout.total_bytes, dtype=torch.uint8, device=dev())
    c = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=2, workspace=ws)
    assert torch.equal(c, b)
    slot = torch.zeros((3, 1, 9, 2), device=dev())  # out=: vote straight into a slot of a staging block
    d = voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=2, out=slot[1])
    assert d.data_ptr() == slot[1].data_ptr() and torch.equal(slot[1], b) and (slot[0] == 0).all() and (slot[2] == 0).all()
    with pytest.raises(RuntimeError, match="out must be"):
        voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=2, out=slot[:, 0])
    with pytest.raises(RuntimeError, match="workspace"):
        voting.ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, seed=2, workspace=ws[:1000])
    with pytest.raises(RuntimeError, match="VotePlan"):
        plan(m[:, :100], v[:, :100])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_half_precision_fields_and_logits_are_read_in_place(dt):
    """a backbone under autocast emits bf16 / fp16: the field (and the class logits of the fused arg-max entry) are
    widened where they are read -- bit-identical to the float32 path on `.float()`, without the 786 MB copy"""
    mask, planar, _ = synth.make_batch(3, first_index=8100, h=120, w=160, radius=18, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev())
    p16 = torch.from_numpy(planar).to(dev()).to(dt)
    v16 = synth.planar_to_vertex_view(p16)
    v32 = synth.planar_to_vertex_view(p16.float())
    assert v16.dtype == dt and not v16.is_contiguous()
    for literal in (False, True):
        ref, dr = voting.ransac_voting_layer_v3(m, v32, 256, inlier_thresh=0.99, seed=3, literal=literal, return_debug=True)
        ref, cr = ref.clone(), dr["counts"].clone()
        out, do = voting.ransac_voting_layer_v3(m, v16, 256, inlier_thresh=0.99, seed=3, literal=literal, return_debug=True)
        assert torch.equal(out, ref) and torch.equal(do["counts"], cr)
    seg = torch.stack([1.0 - m.float(), m.float()], 1).contiguous()
    seg = (seg + 0.05 * torch.randn(seg.shape, device=dev())).to(dt)
    a = voting.ransac_voting_layer_v3(torch.argmax(seg.float(), 1), v32, 256, inlier_thresh=0.99, seed=4)
    b = voting.ransac_voting_layer_v3_from_logits(seg, v16, 256, inlier_thresh=0.99, seed=4)
    assert torch.equal(a, b)
    plan = voting.VotePlan(m, v16, 256, inlier_thresh=0.99)
    assert torch.equal(plan(m, v16, seed=3), ref if False else voting.ransac_voting_layer_v3(m, v32, 256, inlier_thresh=0.99, seed=3))
    mo = voting.ransac_motion_voting(m, v16)
    assert torch.equal(mo, voting.ransac_motion_voting(m, v32))


@pytest.mark.parametrize("kg", ["1", "3", "9"])
def test_compaction_is_deterministic_for_every_tiling(kg, monkeypatch):
    """regression: one build of compact_kernel<false, 1> returned, in 40-100 % of the calls, up to 64 records of one wave
    from pixels a few ranks away -- tied to "VGPR allocation used to its last granule + a second workgroup on the CU"
    (profiles/r02_compaction_flake_investigation.txt); every kernel keeps a spare granule since and the rank search is
    branch-free.  80 repeated calls per tiling must reproduce the records, hypotheses and counts of the first bit for bit
    (3 images x 9 key-points at tiling 1 = 378 workgroups: more than one per CU, the condition that failed)."""
    mask, planar, _ = synth.make_batch(3, first_index=1300, h=200, w=280, radius=31, noise=True, background="normal")
    m, v = to_dev(mask, planar)
    monkeypatch.setenv("PVNET_COMPACT_KG", "3")
    voting.reload_tuning()
    try:
        _, d = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, return_debug=True)
        tn = [int(x) for x in d["tn"]]
        ref = (d["rec"].clone(), d["hyp"].clone(), d["counts"].clone())
        monkeypatch.setenv("PVNET_COMPACT_KG", kg)
        voting.reload_tuning()
        for rep in range(80):
            _, g = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, return_debug=True)
            for bi in range(3):
                assert torch.equal(g["rec"][bi, :, :tn[bi]], ref[0][bi, :, :tn[bi]]), (kg, rep, bi)
            assert torch.equal(g["hyp"], ref[1]) and torch.equal(g["counts"], ref[2]), (kg, rep)
    finally:
        monkeypatch.delenv("PVNET_COMPACT_KG")
        voting.reload_tuning()
