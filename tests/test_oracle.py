"""Pins the CPU oracle (oracle/) -- the checker every GPU parity test relies on.

The reference ships no tests or golden vectors and cannot run here, so the pins are:
G1 the reference's demo fixture with its analytic answer, G4 hand-computed op-level cases, the committed
float64 outputs on seeded noisy inputs (G3), and bit-exact agreement of two independent restatements
(numpy float32 vs plain C)."""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _demo_inputs(demo_fixture):
    mask = demo_fixture["mask"]
    planar = synth.field_from_keypoints(mask.astype(bool), demo_fixture["points_2d"])
    return mask[None].astype(np.int64), synth.planar_to_vertex_view(planar[None])


# ---------------------------------------------------------------- G1: demo fixture known answer
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_demo_fixture_known_answer(demo_fixture, dtype):
    mask, vertex = _demo_inputs(demo_fixture)
    assert int(mask.sum()) == 2289  # SURVEY finding 6
    out = O.ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99, dtype=dtype)  # demo.py:55
    assert np.abs(out[0] - demo_fixture["points_2d"]).max() < 1e-4


def test_demo_fixture_rng_independent(demo_fixture):
    """on a clean field every non-degenerate hypothesis ties at count == tn; output must not depend on idxs."""
    mask, vertex = _demo_inputs(demo_fixture)
    a, da = O.ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99, seed=1, return_debug=True)
    b = O.ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99, seed=2)
    assert np.abs(a - b).max() < 1e-5
    assert (da[0]["win_cnt"] == 2289).all()


def test_demo_fixture_c_oracle(demo_fixture):
    mask, vertex = _demo_inputs(demo_fixture)
    out, wi, wc = cref.vote_v3(mask != 0, vertex, 512, 0.99, return_winners=True)
    assert np.abs(out[0] - demo_fixture["points_2d"]).max() < 1e-4
    assert (wc == 2289).all()


def test_reference_order_fp32_lsq_is_the_noise_floor(demo_fixture):
    """the reference's own un-centred float32 LSQ (ransac_voting_gpu.py:591-594) sits ~1e-3 px from float64."""
    mask, vertex = _demo_inputs(demo_fixture)
    o64 = O.ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99)
    o32 = O.ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99, dtype=np.float32, lsq_dtype=np.float32)
    d = np.abs(o32 - o64).max()
    assert 1e-6 < d < 2e-2


# ---------------------------------------------------------------- G4: op-level hand-computed cases
def test_hypothesis_hand_cases():
    # pixel 0 at (0,0) pointing along +x, pixel 1 at (4,-3) pointing along +y  -> lines y=0 and x=4 -> (4,0)
    coords = np.array([[0, 0], [4, -3], [10, 0]], np.float32)
    direct = np.array([[[1, 0]], [[0, 1]], [[2, 0]]], np.float32)  # [tn=3, vn=1, 2]
    idxs = np.array([[[0, 1]], [[1, 0]], [[0, 2]], [[1, 1]]], np.int32)  # last two: parallel / same pixel
    for fn in (lambda: O.generate_hypothesis(direct, coords, idxs, np.float64),
               lambda: O.generate_hypothesis(direct, coords, idxs, np.float32),
               lambda: cref.generate_hypothesis(direct, coords, idxs)):
        hyp = fn()
        np.testing.assert_allclose(hyp[0, 0], [4, 0], atol=1e-6)
        np.testing.assert_allclose(hyp[1, 0], [4, 0], atol=1e-6)
        np.testing.assert_array_equal(hyp[2, 0], [0, 0])  # parallel -> zeros (kernel.cu:42-43 + :75)
        np.testing.assert_array_equal(hyp[3, 0], [0, 0])  # t0 == t1 -> zeros


def test_voting_threshold_edge_and_skips():
    # hypothesis at origin; pixels on a circle of radius 10; direction of pixel i points at angle a_i off the
    # exact direction to the hypothesis -> cos = cos(a_i)
    angs = np.deg2rad([0.0, 5.0, 8.0, 8.2, 20.0, 180.0])
    px = np.array([[10, 0]] * len(angs), np.float32)
    to_h = np.array([-1.0, 0.0])
    direct = np.stack([np.cos(angs) * to_h[0] - np.sin(angs) * to_h[1],
                       np.sin(angs) * to_h[0] + np.cos(angs) * to_h[1]], 1).astype(np.float32)[:, None, :]
    hyp = np.zeros((1, 1, 2), np.float32)
    thresh = 0.99  # acos(0.99) = 8.11 deg
    expect = np.array([1, 1, 1, 0, 0, 0], np.uint8)
    for dt in (np.float64, np.float32):
        inl = O.voting_for_hypothesis(direct, px, hyp, thresh, dt)
        np.testing.assert_array_equal(inl[0, 0], expect)
    buf = np.zeros((1, 1, len(angs)), np.uint8)
    buf[0, 0, 5] = 7  # the op only ever SETS ones (kernel.cu:124-125): foreign values survive
    cref.voting_for_hypothesis(direct, px, hyp, buf, thresh)
    np.testing.assert_array_equal(buf[0, 0], [1, 1, 1, 0, 0, 7])
    # zero direction (norm1 < 1e-6) and hypothesis on the pixel (norm2 < 1e-6) are skipped (kernel.cu:121)
    direct0 = np.array([[[0, 0]], [[1, 0]]], np.float32)
    coords0 = np.array([[5, 5], [0, 0]], np.float32)
    for dt in (np.float64, np.float32):
        assert O.voting_counts(direct0, coords0, hyp, -2.0, dt)[0, 0] == 0
    assert cref.voting_counts(direct0, coords0, hyp, -2.0)[0, 0] == 0


def _vp_case(seed, tn=300, vn=4, hn=200):
    """random rays plus every special case of the two vanishing-point kernels"""
    rng = np.random.default_rng(seed)
    coords = rng.integers(0, 200, (tn, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (tn, vn))
    direct = (np.stack([np.cos(ang), np.sin(ang)], -1) * rng.uniform(0.5, 2.0, (tn, vn, 1))).astype(np.float32)
    idxs = rng.integers(0, tn, (hn, vn, 2)).astype(np.int32)
    direct[3] = direct[5]            # parallel rays at different pixels: z = 0 exactly
    idxs[0, :] = [3, 5]
    idxs[1, :] = [7, 7]              # the same pixel twice: the zero vector
    direct[9] = 0.0                  # a zero direction: zero line coordinates
    idxs[2, :] = [9, 11]
    direct[13, :, 0] = 0.0           # axis-aligned rays: val_x == 0 exactly (neither < 0 nor a sign conflict)
    idxs[3, :] = [13, 15]
    return coords, direct, idxs


def test_vanishing_point_hand_cases():
    """ransac_voting_kernel.cu:170-229, :268-310 on rays whose intersection is known"""
    s = np.float32(np.sqrt(0.5))
    coords = np.array([[0, 0], [10, 0], [0, 4], [10, 10]], np.float32)
    direct = np.array([[[s, s]], [[-s, s]], [[s, s]], [[s, s]]], np.float32)  # [tn=4, vn=1, 2]
    idxs = np.array([[[0, 1]],    # rays from (0,0) along (1,1) and from (10,0) along (-1,1) meet at (5,5)
                     [[0, 2]],    # parallel (same direction, different pixels): a point at infinity, z = 0
                     [[0, 3]],    # (0,0) and (10,10), both along (1,1): one and the same line
                     [[1, 1]]], np.int32)
    for hyp in (O.generate_hypothesis_vanishing_point(direct, coords, idxs, np.float64),
                O.generate_hypothesis_vanishing_point(direct, coords, idxs, np.float32),
                cref.generate_hypothesis_vanishing_point(direct, coords, idxs)):
        assert hyp.shape == (4, 1, 3)
        x, y, z = hyp[0, 0]
        assert z != 0 and abs(x / z - 5) < 1e-5 and abs(y / z - 5) < 1e-5
        assert hyp[1, 0, 2] == 0 and np.abs(hyp[1, 0, :2]).max() > 0      # direction of the common vanishing point
        np.testing.assert_array_equal(hyp[3, 0], [0, 0, 0])
        # the pixels that produced hypothesis 0 vote for it; pixel 3 lies on the line through it (|cos| = 1) but looks
        # away from it: the direction is wrong (:306)
        inl = O.voting_for_hypothesis_vanishing_point(direct, coords, hyp[:1].astype(np.float64), 0.99)
        np.testing.assert_array_equal(inl[0, 0], [1, 1, 0, 0])
        buf = np.zeros((1, 1, 4), np.uint8)
        buf[0, 0, 3] = 9                                                   # the op only ever sets ones
        cref.voting_for_hypothesis_vanishing_point(direct, coords, hyp[:1].astype(np.float32), buf, 0.99)
        np.testing.assert_array_equal(buf[0, 0], [1, 1, 0, 9])
    # where the affine op (pinned by the reference's device code) has an answer, x/z, y/z is that answer
    coords, direct, idxs = _vp_case(1)
    vp = O.generate_hypothesis_vanishing_point(direct, coords, idxs, np.float64)
    aff = O.generate_hypothesis(direct, coords, idxs, np.float64)
    ok = (np.abs(vp[..., 2]) > 1e-3) & (np.abs(aff).sum(-1) > 0)
    assert ok.mean() > 0.3
    assert np.abs(vp[ok][:, :2] / vp[ok][:, 2:3] - aff[ok]).max() < 1e-6 * max(1.0, np.abs(aff[ok]).max())


@pytest.mark.parametrize("seed", [2, 5])
def test_vanishing_point_numpy_f32_equals_c_bit_exact(seed):
    coords, direct, idxs = _vp_case(seed)
    hyp_np = O.generate_hypothesis_vanishing_point(direct, coords, idxs, np.float32)
    hyp_c = cref.generate_hypothesis_vanishing_point(direct, coords, idxs)
    assert hyp_np.dtype == np.float32 and hyp_np.tobytes() == hyp_c.tobytes()          # signed zeros included
    assert (hyp_c[0, :, 2] == 0).all() and not hyp_c[1].any() and not hyp_c[2].any()
    assert 0.2 < (np.abs(hyp_c).sum(-1) == 0).mean() < 0.95                           # rays that do not meet: zeros
    hyp_c[4, :, :] = [coords[6, 0] * 2, coords[6, 1] * 2, 2]                           # a hypothesis ON pixel 6: norm2 = 0
    for thresh in (0.9, 0.999):
        inl_np = O.voting_for_hypothesis_vanishing_point(direct, coords, hyp_c, thresh, np.float32)
        inl_c = np.zeros_like(inl_np)
        cref.voting_for_hypothesis_vanishing_point(direct, coords, hyp_c, inl_c, thresh)
        np.testing.assert_array_equal(inl_np, inl_c)
        assert inl_c[:, :, 9].sum() == 0 and inl_c[4, :, 6].sum() == 0 and inl_c[1].sum() == 0
        assert 0 < inl_c.sum() < inl_c.size // 4


# ---------------------------------------------------------------- two restatements, bit for bit
@pytest.mark.parametrize("seed", [0, 7])
def test_numpy_f32_equals_c_bit_exact(seed):
    mask, planar, _ = synth.make_batch(2, first_index=300 + seed, h=120, w=160, radius=12, background="normal",
                                       noise=True)
    vertex = synth.planar_to_vertex_view(planar)
    out, dbg = O.ransac_voting_layer_v3(mask, vertex, 96, inlier_thresh=0.99, seed=seed, dtype=np.float32,
                                        return_debug=True)
    for bi, d in enumerate(dbg):
        hyp_c = cref.generate_hypothesis(d["direct"], d["coords"], d["idxs"])
        assert hyp_c.tobytes() == d["hyp"].astype(np.float32).tobytes()
        cnt_c = cref.voting_counts(d["direct"], d["coords"], hyp_c, 0.99)
        np.testing.assert_array_equal(cnt_c, d["counts"])
    outc, wi, wc = cref.vote_v3(mask != 0, vertex, 96, 0.99, seed=seed, return_winners=True)
    np.testing.assert_array_equal(wi, np.stack([d["win_idx"] for d in dbg]))
    np.testing.assert_array_equal(wc, np.stack([d["win_cnt"] for d in dbg]))
    assert np.abs(outc - out).max() < 1e-4


def test_rng_restatements_agree():
    r = O.rng_u32(0xDEADBEEF12345678, O.TAG_SUB, np.uint64(5), np.arange(64, dtype=np.uint64))
    c = [cref.lib().ref_rng_u32(0xDEADBEEF12345678, O.TAG_SUB, 5, i) for i in range(64)]
    np.testing.assert_array_equal(r, np.array(c, np.uint32))
    ix = O.draw_idxs(3, 1, 1000, 9, 777)
    assert ix.min() >= 0 and ix.max() < 777 and abs(ix.mean() - 388) < 15


# ---------------------------------------------------------------- driver semantics (Appendix A of SURVEY.md)
def test_degenerate_loop_is_idempotent():
    """finding 3: idxs is drawn once outside the while-loop, so extra rounds never change the result."""
    mask, planar, _ = synth.make_batch(1, first_index=11, h=120, w=160, radius=12, noise=True)
    vertex = synth.planar_to_vertex_view(planar)
    a, da = O.ransac_voting_layer_v3(mask, vertex, 32, inlier_thresh=0.999, seed=5, return_debug=True)
    b, db = O.ransac_voting_layer_v3(mask, vertex, 32, inlier_thresh=0.999, seed=5, emulate_rounds=True,
                                     confidence=1.0, max_iter=3, return_debug=True)
    assert db[0]["rounds"] > 1
    np.testing.assert_array_equal(a, b)


def test_min_num_and_max_num_gates():
    mask, planar, _ = synth.make_batch(2, first_index=20, h=120, w=160, radius=10)
    vertex = synth.planar_to_vertex_view(planar)
    mask[1] = 0
    mask[1, 5, 5:8] = 1  # 3 px < min_num=5 -> zeros (ransac_voting_gpu.py:531-534)
    out, dbg = O.ransac_voting_layer_v3(mask, vertex, 32, inlier_thresh=0.99, return_debug=True)
    assert dbg[1]["skipped"] and (out[1] == 0).all() and not dbg[0]["skipped"]
    tn0 = dbg[0]["tn0"]
    out2, dbg2 = O.ransac_voting_layer_v3(mask, vertex, 32, inlier_thresh=0.99, max_num=100, seed=9,
                                          return_debug=True)
    assert 60 < dbg2[0]["tn"] < 140 < tn0  # Bernoulli(max_num/tn0) subsample (:537-540)
    outc = cref.vote_v3(O.foreground(mask), vertex, 32, 0.99, max_num=100, seed=9)
    assert np.abs(outc - out2).max() < 1e-4  # same counter RNG in C and numpy


def test_mask_byte_semantics():
    m = np.array([[0, 1, 256, 257, -1]], np.int64)
    np.testing.assert_array_equal(O.foreground(m), [[False, True, False, True, True]])  # .byte() wraps mod 256
    np.testing.assert_array_equal(O.foreground(np.array([[0.0, 0.9, 1.0, 2.5]], np.float32)),
                                  [[False, False, True, True]])


def test_clean_synthetic_recovers_keypoints():
    mask, planar, kpts = synth.make_batch(2, first_index=40, h=240, w=320, radius=25, background="normal")
    vertex = synth.planar_to_vertex_view(planar)
    out = O.ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99)
    assert np.abs(out - kpts).max() < 2e-3  # G2: float32 field quantisation only


def test_committed_noisy_golden_outputs():
    g = np.load(os.path.join(ROOT, "tests", "golden", "noisy_oracle.npz"))
    for i in range(int(g["ncases"])):
        mask, planar, _ = synth.make_batch(2, first_index=int(g[f"c{i}_first_index"]), h=240, w=320,
                                           radius=int(g[f"c{i}_radius"]), background="normal", noise=True)
        vertex = synth.planar_to_vertex_view(planar)
        out, dbg = O.ransac_voting_layer_v3(mask, vertex, int(g[f"c{i}_hn"]), inlier_thresh=float(g[f"c{i}_thresh"]),
                                            seed=int(g[f"c{i}_seed"]), return_debug=True)
        np.testing.assert_array_equal(np.stack([d["win_idx"] for d in dbg]), g[f"c{i}_win_idx"])
        np.testing.assert_array_equal(np.stack([d["win_cnt"] for d in dbg]), g[f"c{i}_win_cnt"])
        np.testing.assert_allclose(out, g[f"c{i}_out"], atol=1e-5)


def test_motion_voting():
    mask = np.zeros((1, 4, 5), np.int64)
    mask[0, 1, 2] = 1
    mask[0, 3, 4] = 1
    vertex = np.zeros((1, 4, 5, 2, 2), np.float32)
    vertex[0, 1, 2, 0] = [1.0, 2.0]
    vertex[0, 3, 4, 0] = [3.0, -2.0]
    out = O.ransac_motion_voting(mask, vertex)
    np.testing.assert_allclose(out[0, 0], [(2 + 1 + 4 + 3) / 2, (1 + 2 + 3 - 2) / 2])
    np.testing.assert_allclose(out[0, 1], [3.0, 2.0])


def test_thinning_table_header_and_oracles_agree(tmp_path):
    """round 4: the keep rule of `max_num` (ransac_voting_gpu.py:537-540) is a table of 1 424 bins of the random word.  The header
    the kernels include (pvnet_amd/csrc/pvnet_rng.h, compiled here for the host), the numpy oracle and -- through the thinned
    pixel counts of vote_v3 -- the C oracle must state the same table; the table must be monotone, and the kept probability
    must lie in [p, min(p + 1/1024, p 17/16)] for p = max_num / tn0."""
    import ctypes
    import subprocess
    src = tmp_path / "thin.c"
    src.write_text('#include "pvnet_rng.h"\n'
                   "int t_bin(unsigned r) { return pvnet_thin_bin(r); }\n"
                   "int t_kept(long long m, long long n) { return pvnet_thin_bins_kept(m, n); }\n"
                   "int t_last(void) { return PVNET_THIN_LAST; }\n")
    lib = tmp_path / "thin.so"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-I", os.path.join(root, "pvnet_amd", "csrc"), str(src), "-o", str(lib)])
    L = ctypes.CDLL(str(lib))
    L.t_kept.argtypes = [ctypes.c_longlong, ctypes.c_longlong]
    assert L.t_last() == O.THIN_LAST == 1423
    rng = np.random.default_rng(7)
    words = np.concatenate([np.arange(0, 70), (1 << np.arange(0, 32)).astype(np.uint64), (1 << np.arange(1, 33)).astype(np.uint64) - 1,
                            rng.integers(0, 1 << 32, 3000, dtype=np.uint64), rng.integers(0, 1 << 26, 3000, dtype=np.uint64),
                            rng.integers(0, 1 << 12, 300, dtype=np.uint64)])
    words = np.unique(words)
    bins = [L.t_bin(int(r)) for r in words]
    assert bins == [O.thin_bin(int(r)) for r in words]
    assert all(a <= b for a, b in zip(bins, bins[1:])) and bins[0] == 0 and bins[-1] == O.THIN_LAST   # monotone, 0 .. 1423
    for max_num, tn0 in [(100, 60000), (100, 101), (30000, 35000), (30000, 307200), (1, 307200), (0, 500), (7, 9000), (150, 151),
                         (5, 1 << 31), (4999, 5000)] + [tuple(sorted(rng.integers(1, 400000, 2))) for _ in range(200)]:
        max_num, tn0 = int(max_num), int(tn0)
        if tn0 <= max_num:
            continue
        k = L.t_kept(max_num, tn0)
        thr = O.subsample_threshold(max_num, tn0)
        if k > O.THIN_LAST:
            assert thr == 1 << 32                                        # every bin kept
        else:                                                            # the threshold is the FIRST word of bin k
            assert O.thin_bin(thr) >= k and (thr == 0 or O.thin_bin(thr - 1) < k)
        p, kept = max_num / tn0, thr / 2.0 ** 32
        assert p <= kept <= min(p + 1 / 1024, p * 17 / 16) + 1e-12, (max_num, tn0, p, kept)
    # the C oracle thins with the same rule: same winners and counts as the numpy float32 restatement on thinned images
    mask, planar, _ = synth.make_batch(2, first_index=77, h=120, w=160, radius=45, noise=True)
    v = synth.planar_to_vertex_view(planar)
    for max_num in (3000, 150, 7):
        _, dbg = O.ransac_voting_layer_v3(mask, v, 32, inlier_thresh=0.99, max_num=max_num, seed=11, dtype=np.float32,
                                          return_debug=True)
        assert all(d["tn"] < d["tn0"] for d in dbg)
        _, wi, wc = cref.vote_v3(mask != 0, v, 32, 0.99, max_num=max_num, seed=11, return_winners=True)
        np.testing.assert_array_equal(wi, np.stack([d["win_idx"] for d in dbg]))
        np.testing.assert_array_equal(wc, np.stack([d["win_cnt"] for d in dbg]))
