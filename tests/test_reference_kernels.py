"""GPU parity tests against the REFERENCE'S OWN device code: lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu
(generate_hypothesis_kernel :11-49, voting_for_hypothesis_kernel :88-126, the vanishing-point pair :170-229, :268-310) compiled for gfx950 from the reference
tree by `make -C oracle ref` (oracle/_ref/, git-ignored, travels with the tree to the GPU box) and run on the
MI355X next to our kernels on identical device buffers.

This pins the chain: reference kernels == plain-C restatement == numpy float32 oracle == HIP literal mode, all
bit-exact; the default fast mode is then measured against the reference kernels' inlier sets."""
import os

import numpy as np
import pytest
import torch

from oracle import cref, refkernels
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth, voting

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refkernels.available("off"),
                                 reason="oracle/_ref not built (needs the reference tree: make -C oracle ref)")]


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def compacted(first=600, h=120, w=160, radius=16, noise=True):
    mask, planar, kpts = synth.make_batch(1, first_index=first, h=h, w=w, radius=radius, noise=noise,
                                          background="normal")
    vnp = synth.planar_to_vertex_view(planar)
    coords, direct = O.compact(O.foreground(mask[0]), vnp[0])
    return mask, planar, coords.astype(np.float32), np.ascontiguousarray(direct, np.float32)


def test_reference_kernels_equal_c_restatement_and_our_ops():
    _, _, coords, direct = compacted()
    tn = coords.shape[0]
    hn = 96
    idxs = np.random.default_rng(3).integers(0, tn, (hn, 9, 2), dtype=np.int32)
    idxs[0, 0] = [5, 5]           # the same pixel twice: the kernel returns early, the zero-filled output stays (:42-43)
    idxs[1, 1] = [7, 7]
    direct[9, 2] = direct[11, 2]  # parallel directions at different pixels: determinant exactly zero
    idxs[2, 2] = [9, 11]
    d, c, i = (torch.from_numpy(x).to(dev()) for x in (direct, coords, idxs))
    hyp_ref = refkernels.generate_hypothesis(d, c, i)
    assert hyp_ref[0, 0].abs().sum() == 0 and hyp_ref[2, 2].abs().sum() == 0
    hyp_c = cref.generate_hypothesis(direct, coords, idxs)
    assert hyp_ref.cpu().numpy().tobytes() == hyp_c.tobytes(), "C restatement != reference device code"
    assert voting.generate_hypothesis(d, c, i).cpu().numpy().tobytes() == hyp_c.tobytes()
    for thresh in (0.99, 0.999):
        inl_ref = refkernels.voting_for_hypothesis(d, c, hyp_ref, thresh)
        inl_c = np.zeros((hn, 9, tn), np.uint8)
        cref.voting_for_hypothesis(direct, coords, hyp_c, inl_c, thresh)
        np.testing.assert_array_equal(inl_ref.cpu().numpy(), inl_c)
        ours = torch.zeros((hn, 9, tn), dtype=torch.uint8, device=dev())
        voting.voting_for_hypothesis(d, c, hyp_ref, ours, thresh)
        assert torch.equal(ours, inl_ref)
        assert 0 < int(inl_ref.sum()) < hn * 9 * tn
    # degenerate inputs of the voting kernel: zero direction (norm1 < 1e-6) and hypothesis on the pixel (norm2 < 1e-6)
    direct2 = direct.copy()
    direct2[4] = 0.0
    hyp2 = hyp_c.copy()
    hyp2[3, :, :] = coords[6]
    d2, h2 = torch.from_numpy(direct2).to(dev()), torch.from_numpy(hyp2).to(dev())
    inl_ref = refkernels.voting_for_hypothesis(d2, c, h2, 0.99)
    assert int(inl_ref[:, :, 4].sum()) == 0 and int(inl_ref[3, :, 6].sum()) == 0
    ours = torch.zeros_like(inl_ref)
    voting.voting_for_hypothesis(d2, c, h2, ours, 0.99)
    assert torch.equal(ours, inl_ref)


def test_vanishing_point_ops_equal_reference_device_code():
    """the other two ops of the reference's extension (ransac_voting.cpp:57-99; kernels :170-229, :268-310): our HIP
    ops == the reference's own device code == the plain-C restatement, bit for bit (signed zeros included), on real
    compacted pixels plus every special case: parallel rays (z = 0), the same pixel twice, a zero direction,
    axis-aligned rays, rays that do not meet (zeros), a hypothesis on a pixel, pre-set bytes of the in/out tensor."""
    _, _, coords, direct = compacted(first=640, h=140, w=180, radius=18)
    tn = coords.shape[0]
    hn = 160
    idxs = np.random.default_rng(4).integers(0, tn, (hn, 9, 2), dtype=np.int32)
    direct[3] = direct[5]
    idxs[0, :] = [3, 5]
    idxs[1, :] = [7, 7]
    direct[9] = 0.0
    idxs[2, :] = [9, 11]
    direct[13, :, 0] = 0.0
    idxs[3, :] = [13, 15]
    d, c, i = (torch.from_numpy(x).to(dev()) for x in (direct, coords, idxs))
    hyp_ref = refkernels.generate_hypothesis_vanishing_point(d, c, i)
    hyp_c = cref.generate_hypothesis_vanishing_point(direct, coords, idxs)
    assert hyp_ref.shape == (hn, 9, 3)
    assert hyp_ref.cpu().numpy().tobytes() == hyp_c.tobytes(), "C restatement != reference device code"
    ours = voting.generate_hypothesis_vanishing_point(d, c, i)
    assert ours.cpu().numpy().tobytes() == hyp_c.tobytes()
    assert (hyp_c[0, :, 2] == 0).all() and not hyp_c[1].any() and not hyp_c[2].any()
    zero = (np.abs(hyp_c).sum(-1) == 0).mean()
    assert 0.05 < zero < 0.95                                   # some rays meet, some do not
    hyp2 = hyp_c.copy()
    hyp2[4, :, :] = [coords[6, 0] * 2, coords[6, 1] * 2, 2]     # a hypothesis on pixel 6 (norm2 < 1e-6)
    h2 = torch.from_numpy(hyp2).to(dev())
    for thresh in (0.9, 0.99, 0.999):
        inl_ref = refkernels.voting_for_hypothesis_vanishing_point(d, c, h2, thresh)
        inl_c = np.zeros((hn, 9, tn), np.uint8)
        cref.voting_for_hypothesis_vanishing_point(direct, coords, hyp2, inl_c, thresh)
        np.testing.assert_array_equal(inl_ref.cpu().numpy(), inl_c)
        mine = torch.zeros((hn, 9, tn), dtype=torch.uint8, device=dev())
        mine[5, 2, 17] = 7                                       # in/out: foreign bytes survive, ones are only ever set
        voting.voting_for_hypothesis_vanishing_point(d, c, h2, mine, thresh)
        want = inl_ref.clone()
        want[5, 2, 17] = 1 if int(inl_ref[5, 2, 17]) else 7
        assert torch.equal(mine, want)
        assert 0 < int(inl_ref.sum()) < hn * 9 * tn              # (a coherent field: most pixels vote for most hypotheses)
        assert int(inl_ref[:, :, 9].sum()) == 0 and int(inl_ref[4, :, 6].sum()) == 0
    # the reference's argument checks, mirrored (ransac_voting.cpp:66-68, :91-94: CHECK_INPUT)
    with pytest.raises(RuntimeError):
        voting.generate_hypothesis_vanishing_point(d.cpu(), c, i)
    with pytest.raises(RuntimeError):
        voting.voting_for_hypothesis_vanishing_point(d, c, h2[:, :, :2], mine, 0.99)
    import lib.ransac_voting_gpu_layer.ransac_voting as ext      # the module name the reference's driver imports (:2):
    # the compiled extension (pvnet_amd/csrc/ransac_voting_ext.cpp) when it was built, else the Python stand-in
    got = ext.generate_hypothesis_vanishing_point(d, c, i)
    assert got.cpu().numpy().tobytes() == hyp_c.tobytes()
    again = torch.zeros((hn, 9, tn), dtype=torch.uint8, device=dev())
    ext.voting_for_hypothesis_vanishing_point(d, c, h2, again, 0.99)
    assert torch.equal(again, refkernels.voting_for_hypothesis_vanishing_point(d, c, h2, 0.99))


def test_compiled_extension_module_equals_reference_device_code():
    """the reference's plugin is a COMPILED module (`ransac_voting`, src/ransac_voting.cpp: four pybind functions); so is
    ours (pvnet_amd/csrc/ransac_voting_ext.cpp on libpvnet_vote.so, built into the directory the reference's driver imports
    it from).  Same calls, same in/out conventions, same CHECK_INPUT errors -- and bit-equal results with the reference's own
    device code on the MI355X."""
    from pvnet_amd import build as B
    import importlib
    if not os.path.exists(B.ext_path()):
        pytest.skip("compiled ransac_voting module not built (python -m pvnet_amd.build)")
    ext = importlib.import_module("lib.ransac_voting_gpu_layer.ransac_voting")
    assert ext.__file__ == B.ext_path() and "pvnet_amd" in ext.backend      # the extension module shadows the .py stand-in
    _, _, coords, direct = compacted(first=655, h=130, w=170, radius=17)
    tn = coords.shape[0]
    hn = 80
    idxs = np.random.default_rng(8).integers(0, tn, (hn, 9, 2), dtype=np.int32)
    d, c, i = (torch.from_numpy(x).to(dev()) for x in (direct, coords, idxs))
    hyp = ext.generate_hypothesis(d, c, i)                                  # ransac_voting_gpu.py:554
    assert hyp.dtype == torch.float32 and hyp.shape == (hn, 9, 2)
    assert torch.equal(hyp, refkernels.generate_hypothesis(d, c, i))
    inl = torch.zeros((hn, 9, tn), dtype=torch.uint8, device=dev())         # :555-556: zeros, then the op sets ones
    inl[3, 1, 5] = 9
    assert ext.voting_for_hypothesis(d, c, hyp, inl, 0.99) is None
    want = refkernels.voting_for_hypothesis(d, c, hyp, 0.99)
    want[3, 1, 5] = 1 if int(want[3, 1, 5]) else 9
    assert torch.equal(inl, want)
    # on a side stream (the reference launches on the legacy default stream; this module uses the current one)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        hyp2 = ext.generate_hypothesis(d, c, i)
    st.synchronize()
    assert torch.equal(hyp2, hyp)
    # CHECK_INPUT (ransac_voting.cpp:7-9) and the dtype / shape checks
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ext.generate_hypothesis(d.cpu(), c, i)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        ext.generate_hypothesis(d.transpose(0, 1).contiguous().transpose(0, 1), c, i)
    with pytest.raises(RuntimeError, match="int32"):
        ext.generate_hypothesis(d, c, i.long())
    with pytest.raises(RuntimeError):
        ext.voting_for_hypothesis(d, c, hyp, inl[:, :, :-1].contiguous(), 0.99)


@pytest.mark.parametrize("hn,thresh", [(128, 0.99), (256, 0.999)])
def test_literal_path_equals_reference_device_code(hn, thresh):
    """Whole batched path in literal mode vs the reference kernels applied to the path's own compacted pixels:
    hypotheses bit-exact, inlier counts and winners exact."""
    mask, planar, _ = synth.make_batch(3, first_index=700 + hn, h=200, w=260, radius=20, noise=True,
                                       background="normal")
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    seed = 11
    out, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, literal=True,
                                             return_debug=True)
    for bi in range(3):
        tn = int(dbg["tn"][bi])
        rec = dbg["rec"][bi, :, :tn]                                  # [vn,tn,4] = (x, y, ux, uy) in literal mode
        coords = rec[0, :, 0:2].contiguous()                          # [tn,2] (x, y) as torch.nonzero()[:, [1,0]] (:542-543)
        direct = rec[:, :, 2:4].permute(1, 0, 2).contiguous()         # [tn,vn,2]
        idxs = torch.from_numpy(O.draw_idxs(seed, bi, hn, 9, tn)).to(dev())
        hyp_ref = refkernels.generate_hypothesis(direct, coords, idxs)            # [hn,vn,2]
        ours = dbg["hyp"][bi].permute(1, 0, 2).contiguous()
        assert ours.cpu().numpy().tobytes() == hyp_ref.cpu().numpy().tobytes()
        inl = refkernels.voting_for_hypothesis(direct, coords, hyp_ref, thresh)   # [hn,vn,tn]
        counts_ref = inl.sum(2, dtype=torch.int32)                                # :557
        assert torch.equal(dbg["counts"][bi].T.contiguous(), counts_ref)
        win_cnt, win_idx = torch.max(counts_ref, 0)                               # :558 (first maximum, as torch.max does on CPU)
        first = (counts_ref == win_cnt[None]).int().argmax(0)
        assert torch.equal(dbg["win"][bi, :, 0].long(), first)
        assert torch.equal(dbg["win"][bi, :, 1], win_cnt)
        # refinement (:579-594): inliers of the winner by the reference kernel, normal equations in float64
        wp = hyp_ref[first, torch.arange(9, device=dev())][None].contiguous()     # [1,vn,2]
        win_inl = refkernels.voting_for_hypothesis(direct, coords, wp, thresh)[0].double()  # [vn,tn]
        normal = torch.stack([direct[:, :, 1], -direct[:, :, 0]], 2).permute(1, 0, 2).double() * win_inl[:, :, None]
        bvec = (normal * coords.double()[None]).sum(2)
        ata = normal.transpose(1, 2) @ normal
        atb = (normal * bvec[:, :, None]).sum(1)
        ref_pts = torch.linalg.solve(ata, atb[:, :, None])[:, :, 0]
        assert (out[bi].double() - ref_pts).abs().max() < 1e-3


def test_default_mode_against_reference_inlier_sets():
    """Default (exact) mode: hypotheses and EVERY inlier count equal literal mode's, which equals the reference kernels'
    (the tests above) -- no tolerance (the round-1 slack of two votes is gone)."""
    mask, planar, _ = synth.make_batch(2, first_index=820, h=240, w=320, radius=24, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev())
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev()))
    hn, thresh, seed = 256, 0.99, 5
    _, lit = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, literal=True, return_debug=True)
    lit_counts = lit["counts"].clone()
    lit_hyp = lit["hyp"].clone()
    _, fast = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, return_debug=True)
    assert fast["hyp"].cpu().numpy().tobytes() == lit_hyp.cpu().numpy().tobytes()  # hypotheses do not depend on the mode
    assert torch.equal(fast["counts"], lit_counts)
    assert torch.equal(fast["win"], lit["win"])
    # the approximate mode (PVNET_F_APPROX) is the one that may differ, on ~1e-7 of the pair tests
    _, apx = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=seed, approx=True, return_debug=True)
    diff = (apx["counts"] - lit_counts).abs()
    tn = apx["tn"][:2].sum().item()
    assert diff.sum().item() <= 2e-6 * hn * 9 * tn + 2 and diff.max().item() <= 2


@pytest.mark.skipif(not refkernels.available("fast"), reason="fma build of the reference kernels absent")
def test_fma_contraction_moves_the_reference_itself():
    """nvcc (--fmad=true) and hipcc both contract a*b+c by default, so the reference's own binaries differ from the
    one-rounding-per-operation reading of its source: hypotheses by a few ulp, inlier flags on ~1e-6 of the pairs.
    That is the reference's own noise floor; our tolerance (1e-3 px) and the literal mode sit inside it."""
    _, _, coords, direct = compacted(first=610, h=200, w=260, radius=22)
    tn = coords.shape[0]
    hn = 256
    idxs = np.random.default_rng(9).integers(0, tn, (hn, 9, 2), dtype=np.int32)
    d, c, i = (torch.from_numpy(x).to(dev()) for x in (direct, coords, idxs))
    h_off = refkernels.generate_hypothesis(d, c, i, contract="off")
    h_fma = refkernels.generate_hypothesis(d, c, i, contract="fast")
    rel = ((h_off - h_fma).abs() / h_off.abs().clamp_min(1.0)).max().item()
    assert rel < 1e-2  # ill-conditioned intersections (near-parallel lines) amplify the last-bit differences
    i_off = refkernels.voting_for_hypothesis(d, c, h_off, 0.99, contract="off")
    i_fma = refkernels.voting_for_hypothesis(d, c, h_off, 0.99, contract="fast")
    flips = (i_off != i_fma).float().mean().item()
    assert flips < 1e-4
