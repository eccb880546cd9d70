"""Randomised configurations of the voting layer with the launch knobs flipped at random -- the cases of tools/fuzz_parity.py
(rounds 2-4: 5 000 of them, run by hand) as an importable function, so that tests/test_fuzz_gpu.py runs a block of them in the
driver's GPU suite and the tool keeps running thousands.  A case is seeded by its number:

  literal HIP path vs the C oracle          : winners and their counts must be exact;
  DEFAULT (exact) mode vs literal           : every hypothesis, every count, every winner EQUAL; key-points within 1e-3 px;
  approximate mode (approx=True) vs literal : counts within a few votes (the bound grows as thresh -> 1).

Random image sizes / key-point and hypothesis counts / thresholds / thinning limits / mask dtypes / field scales, un-normalised
fields with zero and tiny directions, strided and contiguous fields, the in-flight hint.  Test infrastructure (imports oracle/)."""
import os

import numpy as np
import torch

from oracle import cref, ransac_voting_oracle as O
from pvnet_amd import synth, voting

KNOBS = {"PVNET_SCORE_XCD": ["0", "1"], "PVNET_SCORE_ATOMIC": ["0", "1"], "PVNET_SCORE_WGS_PER_CU": ["-1", "0", "2", "8", "12"],
         "PVNET_COMPACT_KG": ["1", "3", "9"], "PVNET_EXACT_FOLD": ["-1", "0", "1"],
         "PVNET_SCORE_ACC": ["-1", "1", "2"], "PVNET_SCORE_RUNS": ["-1", "0", "1"],
         #  16 .. 20 pixel tiles per work item: the range in which a fixed run length of 256 items overflowed the
         # packed vote counters of a contiguous run -- the cut now follows the tile count
         "PVNET_SCORE_CHUNK": ["-1", "-1", "-1", "256", "320"],
         "PVNET_SCORE_CULL": ["-1", "0", "1"]}


def clear_knobs():
    for k in KNOBS:
        os.environ.pop(k, None)
    voting.reload_tuning()


def run_case(case, dev=None, verbose=False):
    """-> dict(ok_literal, ok_exact, approx_diff, approx_limit, finite, desc).  Leaves the knobs of the case in os.environ
    (call clear_knobs() when done)."""
    dev = dev or torch.device("cuda:0")
    rng = np.random.default_rng(5000 + case)
    for k, vals in KNOBS.items():
        os.environ[k] = str(rng.choice(vals))
    voting.reload_tuning()
    h, w = int(rng.integers(16, 300)), int(rng.integers(16, 400))
    vn = int(rng.integers(1, 14))
    hn = int(rng.choice([8, 31, 64, 100, 128, 257, 512, 1000, 1500]))
    if hn <= 128 and os.environ["PVNET_SCORE_CHUNK"] != "-1":
        # one hypothesis group: the workgroup's four waves share FOUR chunks -- 1 024 / 1 280 pixels per item are beyond the
        # wrapped vote accumulators (pvnet_vote_layout refuses them)
        os.environ["PVNET_SCORE_CHUNK"] = "-1"
        voting.reload_tuning()
    b = int(rng.integers(1, 6))
    radius = int(rng.integers(3, max(4, min(h, w) // 2)))
    thresh = float(rng.choice([0.5, 0.9, 0.99, 0.999, 0.9999]))
    max_num = int(rng.choice([30000, 1000, 150, 40, 7]))
    mdt = rng.choice(["int64", "uint8", "int32"])
    scale = float(rng.choice([1.0, 1.0, 2.0 ** -3, 2.0 ** 9]))
    mask, planar, _ = synth.make_batch(b, first_index=9000 + 3 * case, h=h, w=w, vn=vn, radius=radius,
                                       noise=bool(rng.integers(0, 2)), background=str(rng.choice(["normal", "zeros"])),
                                       mask_dtype=getattr(np, mdt))
    planar = (planar * np.float32(scale)).astype(np.float32)
    unnorm = rng.integers(0, 3) == 0
    if unnorm:  # un-normalised field: a random positive factor per pixel and plane pair, some of them ~0
        fac = np.exp(rng.normal(0.0, 3.0, size=(b, 1, h, w))).astype(np.float32)
        fac[rng.random(fac.shape) < 0.02] = np.float32(rng.choice([0.0, 1e-7, 1.0000001e-6, 1e-5]))
        planar = (planar.reshape(b, vn, 2, h, w) * fac[:, :, None]).reshape(b, 2 * vn, h, w).astype(np.float32)
    vnp = synth.planar_to_vertex_view(planar)
    m = torch.from_numpy(mask).to(dev)
    p = torch.from_numpy(planar).to(dev)
    v = synth.planar_to_vertex_view(p) if rng.integers(0, 2) else synth.planar_to_vertex_view(p).contiguous()
    seed = int(rng.integers(0, 2 ** 40))
    desc = dict(case=case, h=h, w=w, vn=vn, hn=hn, b=b, radius=radius, thresh=thresh, max_num=max_num, mdt=str(mdt), scale=scale,
                unnorm=bool(unnorm), knobs={k: os.environ[k] for k in KNOBS})
    out, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, literal=True,
                                             return_debug=True)
    counts_l, win_l, nch = dbg["counts"].clone(), dbg["win"].cpu().numpy().copy(), dbg["nchunks"].cpu().numpy().copy()
    ref, wi, wc = cref.vote_v3(O.foreground(mask), vnp, hn, thresh, max_num=max_num, seed=seed, return_winners=True)
    live = nch > 0
    ok_literal = np.array_equal(win_l[:, :, 0][live], wi[live]) and np.array_equal(win_l[:, :, 1][live], wc[live])
    hyp_l, out_l = dbg["hyp"].clone(), out.clone()
    ex, de = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, return_debug=True,
                                           concurrent=bool(rng.integers(0, 2)))  # (the in-flight variant: contiguous item runs)
    same = (de["hyp"].cpu().numpy().tobytes() == hyp_l.cpu().numpy().tobytes() and torch.equal(de["counts"], counts_l)
            and np.array_equal(de["win"].cpu().numpy(), win_l))
    okpx = torch.isfinite(out_l).all(-1) & (out_l.abs() < 1e5).all(-1)
    px = float((ex - out_l)[okpx].abs().max()) if okpx.any() else 0.0
    ok_exact = same and px <= 1e-3 * max(1.0, float(out_l[okpx].abs().max()) / 100 if okpx.any() else 1.0)
    desc["exact_counts_differing"] = int((de["counts"] != counts_l).sum())
    desc["exact_max_count_diff"] = int((de["counts"] - counts_l).abs().max())
    desc["exact_px"] = px
    fast, df = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=seed, approx=True,
                                             return_debug=True)
    # (un-normalised fields: the approximate mode stores |u| < 1e-6 as zero records, so its hypotheses may differ -- not compared)
    cd = 0 if unnorm else int((df["counts"] - counts_l).abs().max())
    # approx vs literal drift apart as thresh -> 1 (the reference's float32 cos is flat there) and with the number of pixels a
    # hypothesis is tested on: at 0.999 and 30 000 pixels literal is off by up to 3 votes against float64 arithmetic where approx
    # is off by 0-1 (tools/experiments/fuzz_case_check.py 1046)
    tn_max = int(df["tn"].max())
    lim = 2 if thresh <= 0.99 else (2 + tn_max // 10000 if thresh <= 0.999 else 12)
    res = dict(ok_literal=bool(ok_literal), ok_exact=bool(ok_exact), approx_diff=cd, approx_limit=lim,
               finite=bool(torch.isfinite(fast).all()), thresh=thresh, desc=desc)
    if verbose and not (res["ok_literal"] and res["ok_exact"] and cd <= lim and res["finite"]):
        print("FUZZ MISMATCH", res, flush=True)
    return res
