"""The exactness argument of the exact scoring mode as a MEASURED margin.

The scoring kernel trusts the matrix pipe's x = s sigma (dt - |cr|) wherever |x| >= 1 and re-evaluates the cell otherwise.  The
bound behind that (band_constant(): 10 u for the reference's own roundings, 10 u sum|terms| for the MFMA's undocumented
summation order, 8 u for the staging) is a proof with one measured constant.  This tool measures the claim itself: on
every workspace it re-evaluates EVERY test both ways (pvnet_vote_band_margin) and reports the largest |x| among the tests on
which the matrix pipe's vote (x > 0) and the reference's float32 vote differ.  1 would be a wrong count; the gap to 1 is
the safety margin.      python tools/band_margin.py [--quick]      (MI355X; ~1e10 tests in the full run)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402


def cases(quick):
    rng = np.random.default_rng(7)
    # (name, batch, h, w, radius, noise, field transform)
    yield "bench field (radius 40, 0.05 rad + 10 % outliers)", 32, 480, 640, 40, True, None
    yield "large objects (radius 97, tn ~ 29.5 k)", 8, 480, 640, 97, True, None
    if not quick:
        yield "clean field", 16, 480, 640, 40, False, None
        yield "HD frame 1080 x 1920, radius 150", 2, 1080, 1920, 150, True, None
    def scale_dirs(planar):   # |u| from 1e-6 to 1e12, a different power of ten per pixel and key-point
        s = 10.0 ** rng.uniform(-6, 12, size=(planar.shape[0], planar.shape[1] // 2, planar.shape[2], planar.shape[3]))
        return (planar * np.repeat(s, 2, axis=1)).astype(np.float32)
    yield "bench field, |u| scaled by 1e-6 .. 1e12 per pixel", 8, 480, 640, 40, True, scale_dirs
    def near_parallel(planar):   # directions squeezed towards one axis: hypotheses up to ~1e6 px from the object
        p = planar.copy()
        p[:, 1::2] *= 1e-4
        return p
    yield "near-parallel field (hypotheses far from the image)", 8, 480, 640, 40, True, near_parallel


def main():
    quick = "--quick" in sys.argv
    dev = torch.device("cuda:0")
    thresholds = (0.99, 0.999) if quick else (0.5, 0.9, 0.99, 0.999, 0.9999)
    total, worst_all = 0, 0.0
    print(f"{'case':58s} {'thresh':>7s} {'tests':>12s} {'in band':>10s} {'disagree':>9s} {'max |x| of a disagreement':>26s}")
    for name, b, h, w, radius, noise, tf in cases(quick):
        mask, planar, _ = synth.make_batch(b, first_index=100, h=h, w=w, radius=radius, noise=noise, background="normal")
        if tf is not None:
            planar = tf(planar)
        m = torch.from_numpy(mask).to(dev)
        v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
        for t in thresholds:
            for seed in ((1,) if quick else (1, 2)):
                _, dbg = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=t, seed=seed, return_debug=True)
                r = voting.band_margin(dbg, t)
                total += r["tests"]
                worst_all = max(worst_all, r["worst"])
                print(f"{name:58s} {t:7.4f} {r['tests']:12d} {r['band']:10d} {r['disagree']:9d} {r['worst']:26.6f}", flush=True)
    print(f"\n{total:.3e} tests; largest |x| of any test on which the matrix pipe and the reference disagree: {worst_all:.6f} "
          f"(the kernel re-evaluates every cell with a |x| < 1: safety factor {1.0 / max(worst_all, 1e-30):.1f})")
    return 0 if worst_all < 0.5 else 1


if __name__ == "__main__":
    sys.exit(main())
