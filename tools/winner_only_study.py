"""Would a winner-only scoring path pay?  (DESIGN section 9, round-3 candidate 4.)  CPU simulation, numpy.

ransac_voting_layer_v3 returns only the refined winner, so a hypothesis that can no longer reach the leader need not be scored to
the end.  The outputs stay BIT-IDENTICAL for every input only with the safe bound: after a fraction q of the (raster-ordered)
pixels hypothesis h is dropped iff  count_q(h) + (pixels left) < count_q(leader)  -- it may still gain every remaining pixel, the
leader none.  This script scores benchmark images completely (float64 predicate) and reports, per q, how many hypotheses that rule
drops and which share of all pair tests dropping them at the best possible moments would save.
    python tools/winner_only_study.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth  # noqa: E402

rng = np.random.default_rng(0)
qs = (0.25, 0.5, 0.75, 0.85, 0.9, 0.95)
alive = {q: [] for q in qs}
saved, votefrac, best = [], [], []
for img in range(4):
    mask, planar, kpts = synth.make_image(img, noise=True, background="normal")
    ys, xs = np.nonzero(mask)
    tn = len(xs)
    c = np.stack([xs, ys], 1).astype(np.float64)
    for k in range(9):
        u = np.stack([planar[2 * k][ys, xs], planar[2 * k + 1][ys, xs]], 1).astype(np.float64)
        idx = rng.integers(0, tn, (1024, 2))
        n0 = np.stack([u[idx[:, 0], 1], -u[idx[:, 0], 0]], 1)
        n1 = np.stack([u[idx[:, 1], 1], -u[idx[:, 1], 0]], 1)
        det = n0[:, 0] * n1[:, 1] - n0[:, 1] * n1[:, 0]
        b0, b1 = (n0 * c[idx[:, 0]]).sum(1), (n1 * c[idx[:, 1]]).sum(1)
        ok = np.abs(det) > 1e-6
        det = np.where(ok, det, 1.0)
        H = np.stack([(b0 * n1[:, 1] - b1 * n0[:, 1]) / det, (n0[:, 0] * b1 - n1[:, 0] * b0) / det], 1) * ok[:, None]
        d = H[:, None, :] - c[None, :, :]
        ang = (d * u[None]).sum(2) / (np.linalg.norm(d, axis=2) * np.linalg.norm(u, axis=1)[None] + 1e-30)
        V = ang > 0.99
        cum = np.cumsum(V, axis=1)                       # count after each pixel
        lead = cum.max(axis=0)                           # the leader's count after each pixel
        left = tn - 1 - np.arange(tn)
        dead = cum + left[None, :] < lead[None, :]       # safely dropped after this pixel
        first = np.where(dead.any(1), dead.argmax(1), tn - 1)
        saved.append(1.0 - (first + 1).mean() / tn)
        for q in qs:
            alive[q].append(1.0 - dead[:, int(tn * q) - 1].mean())
        votefrac.append(V.mean())
        best.append(V.sum(1).max() / tn)
print(f"{len(saved)} (image, key-point) pairs of the benchmark field, 1024 hypotheses each; vote fraction {np.mean(votefrac):.3f}, "
      f"winner's inlier share {np.mean(best):.3f}")
for q in qs:
    print(f"  after {q:4.0%} of the pixels: {np.mean(alive[q]):6.1%} of the hypotheses still have to be scored")
print(f"pair tests saved if every hypothesis were dropped at the FIRST pixel the safe bound allows (per-pixel granularity, no overhead): "
      f"{np.mean(saved):.1%}")
