"""Would exact-count geometric culling of (pixel tile x hypothesis tile) blocks pay?  (VERDICT r04 "Next round" item 1.)
CPU simulation, numpy, float64 predicate; nothing here runs in the product.

The scoring kernel spends 2 MFMAs + 40 vector operations on every block of 32 pixels x 32 hypotheses.  A block could be
skipped with all counts still exact when the outcome of its 1 024 tests is known from geometry alone -- outside the rounding
band of the reference's float32 test (band_constant(), vote_host.hip).  Three questions, per threshold and field:

 (V) the review's formulation: each key-point's hypotheses sorted by distance R from the key-point estimate o (the exact
     mode's band origin), its pixels sorted by |alpha| = deviation of the pixel's direction from the direction to o; a block is
     certain when  alpha_max + asin(rho_t / r_min) < theta0 - band  (all 1 024 vote) or  alpha_min - asin(rho_t / r_min) > theta0 +
     band  (none does), rho_t = the hypothesis tile's largest R, r_min = the pixel tile's smallest distance from o;
 (U) the ceiling of ANY block-level test with those two orderings: blocks whose 1 024 true outcomes are all equal;
 (G) a stronger, finer variant worked out this round: certainty per (pixel, hypothesis tile) -- the pixel's margin at the
     tile's centre q exceeds rho_T / cos(theta0) + band, so its vote is the same for every hypothesis of the tile and is added
     as a constant -- and only the UNCERTAIN pixels of a 256-pixel work item are gathered into MFMA tiles (the A operand of an
     MFMA can take any LDS row per lane); executed steps = ceil(uncertain / 32) per (item, hypothesis tile).  Hypotheses sorted
     along a Hilbert curve (compact tiles), pixels in raster order.

    python tools/cull_study.py            -> profiles/r05_cull_study.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvnet_amd import synth  # noqa: E402


def kband(t):  # band_constant() of vote_host.hip
    u = 2.0 ** -24
    tau = np.sqrt(1 - t * t) / t
    t0 = np.arccos(t)
    d = 10 * u
    lo, hi = np.arccos(min(1, t + d)), np.arccos(t - d)
    return max(np.sin(t0 - lo), np.sin(hi - t0)) / t * 1.001 + u * (1 + tau) * (14.3 + 8)


def hilbert(ix, iy, bits):
    d = np.zeros(len(ix), np.uint64)
    x, y = ix.copy(), iy.copy()
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += np.uint64(s) * np.uint64(s) * ((3 * rx) ^ ry).astype(np.uint64)
        m = ry == 0
        f = m & (rx == 1)
        x = np.where(f, s - 1 - x, x)
        y = np.where(f, s - 1 - y, y)
        x, y = np.where(m, y, x), np.where(m, x, y)
        s >>= 1
    return d


def origin_estimate(c, u, tn):
    """what hypothesis_kernel computes: component-wise median of eight fixed-pair intersections, rounded"""
    ca = (np.arange(8) * 2 + 1) * tn // 16
    cb = (ca + tn // 2) % tn
    n0 = np.stack([u[ca, 1], -u[ca, 0]], 1)
    n1 = np.stack([u[cb, 1], -u[cb, 0]], 1)
    det = n0[:, 0] * n1[:, 1] - n0[:, 1] * n1[:, 0]
    ok = np.abs(det) > 1e-6
    det = np.where(ok, det, 1)
    b0, b1 = (n0 * c[ca]).sum(1), (n1 * c[cb]).sum(1)
    C = np.stack([(b0 * n1[:, 1] - b1 * n0[:, 1]) / det, (n0[:, 0] * b1 - n1[:, 0] * b0) / det], 1)[ok]
    return np.round(np.median(C, axis=0)) if len(C) >= 3 else c[tn // 2]


def fields(name, nimg):
    if name == "demo":   # the reference's demo fixture (tests/golden/demo_cat.npz): ground-truth field of a real mask
        d = np.load(os.path.join(ROOT, "tests", "golden", "demo_cat.npz"))
        h, w = d["shape"]
        fg = np.unpackbits(d["mask_bits"])[: h * w].reshape(h, w).astype(bool)
        planar = synth.field_from_keypoints(fg, d["points_2d"], "zeros")
        yield fg, planar, d["points_2d"]
        yield fg, synth.add_noise(planar, fg, np.random.default_rng(5)), d["points_2d"]   # + the benchmark's noise
        return
    for img in range(nimg):
        if name == "bench":
            yield synth.make_image(img, noise=True, background="normal")
        elif name == "clean":
            yield synth.make_image(img, noise=False, background="zeros")
        elif name == "big":
            yield synth.make_image(img, noise=True, background="normal", radius=97)


def study(thresh, field, nimg, hn=1024, item_px=256, out=sys.stdout):
    rng = np.random.default_rng(0)
    kb, t0 = kband(thresh), np.arccos(thresh)
    tau = np.tan(t0)
    acc = {k: [] for k in ("V", "Vvote", "U", "Uraster", "G", "Gpairs", "rad", "vote")}
    for mask, planar, kpts in fields(field, nimg):
        ys, xs = np.nonzero(mask)
        tn = len(xs)
        c = np.stack([xs, ys], 1).astype(np.float64)
        for k in range(kpts.shape[0]):
            u = np.stack([planar[2 * k][ys, xs], planar[2 * k + 1][ys, xs]], 1).astype(np.float64)
            u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-30)
            idx = rng.integers(0, tn, (hn, 2))
            n0 = np.stack([u[idx[:, 0], 1], -u[idx[:, 0], 0]], 1)
            n1 = np.stack([u[idx[:, 1], 1], -u[idx[:, 1], 0]], 1)
            det = n0[:, 0] * n1[:, 1] - n0[:, 1] * n1[:, 0]
            b0, b1 = (n0 * c[idx[:, 0]]).sum(1), (n1 * c[idx[:, 1]]).sum(1)
            ok = np.abs(det) > 1e-6
            det = np.where(ok, det, 1.0)
            H = np.stack([(b0 * n1[:, 1] - b1 * n0[:, 1]) / det, (n0[:, 0] * b1 - n1[:, 0] * b0) / det], 1) * ok[:, None]
            o = origin_estimate(c, u, tn)
            rho = max(8, 0.6 * np.sqrt(tn / np.pi))
            r = np.linalg.norm(c - o, axis=1)
            R = np.linalg.norm(H - o, axis=1)
            d = H[:, None, :] - c[None]
            nd = np.linalg.norm(d, axis=2)
            cosang = (d * u[None]).sum(2) / (nd + 1e-30)
            Vt = cosang > thresh                                  # [hn, tn] true outcomes
            acc["vote"].append(Vt.mean())
            # ---- (V), (U): hypotheses by R, pixels by |alpha|
            to_o = (o - c) / np.maximum(r, 1e-9)[:, None]
            alpha = np.arccos(np.clip((to_o * u).sum(1), -1, 1))
            ph, pp = np.argsort(R, kind="stable"), np.argsort(alpha, kind="stable")
            npt, nht = tn // 32, hn // 32                          # whole tiles only
            Vs = Vt[ph][:, pp][: nht * 32, : npt * 32].reshape(nht, 32, npt, 32)
            s = Vs.sum((1, 3))
            acc["U"].append(((s == 0) | (s == 1024)).mean())
            Vr = Vt[: nht * 32, : npt * 32].reshape(nht, 32, npt, 32).sum((1, 3))
            acc["Uraster"].append(((Vr == 0) | (Vr == 1024)).mean())
            rho_t = R[ph][: nht * 32].reshape(nht, 32).max(1)
            a_s, r_s = alpha[pp][: npt * 32].reshape(npt, 32), r[pp][: npt * 32].reshape(npt, 32)
            amax, amin, rmin = a_s.max(1), a_s.min(1), r_s.min(1)
            ratio = rho_t[:, None] / np.maximum(rmin[None], 1e-9)
            gam = np.where(ratio < 1, np.arcsin(np.minimum(ratio, 1)), np.inf)
            eps = 1e-4                                              # rad: far above the band (~1e-5), generous to the method
            allv = amax[None] + gam < t0 - eps
            none = amin[None] - gam > t0 + eps
            acc["V"].append((allv | none).mean())
            acc["Vvote"].append(allv.mean())
            # ---- (G): per (pixel, hypothesis tile) certainty, uncertain pixels of an item gathered
            f = np.clip((H - o + 256) * 8, 0, 4095).astype(np.int64)
            od = np.argsort(hilbert(f[:, 0], f[:, 1], 12), kind="stable")
            Hs = H[od][: nht * 32].reshape(nht, 32, 2)
            q = (Hs.max(1) + Hs.min(1)) / 2
            rt = np.linalg.norm(Hs - q[:, None], axis=2).max(1)
            Rq = np.linalg.norm(q - o, axis=1)
            dq = q[:, None, :] - c[None]
            mm = tau * (dq * u[None]).sum(2) - np.abs(dq[:, :, 0] * u[None, :, 1] - dq[:, :, 1] * u[None, :, 0])
            thr = rt[:, None] / thresh + kb * (Rq[:, None] + rt[:, None] + rho) * (1 + r[None] / rho)
            Un = np.abs(mm) <= thr
            nit = (tn + item_px - 1) // item_px
            Up = np.concatenate([Un, np.zeros((nht, nit * item_px - tn), bool)], 1).reshape(nht, nit, item_px)
            steps = np.ceil(Up.sum(2) / 32)
            full = np.ceil(np.minimum(item_px, tn - np.arange(nit) * item_px) / 32)
            acc["G"].append(steps.sum() / (nht * full.sum()))
            acc["Gpairs"].append(Un.mean())
            acc["rad"].append(np.median(rt))
    m = {k: float(np.mean(v)) for k, v in acc.items()}
    print(f"thresh {thresh:<6} field {field:<6} ({len(acc['V'])} key-points, {hn} hypotheses, vote fraction {m['vote']:.3f})\n"
          f"    (V) certain blocks, |alpha|-sorted pixels x R-sorted hypotheses : {m['V']:6.1%}   (all-vote {m['Vvote']:.1%})\n"
          f"    (U) blocks with 1 024 equal outcomes, same orderings (ceiling)  : {m['U']:6.1%}   (raster x caller order: {m['Uraster']:.1%})\n"
          f"    (G) executed steps with per-pixel certainty + gather            : {m['G']:6.1%} of all  -> {1 - m['G']:.1%} skipped"
          f"   (uncertain (pixel, tile) pairs {m['Gpairs']:.1%}, median tile radius {m['rad']:.2f} px)", file=out, flush=True)
    return m


if __name__ == "__main__":
    path = os.path.join(ROOT, "profiles", "r05_cull_study.txt")
    quick = "--quick" in sys.argv
    with open(path, "w") as fh:
        class Tee:
            def write(self, s):
                sys.stdout.write(s)
                fh.write(s)

            def flush(self):
                sys.stdout.flush()
                fh.flush()
        t = Tee()
        print(__doc__.split("\n\n")[0] + "\n", file=t)
        for field, nimg in (("bench", 1 if quick else 3), ("clean", 1), ("demo", 0), ("big", 1)):
            for thresh in (0.9, 0.99, 0.999):
                study(thresh, field, nimg, out=t)
