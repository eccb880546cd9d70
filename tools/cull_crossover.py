"""Round 6 (VERDICT r05 "Next" 1c): where does disc culling start to pay?  Sweeps the synthetic field's angular noise sigma x the share
of outlier directions x the inlier threshold at the benchmark shape (batch 32, 480 x 640, 9 key-points, 1024 hypotheses) and, per
configuration, runs the exact mode with NO key-point culled and with EVERY key-point culled (PVNET_SCORE_CULL = 0 / 1) on one box:
scoring stage and whole call (one stream, event-timed), the share of the full kernel's steps the fine pass still executes, and the
statistic K3's selection uses -- q = S / (rho tan(theta0)), S = spread of the eight candidate intersections of the band-origin
estimate (kp_preamble, k3_hypotheses.hip), recomputed here from the records of the call.  The crossover in q is what
PVNET_CULL_Q_MILLI is set from; the last column is what the library's own selection (the default) then does.
    python tools/cull_crossover.py [quick | outliers]       (MI355X)   -> profiles/r06_cull_crossover.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
QUICK = "quick" in sys.argv


def set_cull(v):
    if v is None:
        os.environ.pop("PVNET_SCORE_CULL", None)
    else:
        os.environ["PVNET_SCORE_CULL"] = str(v)
    voting.reload_tuning()


def spread_q(dbg, thresh):
    """q per (image, key-point) as kp_preamble computes it (float64 here; the candidates are the same fixed record pairs)"""
    rec = dbg["rec"].cpu().numpy().astype(np.float64)
    tns = dbg["tn"].cpu().numpy()
    b, vn = rec.shape[0], rec.shape[1]
    tau = np.sqrt(1.0 - thresh * thresh) / thresh
    q = np.full((b, vn), np.nan)
    q7 = np.full((b, vn), np.nan)   # the same with the SEVENTH smallest distance (of eight) instead of the median one
    for bi in range(b):
        tn = int(tns[bi])
        if tn <= 0:
            continue
        rho = max(8.0, 0.6 * np.sqrt(0.3183 * tn))
        for k in range(vn):
            cand = []
            for j in range(8):
                ta = ((2 * j + 1) * tn) >> 4
                tb = ta + tn // 2
                tb = tb - tn if tb >= tn else tb
                x0, y0, ux0, uy0 = rec[bi, k, ta]
                x1, y1, ux1, uy1 = rec[bi, k, tb]
                nx0, ny0, nx1, ny1 = uy0, -ux0, uy1, -ux1
                dety, detx = nx1 * ny0 - nx0 * ny1, ny1 * nx0 - ny0 * nx1
                if abs(dety) < 1e-6 or abs(detx) < 1e-6:
                    continue
                b0, b1 = nx0 * x0 + ny0 * y0, nx1 * x1 + ny1 * y1
                hy, hx = (nx1 * b0 - nx0 * b1) / dety, (ny1 * b0 - ny0 * b1) / detx
                if abs(hx) < 1048576 and abs(hy) < 1048576:
                    cand.append((hx, hy))
            if len(cand) < 3:
                continue
            c = np.array(cand)
            n = len(c)
            med = np.array([np.sort(c[:, 0])[n // 2], np.sort(c[:, 1])[n // 2]])
            ds = np.sort(np.abs(c - med).max(1))
            q[bi, k] = ds[n // 2] / (rho * tau)
            q7[bi, k] = ds[max(n // 2, n - 2)] / (rho * tau)
    return q, q7


def measure(m, v, thresh, reps):
    L = voting.vote_layout(m.shape[0], m.shape[1], m.shape[2], 9, 1024, 30000)
    ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
    out, dbg = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=7, workspace=ws, return_debug=True, band_stats=True)
    res = dict(counts=dbg["counts"].clone(), steps=dbg["cull_stats"], bits=float(dbg["cull_bits"].float().mean()), dbg=dbg)
    for i in range(3):
        voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=i, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=i, workspace=ws)
    torch.cuda.synchronize()
    res["call_us"] = (time.perf_counter() - t0) / reps * 1e6
    # what the library culls in the STEADY state: whether a batch may be culled follows the previous call on its workspace (vote_common.h)
    _, dbg2 = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=7, workspace=ws, return_debug=True)
    res["bits"] = float(dbg2["cull_bits"].float().mean())
    st = []
    for i in range(5):
        _, t = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=thresh, seed=i, workspace=ws, stage_times=True)
        st.append(t)
    res["stage_us"] = {k: float(np.median([s[k] for s in st])) * 1e3 for k in st[0]}
    return res


print("batch 32, 480 x 640, R = 40 (tn ~ 5 027), 9 key-points, 1024 hypotheses, one stream; q = spread / (rho tan theta0), median over "
      "the 288 (image, key-point)s [min .. max]")
print(f"{'sigma':>6s} {'outl':>5s} {'thr':>6s} | {'q median [min .. max]':>26s} | {'score off':>9s} {'all':>7s} | {'hyp off':>7s} {'all':>6s} | "
      f"{'call off':>8s} {'all':>7s} | {'steps left':>10s} | {'culling':>8s} | library: share culled, call us")
sigmas = (0.0, 0.01, 0.05) if QUICK else (0.0, 0.005, 0.01, 0.02, 0.03, 0.05)
OUTL = (0.0, 0.10) if QUICK else (0.0, 0.02, 0.10)
if "outliers" in sys.argv:   # many outliers, little noise: the candidates agree, the hypothesis cloud is wide
    sigmas, OUTL = (0.0, 0.005, 0.01), (0.2, 0.3, 0.5)
for thresh in ((0.99,) if QUICK or "outliers" in sys.argv else (0.99, 0.999)):
    for outl in OUTL:
        for sigma in sigmas:
            noise = sigma > 0 or outl > 0
            mask, planar, _ = synth.make_batch(32, radius=40, noise=noise, background="normal", noise_sigma=sigma, outlier_frac=outl)
            m = torch.from_numpy(mask).to(dev)
            v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
            reps = 30 if QUICK else 60
            set_cull(0)
            measure(m, v, thresh, 10)   # (unrecorded: the first configuration measured on fresh tensors reads ~8 % slow)
            a = measure(m, v, thresh, reps)
            set_cull(1)
            c = measure(m, v, thresh, reps)
            set_cull(None)
            l = measure(m, v, thresh, reps)
            same = torch.equal(a["counts"], c["counts"]) and torch.equal(a["counts"], l["counts"])
            q, q7 = spread_q(a["dbg"], thresh)
            img7 = np.nanmedian(q7, axis=1)   # per image: the median over its key-points of the seventh-smallest statistic
            ex, full = c["steps"]
            verdict = "WINS" if c["call_us"] < a["call_us"] else "loses"
            print(f"{sigma:6.3f} {outl:5.2f} {thresh:6.3f} | {np.nanmedian(q):8.3f} [{np.nanmin(q):6.3f} .. {np.nanmax(q):7.3f}] | "
                  f"{a['stage_us']['score']:9.1f} {c['stage_us']['score']:7.1f} | {a['stage_us']['hypotheses']:7.1f} {c['stage_us']['hypotheses']:6.1f} | "
                  f"{a['call_us']:8.1f} {c['call_us']:7.1f} | {ex / max(full, 1):10.3f} | {verdict:>8s} | {l['bits']:5.2f} {l['call_us']:7.1f}"
                  f" | q7 median {np.nanmedian(q7):7.3f}  images' median q7: min {np.nanmin(img7):6.3f} median {np.nanmedian(img7):6.3f} max {np.nanmax(img7):7.3f}"
                  + ("" if same else "   COUNTS DIFFER"), flush=True)
