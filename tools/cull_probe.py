"""Development aid (round 5): the disc-culling scoring kernel against the full exact kernel on one box -- same inputs, same draw:
counts / winners / key-points must be EQUAL; the scoring stage's time (device clock stamps, back to back), the whole call on one
stream, and how many of the full kernel's steps the fine pass still executes (PVNET_F_BAND_STATS).
    python tools/cull_probe.py [quick]      (MI355X)"""
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
    os.environ["PVNET_SCORE_CULL"] = str(v)
    voting.reload_tuning()


def measure(m, v, hn, thresh, conc):
    L = voting.vote_layout(m.shape[0], m.shape[1], m.shape[2], 9, hn, 30000)
    ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
    out, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=7, workspace=ws, concurrent=conc,
                                             return_debug=True, band_stats=True)
    res = dict(out=out.clone(), counts=dbg["counts"].clone(), win=dbg["win"].clone(), cull=dbg["cull"],
               band=dbg["band_stats"], steps=dbg["cull_stats"])
    for i in range(3):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=i, workspace=ws, concurrent=conc)
    torch.cuda.synchronize()
    n = 40 if QUICK else 100
    t0 = time.perf_counter()
    for i in range(n):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=i, workspace=ws, concurrent=conc)
    torch.cuda.synchronize()
    res["call_us"] = (time.perf_counter() - t0) / n * 1e6
    st = []
    for i in range(5):
        _, t = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, seed=i, workspace=ws, concurrent=conc, stage_times=True)
        st.append(t)
    res["stage_us"] = {k: float(np.median([s[k] for s in st])) * 1e3 for k in st[0]}
    if not conc:
        res["score_clock_us"] = voting.stage_repeat_ms(m, v, hn, inlier_thresh=thresh, stage="score", repeats=30 if QUICK else 100, seed=7) * 1e3
    return res


def case(name, b, radius, hn, thresh, noise=True, background="normal"):
    mask, planar, _ = synth.make_batch(b, radius=radius, noise=noise, background=background)
    m = torch.from_numpy(mask).to(dev)
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    for conc in (False, True):
        set_cull(0)
        a = measure(m, v, hn, thresh, conc)
        set_cull(1)
        c = measure(m, v, hn, thresh, conc)
        same = torch.equal(a["counts"], c["counts"]) and torch.equal(a["win"], c["win"])
        px = float((a["out"] - c["out"]).abs().max())
        ndiff = int((a["counts"] != c["counts"]).sum())
        ex, full = c["steps"]
        print(f"{name:34s} b={b:3d} hn={hn:5d} thr={thresh} conc={int(conc)}: culled={c['cull']} EQUAL={same} (differing counts {ndiff}, "
              f"max |d kpt| {px:.2e} px) | score {a['stage_us']['score']:7.1f} -> {c['stage_us']['score']:7.1f} us (events)"
              + (f", {a['score_clock_us']:7.1f} -> {c['score_clock_us']:7.1f} us (clock stamps)" if not conc else "")
              + f" | hyp stage {a['stage_us']['hypotheses']:5.1f} -> {c['stage_us']['hypotheses']:5.1f} us | call {a['call_us']:7.1f} -> "
              f"{c['call_us']:7.1f} us | fine steps {ex} of {full} ({ex / max(full, 1):.3f}) | flagged cells {a['band'][0]} -> {c['band'][0]}",
              flush=True)
    os.environ.pop("PVNET_SCORE_CULL", None)
    voting.reload_tuning()


case("headline (noisy, R=40)", 32, 40, 1024, 0.99)
case("thresh 0.999", 32, 40, 1024, 0.999)
case("thresh 0.9", 32, 40, 1024, 0.9)
case("clean field", 32, 40, 1024, 0.99, noise=False, background="zeros")
if not QUICK:
    case("R=97 (tn ~29.5 k)", 32, 97, 1024, 0.99)
    case("hn 2048, batch 8", 8, 40, 2048, 0.99)
    case("hn 777 (padding), batch 4", 4, 40, 777, 0.99)
    case("batch 1", 1, 40, 1024, 0.99)
