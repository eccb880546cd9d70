"""Development aid: which scoring layout suits the shapes the real callers use?  For each shape (batch,
hypotheses) sweep PVNET_SCORE_HPL / PVNET_SCORE_CHUNK / PVNET_SCORE_WGS_PER_CU, interleaved over rounds (run-to-run drift is a few
per cent), and print the scoring stage's event time, the whole call back to back and T pair tests/s.  Default mode (exact).
    python tools/shape_sweep.py [quick]      (MI355X)"""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
KN = ("PVNET_SCORE_HPL", "PVNET_SCORE_CHUNK", "PVNET_SCORE_WGS_PER_CU")
SHAPES = [(32, 256), (32, 512), (8, 1024), (32, 1024)]
ROUNDS = 2 if "quick" in sys.argv else 3


def measure(m, v, hn, steps):
    L = voting.vote_layout(m.shape[0], 480, 640, 9, hn, 30000)
    ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
    for i in range(3):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=i, workspace=ws, concurrent=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=i, workspace=ws, concurrent=False)
    torch.cuda.synchronize()
    call = (time.perf_counter() - t0) / steps
    sc = []
    for i in range(5):
        _, t = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=i, workspace=ws, concurrent=False, stage_times=True)
        sc.append(t["score"])
    return call, float(np.median(sc)) * 1e-3, L


for b, hn in SHAPES:
    mask, planar, _ = synth.make_batch(b, radius=40, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev)
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    tn = float((mask != 0).sum()) / b
    cfgs = [dict()]  # the library's own choice first
    for hpl, chunk, wgs in itertools.product([2, 4, 8], [64, 128, 256], [8, 12, 16]):
        cfgs.append(dict(PVNET_SCORE_HPL=hpl, PVNET_SCORE_CHUNK=chunk, PVNET_SCORE_WGS_PER_CU=wgs))
    res = {}
    for rnd in range(ROUNDS):
        for i, c in enumerate(cfgs):
            for k in KN:
                os.environ.pop(k, None)
            for k, x in c.items():
                os.environ[k] = str(x)
            voting.reload_tuning()
            try:
                res.setdefault(i, []).append(measure(m, v, hn, 60 if b >= 32 else 120))
            except RuntimeError as e:   # a layout the library refuses
                res.setdefault(i, []).append(None)
    for k in KN:
        os.environ.pop(k, None)
    voting.reload_tuning()
    rows = []
    for i, c in enumerate(cfgs):
        ok = [r for r in res[i] if r is not None]
        if not ok:
            continue
        call = float(np.median([r[0] for r in ok]))
        score = float(np.median([r[1] for r in ok]))
        L = ok[0][2]
        rows.append((score, call, c, L))
    base = rows[0]
    rows.sort(key=lambda r: r[0])
    print(f"--- batch {b}, hn {hn}, tn ~{tn:.0f}: library default: score {base[0] * 1e6:.1f} us, call {base[1] * 1e6:.1f} us "
          f"({hn * 9 * tn * b / base[0] / 1e12:.2f} T tests/s; layout hpl {base[3].hpl} wg_g {base[3].wg_g} chunk {base[3].chunk})", flush=True)
    for score, call, c, L in rows[:8]:
        print("    " + " ".join(f"{k[12:]}={x}" for k, x in c.items()) + f" (mh {L.wg_g * L.hpl // 2}, item {L.wg_s * L.chunk} px): "
              f"score {score * 1e6:.1f} us  call {call * 1e6:.1f} us  {hn * 9 * tn * b / score / 1e12:.2f} T tests/s", flush=True)
