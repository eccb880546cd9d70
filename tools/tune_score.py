"""Development aid: sweep the scoring-kernel knobs (env vars read by pvnet_vote_layout / launch_all) on one
synthetic batch and print per-stage GPU times.   python tools/tune_score.py"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("TUNE_B", 32))
R = int(os.environ.get("TUNE_R", 40))
mask, planar, _ = synth.make_batch(B, radius=R, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))


def run(n=10, **kw):
    acc = {}
    for i in range(n + 2):
        _, t = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i, stage_times=True, **kw)
        if i >= 2:
            for k, x in t.items():
                acc[k] = acc.get(k, 0) + x / n
    return acc


configs = []
DEFAULT_ENV = dict(PVNET_SCORE_MODE="", PVNET_SCORE_WGS_PER_CU="", PVNET_SCORE_HPL="", PVNET_SCORE_CHUNK="")
if os.environ.get("TUNE_DEFAULTS", "1") == "1":
    for wgs, hpl, chunk in itertools.product([4, 8], [4, 8], [128, 256]):
        configs.append(dict(PVNET_SCORE_WGS_PER_CU=wgs, PVNET_SCORE_HPL=hpl, PVNET_SCORE_CHUNK=chunk))
for extra in sys.argv[1:]:
    configs.append(dict(kv.split("=") for kv in extra.split(",")))
ROUNDS = int(os.environ.get("TUNE_ROUNDS", 3))
res = {}
for rnd in range(ROUNDS):  # interleaved rounds: run-to-run drift is a few percent on this part
    for i, c in enumerate(configs):
        for k in DEFAULT_ENV:  # every configuration starts from the library defaults
            os.environ.pop(k, None)
        for k, x in c.items():
            os.environ[k] = str(x)
        voting.reload_tuning()
        res.setdefault(i, []).append(run())
for i, c in enumerate(configs):
    ts = res[i]
    med = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
    tot = sorted(sum(t.values()) for t in ts)
    print(" ".join(f"{k[6:]}={x}" for k, x in c.items()), "| score med %.1f min %.1f us | total med %.1f min %.1f us |" %
          (med["score"] * 1e3, min(t["score"] for t in ts) * 1e3, tot[len(tot) // 2] * 1e3, tot[0] * 1e3),
          " ".join(f"{k}={x * 1e3:.1f}" for k, x in med.items() if k != "score"), flush=True)
