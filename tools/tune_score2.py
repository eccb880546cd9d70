"""Development aid: scoring-kernel knob sweep by the kernel's own duration (device clock stamps, stage_repeat_ms).
    python tools/tune_score2.py"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(32, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
KNOBS = ("PVNET_SCORE_WGS_PER_CU", "PVNET_SCORE_HPL", "PVNET_SCORE_CHUNK", "PVNET_SCORE_XCD")
configs = [dict(zip(KNOBS, c)) for c in itertools.product([4, 6, 8, 12, 16], [4, 8], [64, 128, 256], [1])]
configs += [dict(zip(KNOBS, c)) for c in [(8, 8, 128, 0), (0, 8, 128, 1)]]
res = {}
for rnd in range(3):
    for i, c in enumerate(configs):
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, x in c.items():
            os.environ[k] = str(x)
        voting.reload_tuning()
        try:
            res.setdefault(i, []).append(voting.stage_repeat_ms(m, v, 1024, inlier_thresh=0.99, stage="score", repeats=40, both=True))
        except RuntimeError as e:
            res.setdefault(i, []).append((float("nan"), float("nan")))
for i, c in sorted(enumerate(configs), key=lambda ic: np.nanmedian([x[0] for x in res[ic[0]]])):
    d = [x[0] * 1e3 for x in res[i]]
    e = [x[1] * 1e3 for x in res[i]]
    print(" ".join(f"{k[12:]}={x}" for k, x in c.items()), "| kernel med %.1f min %.1f us | events med %.1f us" % (np.nanmedian(d), np.nanmin(d), np.nanmedian(e)), flush=True)
