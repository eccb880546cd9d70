"""Secondary workloads (not the headline): per-call latency / throughput of the voting layer on other configs of
SURVEY.md section 8(d) and of the reference's call sites.   python tools/bench_configs.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")


def run(name, b, radius, hn, thresh, max_num=30000, mask_dtype=torch.int64, steps=100, **kw):
    mask, planar, _ = synth.make_batch(b, radius=radius, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev).to(mask_dtype)
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    for i in range(3):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=i, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=i, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _, dbg, st = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=0,
                                               return_debug=True, stage_times=True, **kw)
    tn = float(dbg["tn"].float().mean())
    print(f"{name:44s} b={b:3d} tn~{tn:7.0f} hn={hn:5d} thr={thresh}: {dt * 1e6:8.1f} us/call  {b / dt:10.0f} votings/s  "
          f"score {st['score'] * 1e3:7.1f} us  ({hn * 9 * tn * b / st['score'] / 1e9:6.2f} Tpairs/s)", flush=True)


run("headline (cfg 3): R=40 int64", 32, 40, 1024, 0.99)
run("uint8 mask", 32, 40, 1024, 0.99, mask_dtype=torch.uint8)
run("thresh 0.999", 32, 40, 1024, 0.999)
run("stress: R=97 (tn~29.5k)", 32, 97, 1024, 0.99)
run("demo call site: b=1 hn=512", 1, 27, 512, 0.99)
run("eval call site: b=1 hn=128 max_num=100", 1, 40, 128, 0.99, max_num=100)
run("batch 8", 8, 40, 1024, 0.99)
run("batch 128", 128, 40, 1024, 0.99, steps=10)
run("literal mode (reference fp32 order)", 32, 40, 1024, 0.99, literal=True, steps=5)
run("hn=4096 (distribution estimate), b=8", 8, 40, 4096, 0.99, steps=10)
run("hn=256, b=32", 32, 40, 256, 0.99)
