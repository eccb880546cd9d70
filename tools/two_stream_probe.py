"""Development aid: does pipelining independent batches over two HIP streams raise votings/s?  (The scoring kernel
keeps 12 of a CU's 32 wave slots, so the small latency-bound stages of the next batch can run beside it.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
B, K = 32, int(os.environ.get('PROBE_K', 100))
sets = []
for i in range(2):
    mask, planar, _ = synth.make_batch(B, first_index=1000 * i, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))


def run(nstreams):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    outs = [None] * nstreams
    for it in range(10 + K):
        if it == 10:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        s = streams[it % nstreams]
        with torch.cuda.stream(s):
            m, v = sets[it % 2]
            outs[it % nstreams] = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=it)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K, t_issue / K


for n in ([int(x) for x in os.environ['PROBE_STREAMS'].split(',')] if 'PROBE_STREAMS' in os.environ else (1, 2, 4, 1, 4, 8, 1, 4)):
    dt, ti = run(n)
    print(f"{n} stream(s): {dt * 1e3:.4f} ms per batch of {B} -> {B / dt:,.0f} votings/s   (host issue {ti * 1e3:.4f} ms per call)", flush=True)
