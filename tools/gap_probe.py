"""Development aid: are there periodic holes in the six-stream steady state WITHOUT a profiler attached?  One event per call
(recorded on the call's stream); the completion times of the batches, sorted, should be ~one step apart.
    python tools/gap_probe.py [steps] [streams]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 900
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
sets = []
for s in range(4):
    mask, planar, _ = synth.make_batch(32, first_index=s * 32, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
L = voting.vote_layout(32, 480, 640, 9, 1024, 30000)
streams = [torch.cuda.Stream(dev) for _ in range(S)]
ws = [torch.empty(L.total_bytes, dtype=torch.uint8, device=dev) for _ in range(S)]
outs = [torch.empty((32, 9, 2), dtype=torch.float32, device=dev) for _ in range(S)]


import time
host = []


def run(n, events=None):
    for i in range(n):
        m, v = sets[i % 4]
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[i % S]):
            voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i, workspace=ws[i % S], out=outs[i % S])
            if events is not None:
                events[i].record(streams[i % S])
        if events is not None:
            host.append(time.perf_counter() - t0)


run(60)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
start = torch.cuda.Event(enable_timing=True)
start.record(streams[0])
t_issue0 = time.perf_counter()
run(K, ev)
t_issue = time.perf_counter() - t_issue0
torch.cuda.synchronize()
t_all = time.perf_counter() - t_issue0
h = np.array(host) * 1e6
print(f"host: issuing {K} calls took {t_issue * 1e3:.1f} ms of the {t_all * 1e3:.1f} ms until the device was idle; per call (us): median {np.median(h):.1f}, "
      f"p90 {np.percentile(h, 90):.1f}, p99 {np.percentile(h, 99):.1f}, max {h.max():.1f}; calls above 100 us: {(h > 100).sum()} holding {h[h > 100].sum() / 1e3:.1f} ms")
t = np.sort(np.array([start.elapsed_time(e) for e in ev]))  # ms
d = np.diff(t) * 1e3  # us
steady = d[50:-20]
print(f"{K} batches on {S} streams: {t[-1] / K * 1e3:.1f} us per batch overall; completion-to-completion deltas (us): median {np.median(steady):.1f}, "
      f"p90 {np.percentile(steady, 90):.1f}, p99 {np.percentile(steady, 99):.1f}, max {steady.max():.1f}")
big = np.where(d > 2.0 * np.median(steady))[0]
print(f"deltas above twice the median: {len(big)}; at batch indices {big[:30].tolist()}")
if len(big) > 2:
    print("distance between them (batches):", np.diff(big)[:30].tolist(), " time lost in them:", round(float((d[big] - np.median(steady)).sum()) / 1e3, 2), "ms of", round(float(t[-1]), 1), "ms")
