"""Development aid: multi-stream throughput of the whole path against one launch-time knob (default: the scoring grid
size PVNET_SCORE_WGS_PER_CU; KNOB=<env var> WGS_LIST=<comma list> select another).
    python tools/wgs_probe.py [streams] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
sets = []
for s in range(2):
    mask, planar, _ = synth.make_batch(32, first_index=s * 32, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
streams = [torch.cuda.Stream(dev) for _ in range(S)]


def run(n):
    for i in range(n):
        m, v = sets[i % 2]
        with torch.cuda.stream(streams[i % S]):
            voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i)
    torch.cuda.synchronize()


run(400)
for rnd in range(2):
    for wgs in (os.environ.get("WGS_LIST", "3,4,6,8,12,0")).split(","):
        os.environ[os.environ.get("KNOB", "PVNET_SCORE_WGS_PER_CU")] = wgs
        voting.reload_tuning()
        run(100)
        t0 = time.perf_counter()
        run(K)
        dt = (time.perf_counter() - t0) / K
        print(f"{S} streams, {os.environ.get('KNOB', 'PVNET_SCORE_WGS_PER_CU')}={wgs:>2s}: {dt * 1e3:.4f} ms per batch of 32",
              flush=True)
