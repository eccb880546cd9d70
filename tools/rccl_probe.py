"""What does an initialised RCCL communicator cost the six-stream voting loop at world = 1?  
One process, the same 6-stream loop of independent batches timed in four states:
  plain | process group initialised (nccl = RCCL, one rank) | after one all-reduce (communicator really created) | group destroyed
    python tools/rccl_probe.py [steps]
Under rocprofv3 --kernel-trace the dispatches' queue ids show which hardware queues the voting streams are mapped to in each state."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
sets = []
for s in range(4):
    mask, planar, _ = synth.make_batch(32, first_index=1000 * s, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
L = voting.vote_layout(32, 480, 640, 9, 1024, 30000)


def loop(nstreams, streams, spaces, n):
    for i in range(n):
        m, v = sets[i % 4]
        with torch.cuda.stream(streams[i % nstreams]):
            voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i, workspace=spaces[i % nstreams],
                                          concurrent=nstreams > 1)


def measure(tag, nstreams=6):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    spaces = [torch.empty(L.total_bytes, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    loop(nstreams, streams, spaces, 200)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        loop(nstreams, streams, spaces, steps)
        torch.cuda.synchronize()
        best = max(best, 32 * steps / (time.perf_counter() - t0))
    print(f"{tag:58s} {nstreams} streams: {best / 1e3:7.1f} k votings/s", flush=True)
    return best


measure("plain")
measure("plain", 1)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
measure("process group initialised (no collective yet)")
t = torch.zeros(1, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
measure("after one all-reduce (communicator created)")
measure("after one all-reduce (communicator created)", 1)
extra = [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]   # does a high-priority stream by itself do the same?
measure("+ two idle high-priority torch streams")
dist.destroy_process_group()
torch.cuda.synchronize()
measure("process group destroyed")
