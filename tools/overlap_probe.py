"""Development aid: what do the five small stages cost the scoring kernel when independent batches overlap on
several HIP streams?  Times K steps on S streams with (a) every stage, (b) the scoring kernel alone, (c) everything
but the scoring kernel (PVNET_DEV_STAGES; single stages re-run on the workspace a complete call left behind).
    python tools/overlap_probe.py [streams] [steps]"""
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


run(400)  # complete calls: every stream's workspace block now holds a valid plan
CASES = (("all stages", 0x3F), ("scoring kernel only", 0x10), ("all but scoring", 0x2F),
         ("mask+thinning only", 0x03), ("compaction only", 0x04), ("hypotheses only", 0x08),
         ("select/refine only", 0x20), ("all stages", 0x3F))
if os.environ.get("PROBE_MARGINAL"):  # marginal cost of each small stage beside the scoring kernels of other batches
    CASES = (("all stages", 0x3F), ("without mask+thinning", 0x3C), ("without thinning check", 0x3D),
             ("without compaction", 0x3B), ("without hypotheses", 0x37), ("without select/refine", 0x1F),
             ("scoring kernel only", 0x10), ("all stages", 0x3F))
elif os.environ.get("PROBE_SHORT"):
    CASES = CASES[:3] + CASES[-1:]
print("library:", os.environ.get("PVNET_VOTE_LIB", "default"))
for name, mask in CASES:
    os.environ["PVNET_DEV_STAGES"] = str(mask)
    voting.reload_tuning()
    run(100)
    t0 = time.perf_counter()
    run(K)
    dt = (time.perf_counter() - t0) / K
    print(f"{S} streams, {name:22s}: {dt * 1e3:.4f} ms per batch of 32", flush=True)
    if mask != 0x3F:  # restore valid workspaces
        os.environ["PVNET_DEV_STAGES"] = str(0x3F)
        voting.reload_tuning()
        run(2 * S)
