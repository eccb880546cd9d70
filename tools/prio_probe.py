"""Development aid: does stream priority make the small stages overlap the scoring kernels of other batches?
Per batch: K1..K3 on a HIGH-priority stream, K4 on a normal stream (event dependency), K5 back on the high one.
Stages are selected with PVNET_DEV_STAGES; every batch slot owns its workspace.   python tools/prio_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
lib = voting.load_library()
B, H, W, VN, HN = 32, 480, 640, 9, 1024
sets = []
for s in range(2):
    mask, planar, _ = synth.make_batch(B, first_index=s * B, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
L = voting.vote_layout(B, H, W, VN, HN, 30000)
SLOTS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ws = [torch.empty(L.total_bytes, dtype=torch.uint8, device=dev) for _ in range(SLOTS)]
outs = [torch.empty((B, VN, 2), dtype=torch.float32, device=dev) for _ in range(SLOTS)]


def call(slot, i, stages, stream):
    m, v = sets[i % 2]
    os.environ["PVNET_DEV_STAGES"] = str(stages)
    voting.reload_tuning()
    rc = lib.pvnet_vote_v3(C.c_void_p(m.data_ptr()), 3, voting._strides(m, 3), C.c_void_p(v.data_ptr()),
                           voting._strides(v, 5), B, H, W, VN, HN, C.c_float(0.99), 5, 30000, C.c_uint64(i), 0, None, 0,
                           C.c_void_p(outs[slot].data_ptr()), None, C.c_void_p(ws[slot].data_ptr()),
                           C.c_size_t(L.total_bytes), C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


def run(n, mode, hi, lo):
    evs = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(SLOTS)]
    for i in range(n):
        s = i % SLOTS
        if mode == "plain":
            call(s, i, 0x3F, lo[s % len(lo)])
        else:
            h, l = hi[s % len(hi)], lo[s % len(lo)]
            h.wait_event(evs[s][1])              # the slot's previous K4 has consumed its workspace
            call(s, i, 0x0F, h)
            evs[s][0].record(h)
            l.wait_event(evs[s][0])
            call(s, i, 0x10, l)
            evs[s][1].record(l)
            if mode == "split3":
                h.wait_event(evs[s][1])
                call(s, i, 0x20, h)
            else:
                call(s, i, 0x20, l)
    torch.cuda.synchronize()


lo_pri, hi_pri = 0, -1
for mode, nhi, nlo in (("plain", 0, SLOTS), ("split2", 1, 1), ("split2", 2, 2), ("split2", SLOTS, SLOTS),
                       ("split3", 1, 1), ("split3", 2, 2), ("split3", SLOTS, SLOTS), ("split2", 1, 2), ("plain", 0, SLOTS)):
    hi = [torch.cuda.Stream(dev, priority=hi_pri) for _ in range(nhi)]
    lo = [torch.cuda.Stream(dev, priority=lo_pri) for _ in range(nlo)]
    run(200, mode, hi, lo)
    t0 = time.perf_counter()
    run(K, mode, hi, lo)
    dt = (time.perf_counter() - t0) / K
    print(f"{mode:7s} slots {SLOTS} hi-streams {nhi} lo-streams {nlo}: {dt * 1e3:.4f} ms per batch of 32", flush=True)
