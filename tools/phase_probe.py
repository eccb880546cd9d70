"""Where does the exact scoring kernel spend its time?  The clock-stamping instantiation (pvnet_vote_v3_stage_repeat) also
accumulates, per workgroup, the shader-clock cycles its wave 0 spent in the four phases of a work item:
  0 staging (record loads, A rows into LDS, the barrier)   1 the scoring loop   2 count flush + cell list + barrier
  3 literal re-evaluation of the flagged cells + the next item's first barrier
    python tools/phase_probe.py [thresh] [clean]
With PVNET_SCORE_CULL=1 in the environment the disc-culling kernel is the one timed; its four phases are
  0 staging   1 coarse pass + barrier   2 fine pass (gathered groups)   3 flush + cell list + re-evaluation + barriers"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

thresh = next((float(x) for x in sys.argv[1:] if x.replace(".", "").isdigit()), 0.99)
dev = torch.device("cuda:0")
lib = voting.load_library()
CLEAN = "clean" in sys.argv
CULL = os.environ.get("PVNET_SCORE_CULL") == "1"
mask, planar, _ = synth.make_batch(32, first_index=0, radius=40, noise=not CLEAN, background="zeros" if CLEAN else "normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
b, h, w, vn, hn = 32, 480, 640, 9, 1024
L = voting.vote_layout(b, h, w, vn, hn, 30000)
ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
ms = (C.c_float * 2)()
for rep in range(3):
    voting._check(lib.pvnet_vote_v3_stage_repeat(
        C.c_void_p(m.data_ptr()), voting._MASK_CODES[m.dtype], voting._strides(m, 3), C.c_void_p(v.data_ptr()),
        voting._strides(v, 5), b, h, w, vn, hn, C.c_float(thresh), 5, 30000, C.c_uint64(1), 0, None, 0,
        C.c_void_p(out.data_ptr()), None, C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), voting.STAGE_NAMES.index("score"), 20, ms), "stage_repeat")
torch.cuda.synchronize()
grid = 256 * int(os.environ.get("PVNET_SCORE_WGS_PER_CU", "9" if CULL else "12"))   # (a batch alone: 12 workgroups per CU, the culling kernel 9)
raw = ws[L.off_pix:L.off_pix + 8 * 6 * grid].view(torch.int64).cpu().numpy()
st = raw[:2 * grid].reshape(grid, 2)
ph = raw[2 * grid:6 * grid].reshape(grid, 4).astype(np.float64)
life_us = (st[:, 1] - st[:, 0]) / 100.0
tot = ph.sum(1)
mhz = np.median(tot / np.maximum(life_us, 1e-9))
print(f"{'clean' if CLEAN else 'noisy'} field, {'every key-point culled' if CULL else 'full kernel'}, thresh {thresh}: kernel {ms[0] * 1e3:.1f} us by stamps ({ms[1] * 1e3:.1f} us by events); {grid} workgroups; "
      f"shader clock ~{mhz:.0f} MHz (cycles / lifetime)")
names = ("staging+barrier", "coarse pass+barrier", "fine pass (gathered)", "flush+cells+re-evaluation") if CULL else \
    ("staging+barrier", "scoring loop", "flush+cells+barrier", "re-evaluation+barrier")
for i, n in enumerate(names):
    print(f"  phase {i} {n:24s} {100 * ph[:, i].sum() / tot.sum():5.1f} % of wave-0 cycles   "
          f"(median per workgroup {np.median(ph[:, i]):9.0f} cycles)")
items = int(voting._debug_views(ws, L)["total_items"])
print(f"  {items} work items, {items / grid:.2f} per workgroup; cycles per item: "
      + ", ".join(f"{ph[:, i].sum() / items:.0f}" for i in range(4)) + f"  (sum {tot.sum() / items:.0f})")
