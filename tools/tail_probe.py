"""How balanced is the persistent scoring grid?  Re-launch the (clock-stamping) scoring kernel once and read every workgroup's
start / end stamp: the launch lasts until the LAST workgroup ends; what the others leave idle is the tail.
    python tools/tail_probe.py [--approx]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

approx = "--approx" in sys.argv
dev = torch.device("cuda:0")
lib = voting.load_library()
first = int(os.environ.get("PROBE_FIRST", "0"))  # first image index of the synthetic batch (which images land on which XCD)
mask, planar, _ = synth.make_batch(32, first_index=first, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
b, h, w, vn, hn = 32, 480, 640, 9, 1024
L = voting.vote_layout(b, h, w, vn, hn, 30000)
ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
ms = (C.c_float * 2)()
flags = voting.F_APPROX if approx else 0
for rep in range(3):
    voting._check(lib.pvnet_vote_v3_stage_repeat(
        C.c_void_p(m.data_ptr()), voting._MASK_CODES[m.dtype], voting._strides(m, 3), C.c_void_p(v.data_ptr()),
        voting._strides(v, 5), b, h, w, vn, hn, C.c_float(0.99), 5, 30000, C.c_uint64(1), 0, None, flags,
        C.c_void_p(out.data_ptr()), None, C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), voting.STAGE_NAMES.index("score"), 3, ms), "stage_repeat")
torch.cuda.synchronize()
grid = 256 * int(os.environ.get("PVNET_SCORE_WGS_PER_CU", "8"))  # the persistent grid of the scoring launch
st = ws[L.off_pix:L.off_pix + 16 * grid].view(torch.int64).cpu().numpy().reshape(grid, 2)
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0   # 100 MHz constant-rate clock -> microseconds
total = end.max()
print(f"{'approx' if approx else 'exact'}: kernel {ms[0] * 1e3:.1f} us by stamps; {grid} workgroups; launch {total:.1f} us")
print("workgroup END times (us) percentiles  5 / 25 / 50 / 75 / 95 / max:", np.percentile(end, [5, 25, 50, 75, 95, 100]).round(1))
print("workgroup START times (us) percentiles 50 / 75 / 95 / max:", np.percentile(start, [50, 75, 95, 100]).round(1))
busy = (end - start).sum()
resident = 256 * 3
print(f"sum of workgroup lifetimes {busy:.0f} us = {busy / resident:.1f} us per resident slot (3 per CU) against the launch's "
      f"{total:.1f} us: {100 * (1 - busy / resident / total):.1f} % of the slot-time is tail / ramp")
life = end - start
print("workgroup lifetime (us) percentiles 5 / 50 / 95:", np.percentile(life, [5, 50, 95]).round(1))
# the workgroups that end last: index, XCD (index % 8), position in its XCD's dispatch order, start, end
order = np.argsort(-end)[:12]
print("last to end:  " + "  ".join(f"wg{i}(x{i % 8},j{i // 8}) {start[i]:.0f}->{end[i]:.0f}" for i in order))
first = np.argsort(end)[:6]
print("first to end: " + "  ".join(f"wg{i}(x{i % 8},j{i // 8}) {start[i]:.0f}->{end[i]:.0f}" for i in first))
# per dispatch position (j = index // 8, averaged over the 8 XCDs): start and lifetime
jj = np.arange(grid) // 8
for lo_j in range(0, grid // 8, 32):
    sel = (jj >= lo_j) & (jj < lo_j + 32)
    print(f"  j {lo_j:3d}..{lo_j + 31:3d}: start {start[sel].mean():6.1f}  lifetime {life[sel].mean():6.1f}  end {end[sel].mean():6.1f} (max {end[sel].max():6.1f})")
# per XCD (workgroup index % 8): when its last workgroup ended, and the sum of its workgroups' lifetimes
for x in range(8):
    sel = (np.arange(grid) % 8) == x
    print(f"  XCD {x}: last end {end[sel].max():6.1f} us   sum of lifetimes {life[sel].sum():8.0f} us   mean lifetime {life[sel].mean():5.1f} us")
