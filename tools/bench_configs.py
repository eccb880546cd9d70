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


def run_plan(name, b, radius, hn, thresh, max_num=30000, steps=300):
    """the same call through a prepared VotePlan (no per-call allocation / layout / argument marshalling): host cost of
    a call = one ctypes call + six launches; and the GPU-side latency of ONE call on an idle stream (event pair)."""
    mask, planar, _ = synth.make_batch(b, radius=radius, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev)
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    plan = voting.VotePlan(m, v, hn, inlier_thresh=thresh, max_num=max_num)
    for i in range(5):
        plan(m, v, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        plan(m, v, seed=i)
    t_issue = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lat = []
    for i in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        plan(m, v, seed=i)
        e1.record()
        torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1) * 1e3)
    # the callers' real situation: the call is enqueued while the GPU is still busy with the backbone (2 ms here), so its
    # launches sit in the queue and run back to back -- GPU time from "backbone done" to "key-points ready"
    big = torch.randn((4096, 4096), device=dev)
    qlat = []
    for i in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for _ in range(8):
            big @ big  # ~2 ms of queued GPU work (and clocks at their working level)
        e0.record()
        plan(m, v, seed=i)
        e1.record()
        torch.cuda.synchronize()
        qlat.append(e0.elapsed_time(e1) * 1e3)
    # the same six launches replayed as ONE hipGraph (seed frozen at capture): does a graph launch beat six launches?
    try:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            plan(m, v, seed=3)
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                plan(m, v, seed=3)
        torch.cuda.synchronize()
        want = plan(m, v, seed=3).clone()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(plan.out, want))
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        g_issue = (time.perf_counter() - t0) / steps
        torch.cuda.synchronize()
        g_dt = (time.perf_counter() - t0) / steps
        glat, gq = [], []
        for i in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            glat.append(e0.elapsed_time(e1) * 1e3)
        for i in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            for _ in range(8):
                big @ big
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            gq.append(e0.elapsed_time(e1) * 1e3)
        print(f"{name:44s} b={b:3d} hn={hn:5d} hipGraph replay of the plan (same result: {same}): {g_dt * 1e6:8.1f} us/call back to back "
              f"(host issue {g_issue * 1e6:6.1f} us), idle-stream latency {np.median(glat):6.1f} us (min {min(glat):.1f}), queued "
              f"{np.median(gq):6.1f} us (min {min(gq):.1f})", flush=True)
    except Exception as e:  # a development aid: report, do not stop the other lines
        print(f"{name:44s} hipGraph replay failed: {type(e).__name__}: {e}", flush=True)
    print(f"{name:44s} b={b:3d} hn={hn:5d} VotePlan: {dt * 1e6:8.1f} us/call back to back (host issue {t_issue * 1e6:6.1f} us), "
          f"GPU latency of one call from an idle stream {np.median(lat):6.1f} us (min {min(lat):.1f}), queued behind GPU work "
          f"{np.median(qlat):6.1f} us (min {min(qlat):.1f})", flush=True)


run("headline (cfg 3): R=40 int64  [exact, default]", 32, 40, 1024, 0.99)
run("headline, approximate mode", 32, 40, 1024, 0.99, approx=True)
run("uint8 mask", 32, 40, 1024, 0.99, mask_dtype=torch.uint8)
run("thresh 0.999", 32, 40, 1024, 0.999)
run("thresh 0.999, approximate mode", 32, 40, 1024, 0.999, approx=True)
run("thresh 0.9", 32, 40, 1024, 0.9)
run("stress: R=97 (tn~29.5k)", 32, 97, 1024, 0.99)
run("stress: R=97, approximate mode", 32, 97, 1024, 0.99, approx=True)
run("demo call site: b=1 hn=512", 1, 27, 512, 0.99)
run("eval call site: b=1 hn=128 max_num=100", 1, 40, 128, 0.99, max_num=100)
run_plan("demo call site: b=1 hn=512", 1, 27, 512, 0.99)
run_plan("eval call site: b=1 hn=128 max_num=100", 1, 40, 128, 0.99, max_num=100)
run("batch 8", 8, 40, 1024, 0.99)
run("batch 128", 128, 40, 1024, 0.99, steps=10)
run("literal mode (reference fp32 order)", 32, 40, 1024, 0.99, literal=True, steps=5)
run("hn=4096 (distribution estimate), b=8", 8, 40, 4096, 0.99, steps=10)
run("hn=256, b=32", 32, 40, 256, 0.99)


# ---- the sibling entry points of SURVEY.md 8(f), timed the same way ------------------------------------------------
def timeit(name, fn, b, steps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:72s} b={b:3d}: {dt * 1e6:8.1f} us/call  {b / dt:10.0f} images/s", flush=True)


B = 32
mask, planar, _ = synth.make_batch(B, radius=40, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev)
pl = torch.from_numpy(planar).to(dev)
v = synth.planar_to_vertex_view(pl)
seg = torch.stack([1.0 - m.float(), m.float()], 1).contiguous() + 0.1 * torch.randn(B, 2, 480, 640, device=dev)
timeit("argmax(seg_pred) + ransac_voting_layer_v3 (tools/demo.py:52-55)",
       lambda: voting.ransac_voting_layer_v3(torch.argmax(seg, 1), v, 1024, inlier_thresh=0.99, seed=1), B)
timeit("fused: ransac_voting_layer_v3_from_logits (f#2)",
       lambda: voting.ransac_voting_layer_v3_from_logits(seg, v, 1024, inlier_thresh=0.99, seed=1), B)
timeit("ransac_voting_layer_v5: v3 + confidence at 0.999 (f#3), max_num=30000",
       lambda: voting.ransac_voting_layer_v5(m, v, 1024, inlier_thresh=0.99, max_num=30000, seed=1), B)
kp = voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=1)
timeit("estimate_voting_distribution_with_mean: 4096 hypotheses + covariance (f#3)",
       lambda: voting.estimate_voting_distribution_with_mean(m, v, kp, seed=1), B, steps=20)
timeit("generate_hypothesis (py-level): all hypotheses + counts, 1024 (f#4)",
       lambda: voting.generate_hypothesis_counts(m, v, 1024, inlier_thresh=0.99, seed=1), B)
timeit("ransac_motion_voting (f#4)", lambda: voting.ransac_motion_voting(m, v), B)
kp_np, cov = kp[0].cpu().numpy().astype(np.float64), None
from pvnet_amd import pnp  # noqa: E402
X = np.random.default_rng(0).uniform(-0.08, 0.08, size=(9, 3))
pose = np.concatenate([pnp.rodrigues(np.array([0.3, -0.2, 0.1])), np.array([[0.02], [-0.03], [0.9]])], 1)
x2 = pnp.project(X, pose, pnp.LINEMOD_K) + np.random.default_rng(1).normal(size=(9, 2)) * 0.3
for backend in ("native", "scipy"):
    t0 = time.perf_counter()
    for _ in range(200):
        pnp.pnp(X, x2, pnp.LINEMOD_K, backend=backend)
    print(f"host pnp (DLT + LM, 9 points), backend={backend:7s}: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us/call",
          flush=True)
W = np.tile([1.0, 0.0, 1.0], (9, 1))
for backend in ("native", "scipy"):
    t0 = time.perf_counter()
    for _ in range(200):
        pnp.uncertainty_pnp(x2, W, X, pnp.LINEMOD_K, backend=backend)
    print(f"host uncertainty_pnp (f#3), backend={backend:7s}: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us/call",
          flush=True)
