#!/usr/bin/env python
"""bench.py -- RANSAC votings/s of the HIP voting layer on synthetic 480x640 fields (BASELINE.json config 3).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole voting path (mask + 9-key-point vector field -> 9 key-points) over one
batch of 32 synthetic images per GPU, inputs resident in HBM.  Steps are independent batches, so they are issued
round-robin on --streams HIP streams (default 6; measured 1/2/3/4/5/6/8/12/16 streams -> 0.143/0.118/0.113/0.122/
0.115/0.112/0.115/0.114/0.114 ms per batch, profiles/r01_streams_probe.txt): the matrix-pipe scoring kernel holds
12 of a CU's 32 wave slots and the next batches' small latency-bound stages run beside it; `single_stream` in the output is the same K steps
issued strictly one after the other.  With N > 1 every rank votes its own 32 images
(weak scaling, no data-path collective) and the step ends with the path's one real exchange: an RCCL
all-gather of the [32, 9, 2] key-points.  Rank 0 prints ONE JSON line.

Besides the contract's fields the line carries
  roofline      the dominant kernel (inlier scoring), compute bound, not HBM bound.  It runs the
                vote's two 3-term fp32 dot products as bf16x3 MFMAs (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
                `achieved` = the matrix flops executed for the algorithmic hn*vn*tn pair tests (2 MFMAs per
                32x32 tests = 64 flop per test) over the kernel's duration, `peak` = 2.5 PFLOP/s dense bf16;
                `algorithmic` restates it with SURVEY.md 8d's 12 fp32 flop per test against the 157.3 TFLOP/s
                fp32 vector peak, `issue_bound` against what the SIMD can issue (64 MFMA + 48 VALU cycles per
                1024 tests).  Duration measured live with hipEvents on the op's stream (pvnet_vote_v3_profiled).
  roofline_hbm  the whole path against the HBM roofline: algorithmic bytes (24 576 072 B per voting with the
                int64 mask) x votings / time of all seven launches, peak 8 TB/s.
  cpu_baseline  the plain-C restatement (oracle, OpenMP over all host cores) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pvnet_amd import synth, voting  # noqa: E402

H, W, VN, HN = 480, 640, 9, 1024
BATCH = 32
THRESH = 0.99
BYTES_PER_VOTING = H * W * 8 + H * W * VN * 2 * 4 + VN * 2 * 4  # 24 576 072 (SURVEY.md 8d, int64 mask)
FLOP_PER_PAIR = 12
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = f32 MFMA dense peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)
MFMA_FLOP_PER_PAIR = 2 * (2 * 32 * 32 * 16) / (32 * 32)  # two 32x32x16 MFMAs (cr, dt) per 32x32 pair tests = 64
ISSUE_CYCLES_PER_1024 = 2 * 32 + 24 * 2  # 2 MFMAs x 32 cycles + 24 VALU (1.5 per test per lane) x 2 cycles, per SIMD
PEAK_CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps issued before the W warmup steps so that short runs do not measure the GPU's "
                         "clock ramp from idle (a step is ~0.15 ms: K=50 alone is an 8 ms burst)")
    ap.add_argument("--radius", type=int, default=40, help="disk radius of the synthetic object mask (tn ~ pi r^2)")
    ap.add_argument("--buffers", type=int, default=2, help="distinct input sets cycled (2 x 786 MB > 256 MiB L3)")
    ap.add_argument("--clean", action="store_true", help="noise-free field (default: noisy, net-like background)")
    ap.add_argument("--streams", type=int, default=6,
                    help="HIP streams the steps are issued on round-robin (independent batches in flight; the scoring "
                         "kernel keeps 12 of a CU's 32 wave slots, so the next batch's small stages run beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def make_inputs(rank, nbuf, radius, noisy, dev):
    sets = []
    for s in range(nbuf):
        mask, planar, _ = synth.make_batch(BATCH, first_index=(rank * nbuf + s) * BATCH, h=H, w=W, vn=VN,
                                           radius=radius, noise=noisy, background="normal" if noisy else "zeros")
        m = torch.from_numpy(mask).to(dev)  # int64, as torch.argmax delivers it (tools/demo.py:52)
        p = torch.from_numpy(planar).to(dev)  # [b, 2vn, h, w] planar, as the backbone emits it
        sets.append((m, synth.planar_to_vertex_view(p), mask, planar))
    return sets


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_traffic.json, written
    by tools/rocpd_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"][kernel]
        return int((2.0 * k.get("fetch_kb", 0.0) + k.get("write_kb", 0.0)) * 1024)  # FETCH_SIZE x2: gfx950 correction
    except (KeyError, ValueError):
        return None


def usable_cores():
    """logical CPUs this process may actually use: affinity mask and cgroup CPU quota (the GPU boxes expose 256
    logical CPUs but cap the container at a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sets, seconds):
    """the oracle (plain-C port) on a bounded sample of the same workload -- reported, never the target.
    All cores: one image per worker thread (images are independent; ctypes releases the GIL; the C code runs
    single-threaded inside each worker), which scales far better than OpenMP inside one image.  Plus one core."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref
    from oracle import ransac_voting_oracle as O
    cref.build()
    _, _, mask, planar = sets[0]
    vnp = synth.planar_to_vertex_view(planar)
    fg = O.foreground(mask)
    cores = usable_cores()

    def one(i):
        cref.set_num_threads(1)  # per-thread OpenMP setting: this worker runs the C code on one core
        cref.vote_v3(fg[i % BATCH:i % BATCH + 1], vnp[i % BATCH:i % BATCH + 1], HN, THRESH, seed=1)

    one(0)  # warm
    t1 = time.perf_counter()
    n1 = 0
    while n1 < 2 or (time.perf_counter() - t1 < 3.0 and n1 < 8):
        one(n1)
        n1 += 1
    one_core = n1 / (time.perf_counter() - t1)
    t0 = time.perf_counter()
    deadline = t0 + seconds  # time-bounded: every worker keeps taking images until the deadline

    def worker(w):
        k = 0
        while time.perf_counter() < deadline:
            one(w + k)
            k += 1
        return k

    with ThreadPoolExecutor(max_workers=cores) as ex:
        n = sum(ex.map(worker, range(cores)))
    el = time.perf_counter() - t0
    return {"value": n / el, "unit": "votings/s", "cores": cores, "kind": "port", "value_1_core": one_core,
            "sample": f"{n} images of the bench workload (480x640, 9 kpts, 1024 hyp) in {el:.1f} s on {cores} worker "
                      f"threads (one image each, oracle/oracle_c/pvnet_vote_ref.c; host shows {os.cpu_count()} logical "
                      f"cpus, {cores} usable under its cgroup quota / affinity); {n1} images on one core"}


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to stdout when its communicator comes up, so
    # fd 1 is pointed at stderr for the run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # under torch.distributed.run (RANK + MASTER_ADDR set) the process group and the key-point all-gather are used at
    # any world size, so that a 1-rank launch exercises exactly the code path the N-rank launches run
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    voting.load_library()

    sets = make_inputs(rank, a.buffers, a.radius, not a.clean, dev)
    # S streams, S + 1 gather targets: the 2.3 KB all-gather of step i (RCCL stream) overlaps the voting of the
    # following steps; a target is reused only after the gather that last wrote it has been waited for
    nstreams = max(1, a.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    gathered = [torch.empty((world * BATCH, VN, 2), dtype=torch.float32, device=dev) for _ in range(nstreams + 1)] \
        if dist is not None else None
    pending = []

    def step(i, ns=nstreams, **kw):
        m, v, _, _ = sets[i % len(sets)]
        if kw:  # profiled / debug calls: current stream, synchronising
            return voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=1234 + i,
                                                 image_offset=rank * BATCH, **kw)
        with torch.cuda.stream(streams[i % ns]):
            out = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=1234 + i,
                                                image_offset=rank * BATCH)
            if dist is not None:
                # the path's single exchange: RCCL all-gather of the [32, 9, 2] key-points over xGMI
                pending.append(dist.all_gather_into_tensor(gathered[i % (nstreams + 1)], out, async_op=True))
                if len(pending) > ns:
                    pending.pop(0).wait()
        return out

    def fence():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(ns):
        for i in range(a.warmup):
            step(i, ns)
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i, ns)
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    t_pre = time.perf_counter()                      # device pre-warm (untimed, reported in config.prewarm_s)
    i_pre = 0
    while time.perf_counter() - t_pre < a.prewarm_seconds:
        for _ in range(20):  # voting only: a time-bounded loop must not issue collectives (ranks would disagree on the count)
            m, v, _, _ = sets[i_pre % len(sets)]
            with torch.cuda.stream(streams[i_pre % nstreams]):
                voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=i_pre, image_offset=rank * BATCH)
            i_pre += 1
        torch.cuda.synchronize(dev)
    fence()
    dt = timed(nstreams)                             # the headline: K steps, independent batches on S streams
    dt1 = timed(1) if nstreams > 1 else dt           # the same K steps strictly one after the other (latency view)

    # ---- roofline of the dominant kernel: live hipEvent stage timing over the same K steps ------------------
    stage_sum = {}
    tn_sum = 0
    nprof = min(a.steps, 200)  # synchronising, profiled calls: 200 are plenty for an average
    for i in range(nprof):
        _, dbg, times = step(i, return_debug=True, stage_times=True)
        if i < len(sets):
            tn_sum += int(dbg["tn"].sum().item())
        for k, v in times.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    stage_ms = {k: v / nprof for k, v in stage_sum.items()}
    tn_per_batch = tn_sum / min(nprof, len(sets))
    pairs = HN * VN * tn_per_batch  # pair tests per launch of the scoring kernel
    score_s = stage_ms["score"] * 1e-3
    path_s = sum(stage_ms.values()) * 1e-3

    if rank == 0:
        votings_per_s = world * BATCH * a.steps / dt
        res = {
            "metric": "RANSAC votings/s (480x640, 9 kpts, batch 32) + HBM GB/s vs roofline",
            "value": votings_per_s, "unit": "votings/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "single_stream": {"value": world * BATCH * a.steps / dt1, "ms_per_step": dt1 / a.steps * 1e3,
                              "note": "the same K steps issued on one stream: per-batch latency of the whole path"},
            "dtype": "bf16x3 products, f32 accumulate (f32-equivalent; refinement f64)", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: batch=32 synthetic 480x640 fields per GPU, 9 keypoints, "
                                   "1024 hypotheses, inlier_thresh 0.99, int64 mask, planar strided field",
                       "batch_per_gpu": BATCH, "global_batch": world * BATCH, "h": H, "w": W, "vn": VN, "hn": HN,
                       "mask_radius": a.radius, "mean_foreground_px": tn_per_batch / BATCH,
                       "field": "clean" if a.clean else "noisy (0.05 rad + 10% outliers), N(0,1) background",
                       "input_sets_cycled": len(sets), "streams": nstreams, "prewarm_s": a.prewarm_seconds,
                       "parallelism": f"images sharded over {world} GPU(s); steps issued round-robin on {nstreams} "
                                      f"HIP stream(s) per GPU"},
            "roofline": {"kernel": "score_mfma_kernel", "bound": "mfma",
                         "achieved": MFMA_FLOP_PER_PAIR * pairs / score_s / 1e12, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": MFMA_FLOP_PER_PAIR * pairs / score_s / 1e12 / PEAK_BF16_TFLOPS,
                         "traffic": measured_traffic("score_mfma_kernel"),
                         "avg_launch_ms": stage_ms["score"], "pair_tests_per_launch": pairs,
                         "flop_per_pair_executed": MFMA_FLOP_PER_PAIR,
                         "algorithmic": {"flop_per_pair": FLOP_PER_PAIR,
                                         "tflops": FLOP_PER_PAIR * pairs / score_s / 1e12,
                                         "vs_fp32_vector_peak": FLOP_PER_PAIR * pairs / score_s / 1e12 / PEAK_F32_TFLOPS},
                         "issue_bound": {"cycles_per_1024_pairs_per_simd": ISSUE_CYCLES_PER_1024,
                                         "pairs_per_s": 1024 / ISSUE_CYCLES_PER_1024 * PEAK_CLOCK_HZ * N_SIMD,
                                         "frac": pairs / score_s / (1024 / ISSUE_CYCLES_PER_1024 * PEAK_CLOCK_HZ * N_SIMD)},
                         "note": "each pair test = two 3-term fp32 dot products + compare; operands split into three "
                                 "bf16 parts, six part products kept per product (K = 15 of 16), fp32 accumulation on "
                                 "the matrix pipe; 1.5 VALU ops per test count the votes.  Matrix and vector issue do "
                                 "not overlap within a SIMD for this mix (tools/ubench_mfma.hip), hence issue_bound"},
            "roofline_hbm": {"bound": "hbm", "achieved": BYTES_PER_VOTING * BATCH / path_s / 1e9,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": BYTES_PER_VOTING * BATCH / path_s / 1e9 / PEAK_HBM_GBS,
                             "bytes_per_voting": BYTES_PER_VOTING, "path_ms": path_s * 1e3,
                             "end_to_end_frac": BYTES_PER_VOTING * votings_per_s / world / 1e9 / PEAK_HBM_GBS},
            "stage_ms": stage_ms,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(sets, a.cpu_seconds)
            except Exception as e:  # the reported baseline must never take the measured line down with it
                res["cpu_baseline"] = {"value": None, "unit": "votings/s", "cores": usable_cores(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
