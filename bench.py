#!/usr/bin/env python
"""bench.py -- RANSAC votings/s of the HIP voting layer on synthetic 480x640 fields (BASELINE.json configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches ITSELF under torch.distributed.run (one
rank per GPU, rendezvous on 127.0.0.1), so `python bench.py --gpus 8` and the explicit launcher line are the same run.

A "step" is one pass of the whole voting path (mask + 9-key-point vector field -> 9 key-points) over one batch of 32
synthetic images per GPU, inputs resident in HBM.  The timed mode is the library's DEFAULT: exact mode -- matrix-pipe scoring
whose inlier counts and winners EQUAL the reference kernels' (pvnet_vote.h, PVNET_F_LITERAL / PVNET_F_APPROX).
The timed region -- barrier + synchronise, EXACTLY K steps, synchronise [each rank's clock stops when ITS K steps and its
last gather are done] + barrier, MAX of the ranks' times -- is run --regions
times (default 15) and the MEDIAN region is reported as `value` / `ms_per_step` (a region of 20 steps lasts ~3 ms: one late
stream moves a single region by several per cent); `regions` carries every region's rate, min, max and spread, and the
GPU's clock and package power sampled while the regions ran.  Steps are independent batches, so they are issued round-robin on
--streams HIP streams (default 6); `single_stream` in the output is the same K steps issued strictly one after the
other (the per-batch latency a caller with ONE frame in flight sees).  With N > 1 every rank votes its own 32 images
(weak scaling, no data-path collective) and its key-points leave through the path's one real exchange, an RCCL
all-gather of [32, 9, 2] per step -- bucketed: the key-points of --gather-bucket consecutive steps (default: one per
stream) are voted straight into a staging block and travel in ONE collective.  Rank 0 prints ONE JSON line.

Besides the contract's fields the line carries
  roofline      the dominant kernel (inlier scoring, score_exact_kernel), compute bound.  Its duration is
                measured live over >= 200 launches on the op's stream (pvnet_vote_v3_stage_repeat): every workgroup
                stamps the device's constant-rate clock, max end - min start per launch = what a kernel trace reports
                for it, free of launch gaps (the same launches between ONE hipEvent pair are reported beside it).
                `achieved` / `peak` / `frac` follow SURVEY.md 8d: ALGORITHMIC flops (12 fp32 flop per pair test x
                `pair_tests_per_launch`) / `avg_launch_ms`, against the dense peak of the pipe the kernel runs on (bf16
                MFMA, 2.5 PFLOP/s) -- recomputable from those three fields.  Side fields: `executed_*` (the matrix
                flops actually issued: the bf16x3 split costs two v_mfma_f32_32x32x16_bf16 per 32x32 tests = 64 flop
                per test), `vs_fp32_vector_peak` (the same algorithmic rate against the 157.3 TFLOP/s the path left),
                `mfma_busy_frac` (matrix-pipe busy cycles / SIMD cycles from a committed PMC pass) and `traffic` (HBM
                bytes per launch from committed FETCH_SIZE / WRITE_SIZE passes).  Everything that is read from a
                committed profile instead of being measured in this run carries `*_from_committed_profile: <tag>`
                next to the value and is null when that profile was taken from other kernel sources than this build.
  roofline_hbm  the whole path against HBM: `compulsory_bytes` (what any implementation must read: the masks + the
                foreground vectors), `measured_bytes` (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE summed over the path's
                kernels, profiles/*_traffic.json), and SURVEY.md 8d's dense-equivalent figure (the bytes a dense
                implementation would stream) -- three different numerators over the same step time.
  parity        the TIMED mode checked on the TIMED inputs after the timed region: every one of the 1024 inlier counts of
                every key-point against literal mode (`counts_equal_literal`) and, for the first images, against the
                REFERENCE'S OWN voting kernel compiled for gfx950 (`counts_equal_reference`, oracle/_ref); winners against
                literal mode and the plain-C oracle; key-points against the C oracle (all) and the float64 oracle (first
                set).  The oracle only checks; it is never timed here.
  approx_mode   votings/s of PVNET_F_APPROX (the round-1/2 "fast" mode: no rounding-band re-evaluation, counts within a few
                votes of the reference's) on the same inputs and streams -- what exactness costs.
  literal_mode  votings/s of PVNET_F_LITERAL (the reference's float32 order for every pair on the VALU) on the same inputs.
  secondary     the configurations the reference itself calls with (its default thresh 0.999, ~29.5 k-pixel objects, the two
                single-frame call sites of tools/demo.py and tools/train_linemod.py): votings/s on one stream, the scoring stage's
                time and each call's own counts-equal-literal check.  Never part of `value`.
  cpu_baseline  the plain-C restatement (oracle) timed on a bounded sample of the same workload.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
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
SEED0 = 1234  # step i votes with seed SEED0 + i on input set i % buffers
BYTES_PER_VOTING = H * W * 8 + H * W * VN * 2 * 4 + VN * 2 * 4  # 24 576 072 (SURVEY.md 8d, int64 mask): dense-equivalent
FLOP_PER_PAIR = 12
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = f32 MFMA dense peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD)
MFMA_FLOP_PER_PAIR = 2 * (2 * 32 * 32 * 16) / (32 * 32)  # two 32x32x16 MFMAs (cr, dt) per 32x32 pair tests = 64
N_SIMD = 256 * 4
N_XCD = 8
SCORE_KERNEL = "score_exact_kernel"
PATH_KERNELS = ("mask_bits_kernel", "compact_kernel", "hypothesis_kernel", SCORE_KERNEL, "select_refine_kernel")
KERNEL_SOURCES = tuple("pvnet_amd/csrc/" + f for f in (
    "vote_common.h", "k1_mask.hip", "k2_compact.hip", "k3_hypotheses.hip", "k4_score_valu.hip", "k4_score_mfma.hip", "k4_exact_body.h",
    "k4_score_exact.hip", "k4_score_cull.hip", "k5_refine.hip", "epilogues.hip", "vote_host.hip", "pvnet_rng.h")) + ("include/pvnet_vote.h",)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps issued before the W warmup steps so that short runs do not measure the GPU's "
                         "clock ramp from idle (a step is ~0.12 ms: K=20 alone is a 2.4 ms burst)")
    ap.add_argument("--radius", type=int, default=40, help="disk radius of the synthetic object mask (tn ~ pi r^2)")
    ap.add_argument("--regions", type=int, default=15,
                    help="how many times the timed K-step region is run; the MEDIAN region is reported (min / max / every "
                         "region's rate alongside)")
    ap.add_argument("--buffers", type=int, default=4,
                    help="distinct input sets cycled: the TOUCHED bytes of a set are its masks (78.6 MB) + its foreground "
                         "vectors (~11.6 MB), so 4 sets = 361 MB exceed the 256 MiB Infinity Cache")
    ap.add_argument("--clean", action="store_true", help="noise-free field (default: noisy, net-like background)")
    ap.add_argument("--streams", type=int, default=6,
                    help="HIP streams the steps are issued on round-robin (independent batches in flight)")
    ap.add_argument("--gather-bucket", type=int, default=0,
                    help="steps whose key-points travel in ONE all-gather (N > 1 or under torch.distributed.run); 0 = the "
                         "number of streams")
    ap.add_argument("--gather", choices=("rccl", "torch"), default=os.environ.get("BENCH_GATHER", "rccl"),
                    help="the all-gather of key-points under torch.distributed.run: 'rccl' = the library's own ncclAllGather on the "
                         "voting stream that filled the bucket (pvnet_vote_allgather; torch.distributed only bootstraps the "
                         "communicator), 'torch' = torch.distributed.all_gather_into_tensor on a communication stream (rounds 1-5)")
    ap.add_argument("--score-repeats", type=int, default=200, help="back-to-back scoring launches timed for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-parity", action="store_true", help="skip the post-run parity check of the timed mode")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (the reference's own call configurations, a few seconds)")
    ap.add_argument("--approx", action="store_true",
                    help="DEVELOPMENT: time PVNET_F_APPROX as the main mode (the line's `mode` says so); for A/B runs against "
                         "earlier rounds, whose default mode this was")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY (tests/test_bench_contract.py): gloo on CPU with a stub voter, to exercise the "
                         "launcher / process-group / gather plumbing without a GPU; the line says so and is no measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv):
    """--gpus N > 1 outside torch.distributed.run: run this very script under it, one rank per GPU.  The ranks inherit
    stdout, rank 0 writes the single JSON line; the launcher's own chatter goes to stderr."""
    a = parse(argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------------------------- inputs
def make_inputs(rank, nbuf, radius, noisy, dev):
    sets = []
    for s in range(nbuf):
        mask, planar, _ = synth.make_batch(BATCH, first_index=(rank * nbuf + s) * BATCH, h=H, w=W, vn=VN,
                                           radius=radius, noise=noisy, background="normal" if noisy else "zeros")
        m = torch.from_numpy(mask).to(dev)  # int64, as torch.argmax delivers it (tools/demo.py:52)
        p = torch.from_numpy(planar).to(dev)  # [b, 2vn, h, w] planar, as the backbone emits it
        sets.append((m, synth.planar_to_vertex_view(p), mask, planar))
    return sets


def source_hash():
    """sha1 over the kernel sources: a committed profile is only quoted when it was taken from these very sources"""
    import hashlib
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def newest_profile(suffix):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)))  # rNNx tags sort by round, then letter
    return files[-1] if files else None


def committed_profile(suffix):
    """(tag, kernels dict, fresh) of the newest committed profiles/*<suffix>; fresh = taken from this build's sources.
    Kernels are keyed by their FULL template instantiation (tools/rocpd_summary.py)."""
    f = newest_profile(suffix)
    if not f:
        return None, {}, False
    try:
        j = json.load(open(f))
    except ValueError:
        return None, {}, False
    tag = os.path.basename(f)[:-len(suffix)]
    return tag, j.get("kernels", {}), j.get("source_hash") == source_hash()


KERNEL_VARIANTS = {"mask_bits_kernel": ("mask_bits_kernel", "mask_bits_pair_kernel")}  # (round 5: aligned int64 masks take two pixels per 16-byte load)


def pick_instantiation(kernels, base):
    """the instantiation of `base` with the most dispatches in the profiled run (the run also makes a few literal-mode
    calls, whose instantiations must not be mistaken for the timed ones)"""
    best = None
    names = KERNEL_VARIANTS.get(base, (base,))
    for name, k in kernels.items():
        if name.split("<")[0] in names and (best is None or k.get("n", 0) > kernels[best].get("n", 0)):
            best = name
    return best


def measured_traffic(base):
    """HBM bytes per launch of kernel `base` from the newest committed PMC summary (profiles/*_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); (None, tag) when absent or taken from other sources."""
    tag, kernels, fresh = committed_profile("_traffic.json")
    name = pick_instantiation(kernels, base)
    if not fresh or name is None:
        return None, tag, name
    k = kernels[name]
    return int((2.0 * k.get("fetch_kb", 0.0) + k.get("write_kb", 0.0)) * 1024), tag, name  # FETCH_SIZE x2: gfx950 correction


def measured_mfma_busy(base=SCORE_KERNEL):
    """matrix-pipe busy fraction of `base` from the newest committed counter summary (profiles/*_pmc.json):
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's SIMDs) / (SIMDs x GRBM_GUI_ACTIVE per XCD)."""
    tag, kernels, fresh = committed_profile("_pmc.json")
    name = pick_instantiation(kernels, base)
    if not fresh or name is None:
        return None, tag
    try:
        k = kernels[name]
        return float(k["SQ_VALU_MFMA_BUSY_CYCLES"]) / (N_SIMD * float(k["GRBM_GUI_ACTIVE"]) / N_XCD), tag
    except (KeyError, ValueError, ZeroDivisionError):
        return None, tag


class GpuSampler:
    """shader clock and package power of one GPU, sampled from sysfs (amdgpu hwmon) by a thread while the timed regions
    run -- no subprocess, nothing on the GPU.  Absent files (other driver versions) simply give no samples."""

    def __init__(self, index, period=0.005):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.freq = self.power = None
        mons = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            try:
                if open(os.path.join(d, "name")).read().strip() == "amdgpu":
                    mons.append(d)
            except OSError:
                pass
        if mons:
            d = mons[index % len(mons)]
            for f in ("freq1_input",):
                if os.path.exists(os.path.join(d, f)):
                    self.freq = os.path.join(d, f)
            for f in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(d, f)):
                    self.power = os.path.join(d, f)
                    break
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((self._read(self.freq) if self.freq else None, self._read(self.power) if self.power else None))
            time.sleep(self.period)

    def __enter__(self):
        if os.environ.get("BENCH_NO_SAMPLER") != "1":   # (development A/B: does reading the sensors disturb short regions?)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=1.0)

    def summary(self):
        def stat(xs, scale):
            xs = [x * scale for x in xs if x is not None]
            return {"mean": sum(xs) / len(xs), "min": min(xs), "max": max(xs), "samples": len(xs)} if xs else None
        return {"sclk_mhz": stat([s[0] for s in self.samples], 1e-6), "power_w": stat([s[1] for s in self.samples], 1e-6),
                "source": "amdgpu hwmon (sysfs freq1_input / power1_average), sampled every "
                          f"{self.period * 1e3:.0f} ms while the timed regions ran"}


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


# ----------------------------------------------------------------------------------------------------- CPU legs
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


def parity_check(sets, rank, o64_images=BATCH, ref_images=8, max_sets=2):
    """The timed mode on the timed inputs, checked after the timed region (the oracle is the CHECKER here, nothing of
    it is timed or shipped).  Step s (s = 0 .. max_sets-1) is exactly the s-th timed step: input set s, seed SEED0 + s,
    this rank's image offset.
      * integer products: all HN inlier counts of every (image, key-point) of the timed (exact) mode against literal mode
        (`counts_equal_literal`: how many of the (image, key-point) count vectors are EQUAL) and, for the first
        `ref_images` images of set 0, against the reference's OWN voting kernel compiled for gfx950 (oracle/_ref,
        `counts_equal_reference`); winners against literal mode and the plain-C oracle;
      * key-points: against the C oracle (float32 votes, float64 least squares) everywhere and against the float64 numpy
        oracle on the first `o64_images` images of set 0.  north_star tolerance: 1e-3 px."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref, refkernels
    from oracle import ransac_voting_oracle as O
    cref.build()
    cores = usable_cores()
    tot = fl_eq = lo_eq = fo_eq = cnt_eq = conc_eq = 0
    max_px_c = max_px_lit = 0.0
    max_count_diff = 0
    ref_eq = ref_tot = 0
    for s, (m, v, mask, planar) in enumerate(sets[:max_sets]):
        fast, df = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + s,
                                                 image_offset=rank * BATCH, return_debug=True, concurrent=False)
        fast = fast.cpu().numpy()
        wf = df["win"].cpu().numpy().copy()
        cf = df["counts"].clone()
        # the kernel variant the multi-stream regions ran (PVNET_F_CONCURRENT): the same integers, the same key-points
        conc, dc = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + s,
                                                 image_offset=rank * BATCH, return_debug=True, concurrent=True)
        conc_eq += int((cf == dc["counts"]).all(2).sum()) if bool((torch.from_numpy(fast).to(conc.device) == conc).all()) else 0
        if s == 0 and ref_images > 0 and refkernels.available("off"):
            for bi in range(min(ref_images, BATCH)):  # the reference's kernel on the path's own compacted pixels
                tn = int(df["tn"][bi])
                rec = df["rec"][bi, :, :tn]
                coords = rec[0, :, 0:2].contiguous()
                direct = rec[:, :, 2:4].permute(1, 0, 2).contiguous()
                hyp = df["hyp"][bi].permute(1, 0, 2).contiguous()
                cref_counts = torch.zeros((HN, VN), dtype=torch.int32, device=m.device)
                stp = max(1, (1 << 27) // (VN * max(tn, 1)))
                for h0 in range(0, HN, stp):
                    inl = refkernels.voting_for_hypothesis(direct, coords, hyp[h0:h0 + stp].contiguous(), THRESH)
                    cref_counts[h0:h0 + stp] = inl.sum(2, dtype=torch.int32)
                ref_eq += int((cf[bi].T == cref_counts).all(0).sum())
                ref_tot += VN
        lit, dl = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + s,
                                                image_offset=rank * BATCH, literal=True, return_debug=True)
        lit = lit.cpu().numpy()
        wl = dl["win"].cpu().numpy().copy()
        cnt_eq += int((cf == dl["counts"]).all(2).sum())
        max_count_diff = max(max_count_diff, int((cf - dl["counts"]).abs().max()))
        vnp = synth.planar_to_vertex_view(planar)
        fg = O.foreground(mask)

        def one(i):  # images are independent: one per worker thread, each on its global RNG stream
            cref.set_num_threads(1)
            return cref.vote_v3(fg[i:i + 1], vnp[i:i + 1], HN, THRESH, seed=SEED0 + s, return_winners=True,
                                image_base=rank * BATCH + i)

        with ThreadPoolExecutor(max_workers=cores) as ex:
            res = list(ex.map(one, range(BATCH)))
        ref = np.concatenate([r[0] for r in res])
        wi = np.concatenate([r[1] for r in res])
        n = wi.size
        tot += n
        fl_eq += int((wf[:, :, 0] == wl[:, :, 0]).sum())
        lo_eq += int((wl[:, :, 0] == wi).sum())
        fo_eq += int((wf[:, :, 0] == wi).sum())
        max_px_c = max(max_px_c, float(np.abs(fast - ref).max()))
        max_px_lit = max(max_px_lit, float(np.abs(fast - lit).max()))
    out = {"keypoints_checked": tot, "hypotheses_per_keypoint": HN, "counts_equal_concurrent_variant": conc_eq,
           "counts_equal_literal": cnt_eq, "max_count_diff_vs_literal": max_count_diff,
           "counts_equal_reference": ref_eq if ref_tot else None, "reference_keypoints_checked": ref_tot,
           "reference": "oracle/_ref/libpvnet_refkernels.so: ransac_voting_kernel.cu:88-126 compiled for gfx950 from the "
                        "reference tree" if ref_tot else "oracle/_ref not built on this box",
           "winners_equal_literal": fl_eq, "literal_winners_equal_c_oracle": lo_eq, "winners_equal_c_oracle": fo_eq,
           "winners_equal": bool(tot and fl_eq == tot and lo_eq == tot),
           "max_px_vs_c_oracle": max_px_c, "max_px_vs_literal": max_px_lit, "tolerance_px": 1e-3}
    if o64_images > 0:
        m, v, mask, planar = sets[0]
        k = min(o64_images, BATCH)
        o64 = O.ransac_voting_layer_v3(mask[:k], synth.planar_to_vertex_view(planar[:k]), HN, inlier_thresh=THRESH,
                                       seed=SEED0, image_offset=rank * BATCH)
        fast = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0,
                                             image_offset=rank * BATCH).cpu().numpy()
        out["max_px_vs_oracle64"] = float(np.abs(fast[:k] - o64).max())
        out["oracle64_images"] = k
    out["pass"] = bool(out["winners_equal"] and cnt_eq == tot and conc_eq == tot and (ref_tot == 0 or ref_eq == ref_tot) and
                       max_px_c <= 1e-3 and out.get("max_px_vs_oracle64", 0.0) <= 1e-3)
    out["mode"] = "exact (the library's default: bf16x3 MFMA scoring + literal re-evaluation inside the rounding band): " \
                  "the mode the timed region ran"
    return out


def secondary_block(sets, rank, dev, budget_s=3.0):
    """The configurations the REFERENCE itself calls with, in the driver's line -- not the headline, never
    `value`: (1) the reference's default inlier_thresh 0.999 (ransac_voting_gpu.py:514) on the headline's batch; (2) objects of
    ~29.5 k pixels (R = 97, just under max_num = 30 000); (3) tools/demo.py:55's call, one frame, 512 hypotheses; (4)
    tools/train_linemod.py:106's call, one frame, 128 hypotheses, max_num = 100; (5, 6) fields the library's selection disc-culls:
    the headline's masks with the ground-truth field, the reference's demo fixture 32 times -- `pair_tests_executed` beside
    `pair_tests_per_launch`; (7) the fused arg-max entry against argmax + v3.  Each: the DEFAULT (exact) mode through a
    prepared VotePlan on ONE stream, calls issued back to back (what a caller with one frame in flight sees), the scoring stage's
    own time from an event pair, and the call's own parity: all inlier counts and winners against literal mode on the same
    inputs and draw.  Time-bounded: the timed calls of the entries together stay below `budget_s` seconds of GPU time."""
    out = {"note": "one stream, default (exact) mode, VotePlan (no per-call allocation); parity = this call's counts / winners "
                   "against literal mode (the reference's float32 order for every pair) on the same inputs and draw",
           "entries": {}}
    t_all = time.perf_counter()

    def entry(name, ref, m, v, hn, thresh, max_num, steps):
        b = int(m.shape[0])
        plan = voting.VotePlan(m, v, hn, inlier_thresh=thresh, max_num=max_num)
        for i in range(5):
            plan(m, v, seed=i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        plan(m, v, seed=0)
        torch.cuda.synchronize(dev)
        one = max(time.perf_counter() - t0, 1e-5)
        steps = int(max(10, min(steps, budget_s / 6 / one)))   # a sixth of the budget each
        t0 = time.perf_counter()
        for i in range(steps):
            plan(m, v, seed=SEED0 + i)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        # (on the plan's workspace: whether a batch may be disc-culled follows the previous call on the SAME workspace, vote_common.h)
        _, dbg, st = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=SEED0, workspace=plan.workspace,
                                                   image_offset=rank * BATCH, return_debug=True, stage_times=True, concurrent=False)
        counts, win = dbg["counts"].clone(), dbg["win"].clone()
        tn = float(dbg["tn"].float().mean())
        # (the statistics come from a call of their own: every wave adds its step counts to ONE word, 24 k same-address atomics that
        #  took the scoring stage of the clean field from 55 to 419 us when they shared the event-timed call -- r06n)
        _, dbg = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=SEED0, image_offset=rank * BATCH,
                                               return_debug=True, concurrent=False, band_stats=True, workspace=plan.workspace)
        # disc culling (the library selects it per image on the device): the share of (image, key-point)s it culled, and the pair tests
        # the launch really EXECUTED -- all of the dense key-points', of the culled ones the fine pass's share (steps executed / steps
        # of the dense kernel) plus the coarse pass (every pixel against the 32 tile centres of a slice)
        share = float(dbg["cull_bits"].float().mean())
        ex, full = dbg["cull_stats"]
        fine = ex / full if full else 0.0
        _, dl = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=thresh, max_num=max_num, seed=SEED0,
                                              image_offset=rank * BATCH, literal=True, return_debug=True)
        pairs = hn * VN * tn * b
        executed = pairs * (1.0 - share) + share * (pairs * fine + VN * tn * b * 32.0 * ((hn + 1023) // 1024))
        out["entries"][name] = {
            "reference_call": ref, "batch": b, "hn": hn, "inlier_thresh": thresh, "max_num": max_num, "mean_kept_px": tn,
            "us_per_call": dt * 1e6, "votings_per_s": b / dt, "calls_timed": steps,
            "score_us": st["score"] * 1e3, "pair_tests_per_s_in_score": pairs / max(st["score"] * 1e-3, 1e-9),
            "pair_tests_per_launch": pairs, "pair_tests_executed": executed, "share_disc_culled": share,
            "stage_us": {k: x * 1e3 for k, x in st.items()},
            "counts_equal_literal": int((counts == dl["counts"]).all(2).sum()), "keypoints_checked": b * VN,
            "winners_equal_literal": int((win == dl["win"]).all(2).sum()),
            "pass": bool((counts == dl["counts"]).all() and (win == dl["win"]).all())}

    m0, v0, _, _ = sets[0]
    entry("thresh_0.999_batch32", "ransac_voting_gpu.py:514 (the layer's default inlier_thresh)", m0, v0, HN, 0.999, 30000, 400)
    mask, planar, _ = synth.make_batch(BATCH, first_index=7000, h=H, w=W, vn=VN, radius=97, noise=True, background="normal")
    mb = torch.from_numpy(mask).to(dev)
    vb = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    entry("object_29k_px_batch32", "SURVEY 8(d) stress shape: R = 97, tn ~ 29.5 k, just under max_num", mb, vb, HN, THRESH, 30000, 100)
    del mb, vb
    mask, planar, _ = synth.make_batch(1, first_index=7100, h=H, w=W, vn=VN, radius=27, noise=True, background="normal")
    m1 = torch.from_numpy(mask).to(dev)
    v1 = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    entry("demo_call_site_b1_hn512", "tools/demo.py:55: ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)",
          m1, v1, 512, 0.99, 30000, 2000)
    mask, planar, _ = synth.make_batch(1, first_index=7101, h=H, w=W, vn=VN, radius=40, noise=True, background="normal")
    m2 = torch.from_numpy(mask).to(dev)
    v2 = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    entry("eval_call_site_b1_hn128_max100", "tools/train_linemod.py:106: ransac_voting_layer_v3(mask, vertex, 128, "
          "inlier_thresh=0.99, max_num=100)", m2, v2, 128, 0.99, 100, 2000)
    del m1, v1, m2, v2
    # VERDICT r05 "Next" 1: fields on which the library's own selection disc-culls -- the ground-truth field of the headline's masks, and
    # the reference's demo fixture (tests/golden/demo_cat.npz: a real mask, the field of its projected key-points) 32 times
    mask, planar, _ = synth.make_batch(BATCH, first_index=0, h=H, w=W, vn=VN, radius=40, noise=False, background="zeros")
    mc = torch.from_numpy(mask).to(dev)
    vc = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    entry("clean_field_batch32", "the headline's masks with the ground-truth field (linemod_dataset.py:68-81), no noise",
          mc, vc, HN, THRESH, 30000, 400)
    del mc, vc
    demo = os.path.join(ROOT, "tests", "golden", "demo_cat.npz")
    if os.path.exists(demo) and (H, W) == (480, 640):
        g = np.load(demo)
        dh, dw = (int(x) for x in g["shape"])
        dm = np.unpackbits(g["mask_bits"])[: dh * dw].reshape(dh, dw)
        dp = synth.field_from_keypoints(dm.astype(bool), g["points_2d"])
        md = torch.from_numpy(np.repeat(dm[None].astype(np.int64), BATCH, 0)).to(dev)
        vd = synth.planar_to_vertex_view(torch.from_numpy(np.repeat(dp[None], BATCH, 0)).to(dev))
        entry("demo_field_batch32", "the reference's demo fixture (data/demo: cat mask, ground-truth field of its key-points), "
              "32 copies", md, vd, HN, THRESH, 30000, 400)
        del md, vd
    # VERDICT r05 "Next" 6c: the fused arg-max entry (tools/demo.py:52 + :55 in one call) against argmax + v3 on the headline's batch
    seg = torch.stack([0.5 - m0.float(), m0.float() - 0.5], 1).contiguous()
    ws = torch.empty(voting.vote_layout(BATCH, H, W, VN, HN, 30000).total_bytes, dtype=torch.uint8, device=dev)
    res = {}
    for name, fn in (("argmax_then_v3", lambda i: voting.ransac_voting_layer_v3(torch.argmax(seg, 1), v0, HN, inlier_thresh=THRESH, seed=i,
                                                                              workspace=ws)),
                     ("logits_entry", lambda i: voting.ransac_voting_layer_v3_from_logits(seg, v0, HN, inlier_thresh=THRESH, seed=i,
                                                                                        workspace=ws))):
        for i in range(5):
            r = fn(SEED0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(200):
            fn(SEED0 + i)
        torch.cuda.synchronize(dev)
        res[name] = ((time.perf_counter() - t0) / 200 * 1e6, r.clone())
    out["entries"]["logits_entry_batch32"] = {
        "reference_call": "tools/demo.py:52,55: torch.argmax(seg_pred, 1) then the layer; pvnet_vote_v3_logits takes the class "
                          "logits in place", "batch": BATCH, "hn": HN, "inlier_thresh": THRESH,
        "us_per_call_argmax_then_v3": res["argmax_then_v3"][0], "us_per_call": res["logits_entry"][0], "calls_timed": 200,
        "keypoints_equal": bool(torch.equal(res["argmax_then_v3"][1], res["logits_entry"][1])),
        "pass": bool(torch.equal(res["argmax_then_v3"][1], res["logits_entry"][1]))}
    out["pass"] = all(e["pass"] for e in out["entries"].values())
    out["wall_s"] = time.perf_counter() - t_all
    return out


# ------------------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(argv))
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
        if a.stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if a.stub:
        return stub_run(a, dist, world, rank, json_fd)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    voting.load_library()

    sets = make_inputs(rank, a.buffers, a.radius, not a.clean, dev)
    # S streams.  The path's one exchange -- the all-gather of every step's [32, 9, 2] key-points -- is BUCKETED: the
    # key-points of G consecutive steps (default G = S) are voted straight into the slots of a staging block and sent by
    # ONE RCCL all-gather of G x 2.3 KB per rank (fewer, larger collectives: a 2.3 KB gather is pure launch latency, and
    # one per step costs the host more than the voting's six launches).  Four staging / target blocks rotate; a block is
    # rewritten only after the gather that read it has completed (event on the communication stream, queried on the host).
    nstreams = max(1, a.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    ws_bytes = voting.vote_layout(BATCH, H, W, VN, HN, 30000).total_bytes
    spaces = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(nstreams)]  # one workspace per stream:
    # calls on one stream are ordered, so they share it; nothing is allocated inside the timed regions
    G = max(1, a.gather_bucket if a.gather_bucket > 0 else nstreams)
    NBLK = 4  # staging / target blocks in rotation
    GATHER_ON_VOTE = os.environ.get("BENCH_GATHER_ON_VOTING_STREAM") == "1"
    # Round 6 (--gather rccl, the default): the collective is the LIBRARY's ncclAllGather (pvnet_vote_allgather) and every voting stream
    # gathers its OWN votes -- a bucket is GS consecutive steps of ONE stream, sent from that stream: stream order is the only
    # dependency (no event, no communication stream, no wait across streams), and no ProcessGroup stream takes a hardware queue.
    rg = None
    gather_fallback = None
    if dist is not None and a.gather == "rccl":
        from pvnet_amd import distributed as D
        try:   # the communicator and one trial collective; every rank must end up on the same path, so the ranks agree on the outcome
            rg = D.RcclGather(dev)
            t_in = torch.full((4,), float(rank), dtype=torch.float32, device=dev)
            t_out = torch.empty((world * 4,), dtype=torch.float32, device=dev)
            rg.all_gather(t_out, t_in)
            torch.cuda.synchronize(dev)
            if t_out.view(world, 4)[:, 0].tolist() != [float(r) for r in range(world)]:
                raise RuntimeError("trial all-gather returned wrong ranks")
        except Exception as e:  # noqa: BLE001 -- any failure of the library's binding: torch.distributed's collective instead
            gather_fallback = f"{type(e).__name__}: {e}"
            rg = None
        ok = torch.tensor([1 if rg is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if rg is not None:
                gather_fallback = "another rank's RCCL binding failed"
            rg = None
    GS = max(1, a.gather_bucket) if a.gather_bucket > 0 else 3   # steps per collective of a stream (rccl mode)
    if rg is not None:
        staging_s = [[torch.empty((GS, BATCH, VN, 2), dtype=torch.float32, device=dev) for _ in range(NBLK)] for _ in range(nstreams)]
        gathered_s = [[torch.empty((world, GS, BATCH, VN, 2), dtype=torch.float32, device=dev) for _ in range(NBLK)] for _ in range(nstreams)]
        sfill = [[0, 0] for _ in range(nstreams)]   # per stream: bucket index, slots filled
        staging, gathered = staging_s[0], gathered_s[0]   # (the exchange timed alone below uses one block)
    if dist is not None and rg is None:
        comm = None if GATHER_ON_VOTE else torch.cuda.Stream(dev)
        staging = [torch.empty((G, BATCH, VN, 2), dtype=torch.float32, device=dev) for _ in range(NBLK)]
        gathered = [torch.empty((world, G, BATCH, VN, 2), dtype=torch.float32, device=dev) for _ in range(NBLK)]
        sent = [None] * NBLK  # event: the gather that last read staging[blk] is done
    bucket = {"n": 0, "fill": 0, "used": set()}  # index of the bucket being filled, slots filled, streams that voted into it
    pending = []

    def flush_stream(si):
        """rccl mode: stream si sends its current bucket (a last, partly filled one whole: same size as every other) from ITSELF"""
        n, k = sfill[si]
        if k == 0:
            return
        with torch.cuda.stream(streams[si]):
            rg.all_gather(gathered_s[si][n % NBLK], staging_s[si][n % NBLK])
        sfill[si] = [n + 1, 0]

    def flush():
        """send the filled slots of the current bucket (all G of them, except for a last partial bucket)"""
        if rg is not None:
            for si in range(nstreams):
                flush_stream(si)
            return
        blk, k = bucket["n"] % NBLK, bucket["fill"]
        if k == 0:
            return
        # one event per STREAM that voted into the bucket, recorded now (it covers every vote the stream was given), instead
        # of one per step: the communication stream is the only one that waits across streams
        used = sorted(bucket["used"])
        # BENCH_GATHER_ON_VOTING_STREAM=1 (experiment): no separate communication stream -- the collective is
        # issued from the voting stream that filled the bucket's last slot, after it has waited for the other streams' votes
        tgt = streams[used[-1]] if GATHER_ON_VOTE else comm
        evs = [streams[si].record_event() for si in used if streams[si] is not tgt]
        with torch.cuda.stream(tgt):
            for ev in evs:
                tgt.wait_event(ev)
            # a last, partly filled bucket is sent whole (its unused slots carry the previous contents): no allocation and
            # no copy inside the timed region, one collective of the same size as every other
            w = dist.all_gather_into_tensor(gathered[blk], staging[blk], async_op=True)
            w.wait()  # comm stream waits for the collective
            sent[blk] = torch.cuda.Event()
            sent[blk].record(tgt)
        pending.append(sent[blk])
        del pending[:-NBLK]  # (older gathers are ordered before these on the communication stream)
        bucket["n"] += 1
        bucket["fill"] = 0
        bucket["used"] = set()

    def step(i, ns=nstreams, mode=None, **kw):
        m, v, _, _ = sets[i % len(sets)]
        mode = mode or {}
        if kw:  # profiled / debug calls: current stream, synchronising
            return voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + i,
                                                 image_offset=rank * BATCH, **mode, **kw)
        st = streams[i % ns]
        conc = ns > 1   # explicit: batches on other streams are in flight exactly when the region uses several
        with torch.cuda.stream(st):
            if dist is None:
                return voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + i,
                                                     image_offset=rank * BATCH, workspace=spaces[i % ns], concurrent=conc,
                                                     **mode)
            if rg is not None:   # a slot of this stream's own staging block; the gather that last read the block is earlier on this stream
                si = i % ns
                n, j = sfill[si]
                out = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + i, image_offset=rank * BATCH,
                                                    out=staging_s[si][n % NBLK][j], workspace=spaces[si], concurrent=conc, **mode)
                sfill[si][1] = j + 1
                if j + 1 == GS:
                    flush_stream(si)
                return out
            blk, j = bucket["n"] % NBLK, bucket["fill"]
            # the gather that last read this block (NBLK buckets ago) must be complete before a vote overwrites a slot: with
            # four blocks in rotation it long is -- a host-side query, and a device-side wait only if it is not
            if sent[blk] is not None and not sent[blk].query():
                st.wait_event(sent[blk])
            out = voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=SEED0 + i,
                                                image_offset=rank * BATCH, out=staging[blk][j], workspace=spaces[i % ns],
                                                concurrent=conc, **mode)
        bucket["used"].add(i % ns)
        bucket["fill"] += 1
        if bucket["fill"] == G:
            flush()
        return out

    token = torch.zeros(1, dtype=torch.int32, device=dev) if dist is not None else None

    def barrier():
        """every rank has arrived: a one-element all-reduce on a preallocated token + a device synchronise -- what
        `dist.barrier()` does on the RCCL backend, without its per-call allocation and device bookkeeping (0.2-0.3 ms,
        inside the timed region of a 20-step run)"""
        dist.all_reduce(token)
        torch.cuda.synchronize(dev)

    def fence():
        """the bracket of a timed region: this rank's work (incl. its last gather) is complete -> clock reading -> every rank
        has arrived (barrier) -> device idle.  Returns the clock reading: a closing bracket stops the rank's clock when ITS K
        steps are done -- the whole job's time is the MAX of those over the ranks (taken by the caller) -- and not after the
        barrier's own all-reduce (0.1-0.3 ms, 3-8 % of a 20-step region), which is not part of the K steps."""
        if dist is not None:
            flush()
        while pending:
            pending.pop(0).synchronize()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        if dist is not None:
            barrier()
        torch.cuda.synchronize(dev)
        return t

    def timed(ns, steps=None, mode=None, **kw):
        """ONE timed region: warm-up, fence, exactly `steps` steps, fence; max over ranks"""
        steps = a.steps if steps is None else steps
        for i in range(a.warmup if not kw else min(a.warmup, 3)):
            step(i, ns, mode, **kw)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, ns, mode, **kw)
        dt_local = fence() - t0
        dt = dt_local
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, dt_local

    def regions(ns, mode=None, n=None):
        """the timed region `n` times; returns the per-region (max-over-ranks, local) durations"""
        return [timed(ns, mode=mode) for _ in range(max(1, a.regions if n is None else n))]

    def median(xs):
        xs = sorted(xs)
        k = len(xs) // 2
        return xs[k] if len(xs) % 2 else 0.5 * (xs[k - 1] + xs[k])

    MAIN = {"approx": True} if a.approx else {}
    t_pre = time.perf_counter()                      # device pre-warm (untimed, reported in config.prewarm_s)
    i_pre = 0
    while time.perf_counter() - t_pre < a.prewarm_seconds:
        for _ in range(200):  # voting only: a time-bounded loop must not issue collectives (ranks would disagree on the count)
            m, v, _, _ = sets[i_pre % len(sets)]
            with torch.cuda.stream(streams[i_pre % nstreams]):
                voting.ransac_voting_layer_v3(m, v, HN, inlier_thresh=THRESH, seed=i_pre, image_offset=rank * BATCH)
            i_pre += 1
        torch.cuda.synchronize(dev)
    fence()
    # with the driver's K = 20 a region lasts 3 ms and the first one after the pre-warm loop (which idles at
    # every synchronise) still saw the clock ramp (188.9 k against 198 k for the other fourteen).  One more region of the very
    # form that is timed -- warm-up, fence, K steps, fence -- runs first and is NOT recorded: the recorded ones start from the
    # state a steady caller is in.  (Not a change of what a region is: W untimed steps, exactly K timed steps, fences around.)
    # (round 4, r04c36-38: with K = 20 the rate climbs over the first six regions -- 205 k -> 219 k -- whatever the length of the
    # pre-warm loop and with or without the sensor thread: twelve unrecorded regions)
    # (the sensor thread is created and started BEFORE the unrecorded regions -- r04c39: whatever its start disturbs, the
    # regions right after it were 5 % low although the unrecorded ones before it had settled -- and its samples are cleared
    # when the recorded regions begin)
    with GpuSampler(local) as sampler:
        pre = regions(nstreams, mode=MAIN, n=int(os.environ.get("BENCH_UNRECORDED", "12")) if a.steps < 200 else 1)
        pre_rates = [world * BATCH * a.steps / r[0] for r in pre]   # (in the line, not only on stderr)
        sampler.samples.clear()
        runs = regions(nstreams, mode=MAIN)          # the headline: R regions of K steps, independent batches on S streams
    dts = [r[0] for r in runs]
    dt = median(dts)
    dt_local = median([r[1] for r in runs])
    runs1 = regions(1, mode=MAIN, n=max(3, a.regions // 3)) if nstreams > 1 else runs  # the same K steps one after the other
    dt1 = median([r[0] for r in runs1])
    runs_apx = regions(nstreams, mode={"approx": True}, n=max(3, a.regions // 3))  # PVNET_F_APPROX on the same inputs
    dt_apx = median([r[0] for r in runs_apx])

    per_rank = [BATCH * a.steps / dt_local]
    gather_ms = None
    if dist is not None:
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        allt = torch.empty((world,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, t)
        per_rank = [BATCH * a.steps / float(x) for x in allt.tolist()]
        fence()
        t0 = time.perf_counter()
        for i in range(50):  # the exchange alone, one bucket after the other: its latency (G x 2.3 KB per rank)
            dist.all_gather_into_tensor(gathered[0], staging[0])
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - t0) / 50 * 1e3
        dist.barrier()

    # ---- per-stage times (event pair per stage, synchronising calls) and the scoring kernel back to back ----------
    stage_sum = {}
    tn_sum = 0
    nprof = min(max(a.steps, 20), 100)
    for i in range(nprof):
        _, dbg, times = step(i, return_debug=True, stage_times=True)
        if i < len(sets):
            tn_sum += int(dbg["tn"].sum().item())
        for k, v in times.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    stage_ms = {k: v / nprof for k, v in stage_sum.items()}
    tn_per_batch = tn_sum / min(nprof, len(sets))
    pairs = HN * VN * tn_per_batch  # pair tests per launch of the scoring kernel
    reps = max(1, a.score_repeats // len(sets))
    both = [voting.stage_repeat_ms(sets[s][0], sets[s][1], HN, inlier_thresh=THRESH, stage="score", repeats=reps,
                                   seed=SEED0 + s, image_offset=rank * BATCH, both=True) for s in range(len(sets))]
    score_ms = float(np.mean([x[0] for x in both]))      # the kernel's own duration (device clock stamps)
    score_b2b_ms = float(np.mean([x[1] for x in both]))  # event pair around back-to-back launches (+ launch boundary)
    score_s = score_ms * 1e-3
    path_s = sum(stage_ms.values()) * 1e-3
    lit_dt, _ = timed(1, steps=5, mode={"literal": True})  # literal mode: a few calls on one stream

    seen = None
    if dist is not None:  # which ranks / devices took part (RCCL really spans them): every rank reports its device
        mine = torch.tensor([rank, local, torch.cuda.current_device()], dtype=torch.int64, device=dev)
        allm = torch.empty((world, 3), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allm, mine)
        seen = allm.tolist()

    if rank == 0:
        votings_per_s = world * BATCH * a.steps / dt
        step_s = dt / a.steps
        rates = [world * BATCH * a.steps / x for x in dts]
        exec_tflops = MFMA_FLOP_PER_PAIR * pairs / score_s / 1e12
        alg_tflops = FLOP_PER_PAIR * pairs / score_s / 1e12
        compulsory = BATCH * H * W * 8 + tn_per_batch * VN * 2 * 4  # the masks + the foreground vectors, per batch
        traffic, traffic_tag, traffic_names = {}, None, {}
        for k in PATH_KERNELS:
            traffic[k], traffic_tag, traffic_names[k] = measured_traffic(k)
        busy, busy_tag = measured_mfma_busy()
        measured = sum(v for v in traffic.values() if v) if all(traffic.values()) else None
        L = voting.vote_layout(BATCH, H, W, VN, HN, 30000)
        res = {
            "metric": "RANSAC votings/s (480x640, 9 kpts, batch 32) + HBM GB/s vs roofline",
            "value": votings_per_s, "unit": "votings/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "regions": {"runs": len(dts), "reported": "median", "min": min(rates), "max": max(rates),
                        "spread": (max(rates) - min(rates)) / votings_per_s, "values": rates,
                        "note": "the timed region (fence, exactly K steps, fence; max over ranks) run `runs` times; value "
                                "and ms_per_step are the MEDIAN region; before them the same region runs unrecorded (when "
                                "K < 200: twelve times) so that none of the recorded ones "
                                "sees the ramp after the device's idle phases",
                        "unrecorded_values": pre_rates,
                        "unrecorded_note": "rates of the regions run BEFORE the recorded ones (same form, same clock): rounds 1-3 "
                                           "recorded from the first region on, so their BENCH numbers include the ramp these show",
                        "gpu": sampler.summary()},
            "mode": "APPROX (--approx, development A/B only)" if a.approx else
                    "exact (library default): inlier counts and winners equal the reference kernels'",
            "single_stream": {"value": world * BATCH * a.steps / dt1, "ms_per_step": dt1 / a.steps * 1e3,
                              "runs": len(runs1),
                              "note": "the same K steps issued on one stream: per-batch latency of the whole path"},
            "approx_mode": {"value": world * BATCH * a.steps / dt_apx, "unit": "votings/s",
                            "ms_per_step": dt_apx / a.steps * 1e3, "runs": len(runs_apx),
                            "note": "PVNET_F_APPROX (the round-1/2 'fast' mode, counts within a few votes of the "
                                    "reference's) on the same inputs and streams: what the default mode's exactness costs"},
            "literal_mode": {"value": world * BATCH * 5 / lit_dt, "unit": "votings/s", "ms_per_step": lit_dt / 5 * 1e3,
                             "note": "PVNET_F_LITERAL: the reference's float32 operation order for every pair on the "
                                     "VALU (5 calls on one stream)"},
            "per_rank_votings_per_s": per_rank, "gather_ms": gather_ms,
            "gather_bucket_steps": G if dist is not None else None,
            "gather_stream": (("the voting stream that filled the bucket (the library's ncclAllGather, pvnet_vote_allgather)" if rg is not None
                               else "a voting stream (BENCH_GATHER_ON_VOTING_STREAM=1)" if GATHER_ON_VOTE else "its own stream")
                              if dist is not None else None),
            "gather_fallback": gather_fallback,
            "rccl_ranks_seen": len(seen) if seen else None,
            "rank_devices": [{"rank": r, "local_rank": l, "device": d} for r, l, d in seen] if seen else None,
            "dtype": "f32 decisions (bf16x3 MFMA products, f32 accumulate; pairs inside the f32 rounding band re-evaluated "
                     "in the reference's f32 order); refinement f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: batch=32 synthetic 480x640 fields per GPU, 9 keypoints, "
                                   "1024 hypotheses, inlier_thresh 0.99, int64 mask, planar strided field",
                       "batch_per_gpu": BATCH, "global_batch": world * BATCH, "h": H, "w": W, "vn": VN, "hn": HN,
                       "mask_radius": a.radius, "mean_foreground_px": tn_per_batch / BATCH,
                       "field": "clean" if a.clean else "noisy (0.05 rad + 10% outliers), N(0,1) background",
                       "input_sets_cycled": len(sets),
                       "touched_input_bytes": int(len(sets) * compulsory),
                       "streams": nstreams, "prewarm_s": a.prewarm_seconds,
                       "concurrent_hint": "explicit: PVNET_F_CONCURRENT (concurrent=True) on the calls of regions that issue on several "
                                          "streams, not in single_stream (voting.concurrent_hint consults the stream history only "
                                          "when a caller passes None); the flag selects a kernel variant -- contiguous item runs + "
                                          "one accumulator pair -- never a result",
                       "parallelism": f"images sharded over {world} GPU(s); steps issued round-robin on {nstreams} "
                                      f"HIP stream(s) per GPU"},
            "roofline": {"kernel": f"{SCORE_KERNEL}<{L.wg_g * L.hpl // 2}, ...>" + (" (score_exact_kernel_both_*: ONE launch for dense and disc-culled work "
                                                                                        "items; on this field K3 culls nothing)" if L.cull else ""),
                         "bound": "mfma",
                         "achieved": alg_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": alg_tflops / PEAK_BF16_TFLOPS,
                         "definition": "SURVEY.md 8d: achieved = algorithmic_flop_per_pair x pair_tests_per_launch / "
                                       "avg_launch_ms; peak = dense bf16 MFMA, the pipe the kernel runs on",
                         "algorithmic_flop_per_pair": FLOP_PER_PAIR, "pair_tests_per_launch": pairs,
                         "avg_launch_ms": score_ms, "launches_timed": reps * len(sets),
                         "avg_launch_ms_back_to_back_events": score_b2b_ms,
                         "timing": "avg_launch_ms = max end - min start of the kernel's workgroups on the device's "
                                   "constant-rate clock, averaged over the launches (pvnet_vote_v3_stage_repeat): the "
                                   "duration a kernel trace reports; the event figure adds the dependent-launch boundary",
                         "variant": "a batch alone: strided work items, one accumulator pair in 128 VGPRs (four waves per SIMD), 12 "
                                    "workgroups per CU; the multi-stream regions of `value` run the variant selected by "
                                    "PVNET_F_CONCURRENT: contiguous item runs (B columns, hypotheses and counters kept per run) "
                                    "in 136 VGPRs (three waves per SIMD), slower alone, +2.5 % with batches in flight "
                                    "(profiles/r04_ab_runs.txt)",
                         "pair_tests_per_s": pairs / score_s,
                         "vs_fp32_vector_peak": alg_tflops / PEAK_F32_TFLOPS,
                         "executed_flop_per_pair": MFMA_FLOP_PER_PAIR, "executed_tflops": exec_tflops,
                         "executed_frac": exec_tflops / PEAK_BF16_TFLOPS,
                         "traffic": traffic[SCORE_KERNEL], "traffic_from_committed_profile": traffic_tag,
                         "mfma_busy_frac": busy, "mfma_busy_frac_from_committed_profile": busy_tag,
                         "note": "frac prices the 12 algorithmic fp32 flop of a pair test (SURVEY 8d); the kernel EXECUTES "
                                 "64 matrix flop per test (bf16x3 split: two v_mfma_f32_32x32x16_bf16 per 32x32 tests, K = "
                                 "15 of 16 slots) = executed_frac, and 2.5 VALU operations per test (1 of them in the fast issue class since round 4) for the vote and the "
                                 "rounding band, which is what binds it (DESIGN.md section 5)"},
            "roofline_hbm": {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "compulsory_bytes": compulsory, "compulsory_gbs": compulsory / step_s / 1e9,
                             "compulsory_frac": compulsory / step_s / 1e9 / PEAK_HBM_GBS,
                             "measured_bytes": measured, "measured_bytes_from_committed_profile": traffic_tag,
                             "measured_gbs": measured / step_s / 1e9 if measured else None,
                             "measured_frac": measured / step_s / 1e9 / PEAK_HBM_GBS if measured else None,
                             "measured_over_compulsory": measured / compulsory if measured else None,
                             "traffic_per_kernel": traffic, "traffic_kernel_instantiations": traffic_names,
                             "dense_equivalent_bytes": BYTES_PER_VOTING * BATCH,
                             "dense_equivalent_gbs": BYTES_PER_VOTING * BATCH / step_s / 1e9,
                             "dense_equivalent_frac": BYTES_PER_VOTING * BATCH / step_s / 1e9 / PEAK_HBM_GBS,
                             "path_ms_serial": path_s * 1e3,
                             "note": "compulsory = int64 masks + foreground vectors (what must cross HBM); measured = "
                                     "rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of the five kernels (committed PMC passes, null "
                                     "when they were taken from other kernel sources than this build); dense_equivalent = "
                                     "SURVEY 8d's 24 576 072 B per voting, the bytes a dense implementation streams -- NOT "
                                     "achieved bandwidth: the path never reads the background of the field.  All three "
                                     "over the multi-stream step time."},
            "stage_ms": stage_ms,
            "kernel_source_hash": source_hash(),
        }
        if not a.no_parity:
            try:
                res["parity"] = parity_check(sets, rank)
            except Exception as e:  # a failing checker is reported, it must not hide the measurement
                res["parity"] = {"pass": False, "error": f"{type(e).__name__}: {e}"}
        if not a.no_secondary:
            try:
                res["secondary"] = secondary_block(sets, rank, dev)
            except Exception as e:  # reported, never allowed to take the measured line down
                res["secondary"] = {"pass": False, "error": f"{type(e).__name__}: {e}"}
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


def stub_run(a, dist, world, rank, json_fd):
    """TEST ONLY: the launcher, the process group, the gather and the max-over-ranks timing on gloo / CPU with a stub
    voter (a fixed-cost sleep returning rank-tagged key-points).  Not a measurement; the line says so."""
    def voter(i):
        time.sleep(2e-4)
        return torch.full((BATCH, VN, 2), float(rank), dtype=torch.float32)

    gathered = torch.empty((world * BATCH, VN, 2), dtype=torch.float32)

    def run(n):
        for i in range(n):
            out = voter(i)
            if dist is not None:
                dist.all_gather_into_tensor(gathered, out)
        if dist is not None:
            dist.barrier()

    run(a.warmup)
    t0 = time.perf_counter()
    run(a.steps)
    dt_local = time.perf_counter() - t0
    dt, per_rank = dt_local, [BATCH * a.steps / dt_local]
    if dist is not None:
        t = torch.tensor([dt_local], dtype=torch.float64)
        allt = torch.empty((world,), dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        dt = float(allt.max())
        per_rank = [BATCH * a.steps / float(x) for x in allt.tolist()]
        ok = all(bool((gathered[r * BATCH:(r + 1) * BATCH] == float(r)).all()) for r in range(world))
    else:
        ok = True
    if rank == 0:
        res = {"metric": "RANSAC votings/s (480x640, 9 kpts, batch 32) + HBM GB/s vs roofline",
               "value": world * BATCH * a.steps / dt, "unit": "votings/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "none", "data": "STUB voter on CPU/gloo -- plumbing test, not a measurement",
               "stub": True, "gather_ok": ok, "per_rank_votings_per_s": per_rank,
               "config": {"workload": "stub", "global_batch": world * BATCH}}
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
