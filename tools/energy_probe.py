"""Development aid: energy per batch of the voting path and of its parts.  For each variant a loop of calls runs for a
few seconds on S streams while a sampler thread reads `rocm-smi --showpower --showclocks`; the table gives the step time,
the mean package power and clock, and their product = energy per batch.
    python tools/energy_probe.py [streams] [--approx]        (default: the library's default = exact mode)
Variants (PVNET_DEV_STAGES masks on workspaces that complete calls left behind): all six stages, the scoring kernel alone,
the five small stages alone; then all stages on ONE stream."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

APPROX = "--approx" in sys.argv
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
S = int(_pos[0]) if _pos else 6
SECONDS = float(os.environ.get("PROBE_SECONDS", 5))
dev = torch.device("cuda:0")
sets = []
for s in range(2):
    mask, planar, _ = synth.make_batch(32, first_index=s * 32, radius=40, noise=True, background="normal")
    sets.append((torch.from_numpy(mask).to(dev), synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))))
L = voting.vote_layout(32, 480, 640, 9, 1024, 30000)


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\): ([0-9.]+)", txt)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if p and c:
                out.append((float(p.group(1)), float(c.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


def variant(name, stages, ns):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    ws = [torch.empty(L.total_bytes, dtype=torch.uint8, device=dev) for _ in range(ns)]

    def run(n, i0=0):
        for i in range(i0, i0 + n):
            m, v = sets[i % 2]
            with torch.cuda.stream(streams[i % ns]):
                voting.ransac_voting_layer_v3(m, v, 1024, inlier_thresh=0.99, seed=i % 64, workspace=ws[i % ns], approx=APPROX)
        torch.cuda.synchronize()

    os.environ["PVNET_DEV_STAGES"] = str(0x3F)
    voting.reload_tuning()
    run(4 * ns)  # valid workspaces
    os.environ["PVNET_DEV_STAGES"] = str(stages)
    voting.reload_tuning()
    run(200)
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        run(500, n)
        n += 500
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    os.environ["PVNET_DEV_STAGES"] = str(0x3F)
    voting.reload_tuning()
    samples = samples[1:] or samples
    pw = sum(s[0] for s in samples) / max(1, len(samples))
    ck = sum(s[1] for s in samples) / max(1, len(samples))
    step = dt / n
    print(f"{name:44s} {ns} stream(s): {step * 1e6:7.1f} us per batch, {pw:7.0f} W, {ck:5.0f} MHz  -> {step * pw * 1e3:6.1f} mJ per batch "
          f"({len(samples)} samples)", flush=True)


idle = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
print("mode:", "approx" if APPROX else "exact")
print("idle:", " ".join(re.findall(r"Power \(W\): [0-9.]+", idle)))
if "--each" in sys.argv:  # every small stage on its own (what does each cost when S batches keep it busy?)
    for nm, bits in (("K1 mask bits alone", 0x01), ("K2 compaction alone", 0x04), ("K3 hypotheses + plan alone", 0x08),
                     ("K5 select / refine alone", 0x20), ("K1 + K2", 0x05), ("K3 + K5", 0x28), ("five small stages", 0x2F)):
        variant(nm, bits, S)
    for nm, bits in (("K1 mask bits alone", 0x01), ("K2 compaction alone", 0x04), ("K3 hypotheses + plan alone", 0x08),
                     ("K5 select / refine alone", 0x20)):
        variant(nm, bits, 1)
    sys.exit(0)
variant("all six stages", 0x3F, S)
variant("scoring kernel alone", 0x10, S)
variant("five small stages alone", 0x2F, S)
variant("all six stages", 0x3F, 1)
variant("scoring kernel alone", 0x10, 1)
variant("five small stages alone", 0x2F, 1)
