"""GPU canary for round 2's compaction flake (ADVICE r02 "track it as an open correctness bug").

`tools/experiments/k2_flake/k2_repro.hip` is the stand-alone copy of `compact_kernel<false, 1>` as it FAILED: its code uses
the wave's VGPR allocation up to the last granule, and with two of its workgroups on a CU up to 64 records of a wave came
from pixels a few ranks away in 97-99 % of the runs.  The cause was never found; the library's rule since -- every kernel
allocates one VGPR granule more than it uses (PVNET_SPARE_VGPRS), enforced at build time by
tools/check_kernel_resources.py -- is empirical.  This file makes the driver's GPU boxes report on it every round:

  * test_spare_granule_build_is_clean  MUST pass: the same kernel with the spare granule, 400 runs, zero wrong records --
    if this ever fails the rule no longer protects the library and every result is suspect;
  * test_tight_allocation_still_flakes  is the canary proper: the kernel as it failed.  It XFAILS (reported, not an error)
    while the hardware / compiler still show the flake, and passes if a driver / compiler / box no longer does."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "experiments", "k2_flake", "k2_repro_{}.bin")


def run(kind, reps):
    exe = BIN.format(kind)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (python -m pvnet_amd.build)")
    p = subprocess.run([exe, str(reps)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"bad runs: (\d+) of (\d+)", p.stdout)
    assert m, p.stdout[-2000:]
    return int(m.group(1)), int(m.group(2)), p.stdout


def test_spare_granule_build_is_clean():
    bad, n, out = run("spare", 400)
    assert n == 400 and bad == 0, out[-1500:]


def test_tight_allocation_still_flakes():
    bad, n, out = run("tight", 200)
    print(f"compaction-flake canary: {bad} of {n} runs of the tight-allocation kernel returned wrong records")
    if bad:
        pytest.xfail(f"the flake is still there on this box: {bad} of {n} runs wrong with the VGPR allocation used to its "
                     f"last granule (0 expected of the spare-granule build, checked separately)")
