"""Builds libpvnet_vote.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python -m pvnet_amd.build            # build if sources are newer than the library
    python -m pvnet_amd.build --force

hipcc cross-compiles without a GPU.  The library lands next to this file so that it travels with the tree."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "pvnet_vote.hip"), os.path.join(HERE, "csrc", "pvnet_nn.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "pvnet_rng.h"), os.path.join(ROOT, "include", "pvnet_vote.h"),
              os.path.join(ROOT, "include", "pvnet_nn.h")]
LIB = os.path.join(HERE, "libpvnet_vote.so")
ARCH = "gfx950"
# host-side pose refinement (plain C++, g++): include/pvnet_pnp.h
PNP_SRC = os.path.join(HERE, "csrc", "pvnet_pnp.cpp")
PNP_DEPS = [PNP_SRC, os.path.join(ROOT, "include", "pvnet_pnp.h")]
PNP_LIB = os.path.join(HERE, "libpvnet_pnp.so")


def hipcc_path() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def flags():
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared", "-fno-fast-math",
            "-fno-slp-vectorize",  # keep the scoring loops as scalar v_sub/v_mul/v_fma (packed f32 gains nothing on gfx950)
            "-mllvm", "-amdgpu-mfma-vgpr-form",  # MFMA results straight into VGPRs: the vote's v_cmp reads them in place
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-Wall"]


def up_to_date() -> bool:
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build_pnp(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(PNP_LIB) and all(os.path.getmtime(PNP_LIB) >= os.path.getmtime(d) for d in PNP_DEPS):
        return PNP_LIB
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler found for libpvnet_pnp.so (set CXX)")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), PNP_SRC,
           "-o", PNP_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return PNP_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    build_pnp(force, verbose)
    if not force and up_to_date():
        return LIB
    cmd = [hipcc_path()] + flags() + SRC + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
