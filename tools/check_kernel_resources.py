"""Build-time check of every kernel's register accounting (the translation units of libpvnet_vote.so, release and development build).

Rule: a kernel must not use the last VGPR granule of its allocation -- `.amdhsa_next_free_vgpr` has to exceed the highest
VGPR an instruction names by at least 8 (PVNET_SPARE_VGPRS in the sources provides the slack).  Background: round 2's
compaction flake (profiles/r02_compaction_flake_investigation.txt): identical code failed in 98 % of the runs with an
allocation that was used to the top and never with one granule more.

    python tools/check_kernel_resources.py            # compiles both sources to assembly (hipcc -S) and checks them
    python tools/check_kernel_resources.py a.s b.s

Second rule: the kernels whose budget is set with `amdgpu_num_vgpr` (hypothesis_kernel, score_exact_kernel_*) rely on
the gfx950 backend DOUBLING the attribute's literal (unified VGPR / AGPR file).  A compiler that stopped doing so would cap them at
half their registers: massive spilling.  So for those kernels the allocation must be EXACTLY the one the occupancy plan assumes
(EXPECTED_ALLOC: not more -- a wave per SIMD lost -- and not less), they must use more than half of it (a halved cap cannot) and
none of them may spill more than its own few dwords (`.amdhsa_private_segment_fixed_size`, limit per pattern in EXPECTED_ALLOC).

Also reports the waves per SIMD each allocation permits (512 VGPRs per SIMD, granule 8).  Exit status 1 if a kernel lacks
the slack.  tests/test_library_cpu.py runs it on every build of the CPU suite.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLACK = 8
# kernels with an `amdgpu_num_vgpr` budget: name pattern -> the allocation (VGPRs, granule-rounded) their occupancy plan assumes, and
# the bytes per lane they may spill (per pattern -- the default dense kernels none to speak of: a spill in their loop is a
# regression; the two-pair variant five dwords; the hypothesis kernel ten, parked across its preamble (after the loads they hold have arrived); the merged dense + disc-culling kernel the per-item constants the compiler hoists
# out of the culling body's item loop).  A halved cap spills hundreds.
EXPECTED_ALLOC = [(r"\d+hypothesis_kernelI", 48, 48), (r"score_exact_kernel_[12]_", 112, 12),
                  (r"score_exact_kernel_4_", 144, 12), (r"score_exact_kernel_8_\d_\d_1_0", 128, 12), (r"score_exact_kernel_8_\d_\d_1_1", 136, 12),
                  (r"score_exact_kernel_8_\d_\d_2_", 168, 24), (r"score_exact_kernel_both_\d_0", 128, 256), (r"score_exact_kernel_both_\d_1", 136, 256)]


def compile_to_asm(src, out, dev=False):
    sys.path.insert(0, ROOT)
    from pvnet_amd import build as B
    flags = [f for f in B.flags() if f not in ("-shared", "-fPIC")] + (["-DPVNET_DEV"] if dev else [])
    subprocess.check_call([B.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", src, "-o", out],
                          stderr=subprocess.DEVNULL)


def kernels(text):
    """[(name, next_free_vgpr, highest VGPR named by an instruction)]"""
    out = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)", text):
        name = m.group(1)
        desc = text[m.start():text.index(".end_amdhsa_kernel", m.start())]
        nfv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        m2 = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc)
        scratch = int(m2.group(1)) if m2 else 0
        i = text.index(name + ":")
        body = text[i:text.index(".Lfunc_end", i)]
        body = "\n".join(l.split(";")[0] for l in body.splitlines())  # comments may mention registers
        used = [int(x) for x in re.findall(r"\bv(\d+)\b", body)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", body)]
        out.append((name, nfv, max(used) if used else -1, scratch))
    return out


def short(name):
    return re.sub(r"^_ZN(3pvd)?\d+_GLOBAL__N_1\d+", "", name)[:56]


def main(argv):
    texts = []
    if argv:
        texts = [open(a).read() for a in argv]
    else:
        sys.path.insert(0, ROOT)
        from pvnet_amd import build as B
        with tempfile.TemporaryDirectory() as d:
            from concurrent.futures import ThreadPoolExecutor
            jobs = [(src, dev, os.path.join(d, f"k{k}{'d' if dev else ''}.s")) for k, src in enumerate(B.SRC) for dev in (False, True)
                    if "rccl" not in src]   # release and development (-DPVNET_DEV) instantiations of every translation unit
            with ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 4) - 1)) as ex:
                list(ex.map(lambda j: compile_to_asm(j[0], j[2], j[1]), jobs))
            for _, _, out in jobs:
                t = open(out).read()
                texts.append(t)
    bad = 0
    n = 0
    done = set()
    for t in texts:
        for name, nfv, vmax, scratch in kernels(t):
            if (name, nfv, vmax, scratch) in done:   # the same kernel in the release and the development build
                continue
            done.add((name, nfv, vmax, scratch))
            n += 1
            slack = nfv - (vmax + 1)
            alloc = (nfv + 7) // 8 * 8
            flag = "" if slack >= SLACK else "   <-- uses its last granule: raise PVNET_SPARE_VGPRS"
            bad += slack < SLACK
            for pat, want, max_scratch in EXPECTED_ALLOC:
                if re.search(pat, name):
                    if alloc != want or vmax + 1 <= want // 2 or scratch > max_scratch:
                        flag += f"   <-- budgeted kernel: expected {want} VGPRs allocated, more than {want // 2} used, no scratch (got {alloc}, {vmax + 1}, {scratch} B)"
                        bad += 1
                    break
            print(f"{short(name):58s} uses v0..v{vmax:<3d} allocates {alloc:3d} (slack {slack:3d}, {min(8, 512 // alloc)} waves/SIMD, scratch {scratch:3d} B){flag}")
    print(f"checked {n} kernels, {bad} without a spare granule / off their register budget")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
