"""Build-time check of every kernel's register accounting (pvnet_vote.hip, pvnet_nn.hip).

Rule: a kernel must not use the last VGPR granule of its allocation -- `.amdhsa_next_free_vgpr` has to exceed the highest
VGPR an instruction names by at least 8 (PVNET_SPARE_VGPRS in the sources provides the slack).  Background: round 2's
compaction flake (profiles/r02_compaction_flake_investigation.txt): identical code failed in 98 % of the runs with an
allocation that was used to the top and never with one granule more.

    python tools/check_kernel_resources.py            # compiles both sources to assembly (hipcc -S) and checks them
    python tools/check_kernel_resources.py a.s b.s

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


def compile_to_asm(src, out):
    sys.path.insert(0, ROOT)
    from pvnet_amd import build as B
    flags = [f for f in B.flags() if f not in ("-shared", "-fPIC")]
    subprocess.check_call([B.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", src, "-o", out],
                          stderr=subprocess.DEVNULL)


def kernels(text):
    """[(name, next_free_vgpr, highest VGPR named by an instruction)]"""
    out = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)", text):
        name = m.group(1)
        desc = text[m.start():text.index(".end_amdhsa_kernel", m.start())]
        nfv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        i = text.index(name + ":")
        body = text[i:text.index(".Lfunc_end", i)]
        body = "\n".join(l.split(";")[0] for l in body.splitlines())  # comments may mention registers
        used = [int(x) for x in re.findall(r"\bv(\d+)\b", body)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", body)]
        out.append((name, nfv, max(used) if used else -1))
    return out


def short(name):
    return re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:56]


def main(argv):
    texts = []
    if argv:
        texts = [open(a).read() for a in argv]
    else:
        sys.path.insert(0, ROOT)
        from pvnet_amd import build as B
        with tempfile.TemporaryDirectory() as d:
            for k, src in enumerate(B.SRC):
                out = os.path.join(d, f"k{k}.s")
                compile_to_asm(src, out)
                texts.append(open(out).read())
    bad = 0
    n = 0
    for t in texts:
        for name, nfv, vmax in kernels(t):
            n += 1
            slack = nfv - (vmax + 1)
            alloc = (nfv + 7) // 8 * 8
            flag = "" if slack >= SLACK else "   <-- uses its last granule: raise PVNET_SPARE_VGPRS"
            bad += slack < SLACK
            print(f"{short(name):58s} uses v0..v{vmax:<3d} allocates {alloc:3d} (slack {slack:3d}, {min(8, 512 // alloc)} waves/SIMD){flag}")
    print(f"checked {n} kernels, {bad} without a spare granule")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
