"""Development aid: binary-level variants of the stand-alone compaction reproducer (k2_repro.hip).

    hipcc --offload-arch=gfx950 -O3 -w -S --cuda-device-only k2_repro.hip -o base.s      # the device assembly
    python asm_variants.py base.s out_dir                                                  # writes out_dir/*.co
    hipcc -O2 -w co_runner.cpp -o co_runner.bin ; for f in out_dir/*.co; do ./co_runner.bin $f 300; done   # on the MI355X

Every variant keeps the instruction stream of base.s and changes ONE thing: the VGPR allocation in the kernel descriptor,
the place of the four highest registers, a wait before s_endpgm, NOPs after an instruction class.  Results of round 2
(bad runs of 300): base 294 | .amdhsa_next_free_vgpr 24 -> 32 / 40 / 64 / 128: 0 | v20..v23 moved to v28..v31 with 32
allocated: 295, with 40 allocated: 0 | v20..v23 swapped with v8..v11 at 24 allocated: 0 | s_waitcnt vmcnt(0) lgkmcnt(0) before
s_endpgm: 298 | s_nop 3 after every instruction of the loop: 52 | after every v_cndmask: 2 | after every ds_*: 194.
"""
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"
KERNEL = "_Z14compact_kernelILi1EEv6Params"


def build(out_dir, name, text):
    s, o, co = (f"{out_dir}/{name}.{e}" for e in ("s", "o", "co"))
    open(s, "w").write(text)
    subprocess.check_call([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "ld.lld", "-shared", o, "-o", co])


def allocation(text, n):
    t = re.sub(r"\.amdhsa_next_free_vgpr \d+", f".amdhsa_next_free_vgpr {n}", text)
    t = re.sub(r"\.amdhsa_accum_offset \d+", f".amdhsa_accum_offset {n}", t)
    return re.sub(r"\.vgpr_count:\s+\d+", f".vgpr_count:     {n}", t)


def permute(text, perm):
    """rename VGPRs inside the kernel body; register ranges must stay contiguous under the permutation"""
    m1 = lambda n: perm.get(n, n)

    def pair(m):
        a, b = int(m.group(1)), int(m.group(2))
        assert all(m1(a + i) == m1(a) + i for i in range(b - a + 1)), (a, b)
        return f"v[{m1(a)}:{m1(a) + b - a}]"
    s, e = text.index(KERNEL + ":"), text.index(".Lfunc_end0")
    body = re.sub(r"\bv\[(\d+):(\d+)\]", pair, text[s:e])
    body = re.sub(r"\bv(\d+)\b", lambda m: f"v{m1(int(m.group(1)))}", body)
    return text[:s] + body + text[e:]


def nops_after(text, prefix, nop="s_nop 3"):
    out = []
    for line in text.splitlines():
        out.append(line)
        if line.strip().startswith(prefix):
            out.append("\t" + nop)
    return "\n".join(out) + "\n"


def main(base, out_dir):
    src = open(base).read()
    top = max(int(x) for x in re.findall(r"\.amdhsa_next_free_vgpr (\d+)", src))
    build(out_dir, "base", src)
    for n in (top + 8, top + 16, 64, 128):
        build(out_dir, f"alloc{n}", allocation(src, n))
    up = {top - 4 + i: top + 4 + i for i in range(4)}
    build(out_dir, "top_quad_up_alloc_exact", allocation(permute(src, up), top + 8))
    build(out_dir, "top_quad_up_alloc_spare", allocation(permute(src, up), top + 16))
    swap = {**{top - 4 + i: 8 + i for i in range(4)}, **{8 + i: top - 4 + i for i in range(4)}}
    try:
        build(out_dir, "top_quad_swapped_with_v8", permute(src, swap))
    except AssertionError as e:  # a register range of this build straddles the swapped quads
        print("swap variant skipped:", e)
    build(out_dir, "wait_before_endpgm", src.replace("\ts_endpgm", "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_endpgm"))
    for cls in ("v_cndmask", "v_cmp", "ds_", "v_lshrrev_b64", "global_"):
        build(out_dir, "nop_after_" + cls.strip("_"), nops_after(src, cls))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
