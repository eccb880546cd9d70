"""Round 6: bisect round 4's strongest lead -- `s_nop 3` after EVERY v_cndmask of the reproducer takes the failure rate from
292 / 300 to 2 / 300 (profiles/r04_flake_report.md).  Which v_cndmask carries the effect?

    hipcc --offload-arch=gfx950 -O3 -w -S --cuda-device-only k2_repro.hip -o base.s
    python bisect_cndmask.py base.s r6_co          # HERE (the assembler is in the image): writes r6_co/*.co + r6_co/sites.txt
    bash run_r6.sh                                  # on the MI355X: co_runner.bin over every variant, 300 launches each

Variants (one change each, the instruction stream of base.s otherwise):
  all            s_nop 3 after every v_cndmask (round 4's 2 / 300)
  blk_<label>    after the v_cndmask of ONE basic block (the label that precedes them)
  one_<i>        after the i-th v_cndmask only (source order; sites.txt lists line, block and the instruction before it)
  pre_<i>        BEFORE the i-th v_cndmask only (between the instruction that writes its mask / operand and the select)
  half_lo / half_hi   after the first / second half of the sites
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_variants import build  # noqa: E402


def sites(lines):
    out, blk = [], "entry"
    for i, ln in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", ln)
        if m:
            blk = m.group(1)
        if ln.strip().startswith("v_cndmask"):
            prev = next((lines[j].strip() for j in range(i - 1, -1, -1)
                         if lines[j].strip() and not lines[j].strip().startswith((";", ".")) and not lines[j].rstrip().endswith(":")), "")
            out.append((i, blk, prev))
    return out


def with_nops(lines, idx_after=(), idx_before=(), nop="s_nop 3"):
    out = []
    for i, ln in enumerate(lines):
        if i in idx_before:
            out.append("\t" + nop)
        out.append(ln)
        if i in idx_after:
            out.append("\t" + nop)
    return "\n".join(out) + "\n"


def main(base, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    lines = open(base).read().splitlines()
    st = sites(lines)
    with open(os.path.join(out_dir, "sites.txt"), "w") as f:
        for n, (i, blk, prev) in enumerate(st):
            f.write(f"{n:3d} line {i + 1:4d} {blk:10s} {lines[i].strip():50s} <- {prev}\n")
    build(out_dir, "base", "\n".join(lines) + "\n")
    build(out_dir, "all", with_nops(lines, {i for i, _, _ in st}))
    half = len(st) // 2
    build(out_dir, "half_lo", with_nops(lines, {i for i, _, _ in st[:half]}))
    build(out_dir, "half_hi", with_nops(lines, {i for i, _, _ in st[half:]}))
    for blk in sorted({b for _, b, _ in st}):
        build(out_dir, "blk_" + blk.strip("."), with_nops(lines, {i for i, b, _ in st if b == blk}))
    for n, (i, _, _) in enumerate(st):
        build(out_dir, f"one_{n:02d}", with_nops(lines, {i}))
        build(out_dir, f"pre_{n:02d}", with_nops(lines, (), {i}))
    # level 2 (round 6, second call): no single site carries the effect, block LBB0_25 as a whole does (6 / 200 against 193 / 200,
    # profiles/r06a_flake_bisect.txt) -- contiguous RANGES of that block's sites (the rank -> bit search's select chains), and every
    # site EXCEPT that block's
    b25 = [n for n, (_, blk, _) in enumerate(st) if blk == ".LBB0_25"]
    if b25:
        lo, hi = b25[0], b25[-1]
        build(out_dir, "allbut_LBB0_25", with_nops(lines, {i for n, (i, _, _) in enumerate(st) if n < lo or n > hi}))
        cuts = [(lo, lo + 2), (lo + 3, hi - 3), (hi - 2, hi), (lo, lo + 7), (lo + 8, hi), (lo + 3, lo + 7), (lo + 8, hi - 3),
                (lo + 3, lo + 5), (lo + 6, lo + 8), (lo + 9, lo + 11), (lo + 12, hi - 3)]
        for a, c in cuts:
            build(out_dir, f"rng_{a:02d}_{c:02d}", with_nops(lines, {st[n][0] for n in range(a, c + 1)}))
        for par in (0, 1):   # every other site of the block
            build(out_dir, f"alt{par}_LBB0_25", with_nops(lines, {st[n][0] for n in b25 if (n - lo) % 2 == par}))
    print(f"{len(st)} v_cndmask sites, {len(os.listdir(out_dir))} files in {out_dir}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
