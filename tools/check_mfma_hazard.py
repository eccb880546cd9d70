"""Build-time check of the matrix-pipe scoring kernels' hand-placed vote epilogues (k4_score_mfma.hip: vote8; k4_exact_body.h: vote_subs / vote_slow_*).

vote8 reads MFMA result VGPRs from inside an inline-asm block.  LLVM inserts the gfx950 "XDL write VGPR -> VALU read"
wait states only for instructions IT schedules; what an INLINEASM block reads is invisible to its hazard recogniser,
so the distance between every v_mfma and the first instruction that reads (or overwrites) one of its result registers
is re-checked here on the generated assembly:

    python tools/check_mfma_hazard.py            # compiles the scoring translation units to assembly (hipcc -S) and checks them
    python tools/check_mfma_hazard.py file.s

Rule (LLVM GCNHazardRecognizer, gfx950 XDL ops): an N-pass MFMA needs N + 3 wait states before a VALU instruction may
read or overwrite its destination -- v_mfma_f32_32x32x16_bf16 is 8 passes -> 11.  A wait state is one issued
instruction; `s_nop k` counts k + 1.  The walk follows fall-through and branch targets (loops are followed around
their back edge).  Exit status 1 and a listing of the offending pairs if any distance is too short.
tests/test_library_cpu.py::test_vote_epilogue_keeps_mfma_hazard_distance runs this on every build of the CPU suite.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {"v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_16x16x32_bf16": 4,
          "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_16x16x4_f32": 8}


KERNELS = ("score_mfma_kernel", "score_exact_kernel")  # every kernel whose epilogue reads MFMA results from inline asm


def required(op):
    return PASSES.get(op, 16) + 3


MFMA_TU = ("k4_score_mfma.hip", "k4_score_exact.hip", "k4_score_cull.hip")   # the translation units with matrix-pipe scoring kernels


def compile_to_asm(out):
    """the translation units of MFMA_TU, release and development instantiations, to ONE assembly text"""
    sys.path.insert(0, ROOT)
    from pvnet_amd import build as B
    flags = [f for f in B.flags() if f not in ("-shared", "-fPIC")]
    text = ""
    for src in B.SRC:
        if os.path.basename(src) not in MFMA_TU:
            continue
        for dev in ([], ["-DPVNET_DEV"]):
            cmd = [B.hipcc_path()] + flags + dev + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", src, "-o", out]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            text += open(out).read() + "\n"
    open(out, "w").write(text)


def vregs(tok):
    """set of VGPR numbers named by an operand token such as v12, |v3|, -v7, v[2:17]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def parse_functions(text):
    """{name: [(kind, payload)]} with kind in {'label', 'inst'}; an inst payload is (mnemonic, [operand tokens])"""
    funcs, cur, name = {}, None, None
    for raw in text.splitlines():
        line = raw.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+|\w+):\s*$", line)
        if m and not line.startswith(".L"):
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            cur.append(("label", m.group(1)))
            continue
        s = line.strip()
        if not s or s.startswith(".") or s.startswith("//"):
            if s.startswith(".Lfunc_end"):
                cur = None
            continue
        parts = s.split(None, 1)
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        cur.append(("inst", (parts[0], ops)))
    return funcs


def check_function(name, items):
    labels = {p: i for i, (k, p) in enumerate(items) if k == "label"}
    problems = []
    n_mfma = 0
    for i, (k, p) in enumerate(items):
        if k != "inst" or not p[0].startswith("v_mfma"):
            continue
        n_mfma += 1
        op, ops = p
        dst = vregs(ops[0])
        need = required(op)
        # walk forward with a budget of `need` wait states; report any toucher of dst inside the budget
        seen = set()
        stack = [(i + 1, 0)]
        while stack:
            j, used = stack.pop()
            while j < len(items) and used < need:
                if (j, used) in seen:
                    break
                seen.add((j, used))
                kk, pp = items[j]
                if kk == "label":
                    j += 1
                    continue
                mn, oo = pp
                touched = set().union(*[vregs(t) for t in oo]) if oo else set()
                if touched & dst and not mn.startswith("s_"):
                    problems.append((name, i, op, ops[0], j, mn, " ".join(oo), used, need))
                    break
                if mn == "s_endpgm":
                    break
                if mn.startswith("s_cbranch") or mn == "s_branch":
                    tgt = oo[0] if oo else None
                    if tgt in labels:
                        stack.append((labels[tgt], used + 1))
                    if mn == "s_branch":
                        break
                used += (int(oo[0], 0) + 1) if mn == "s_nop" and oo else 1
                j += 1
    return n_mfma, problems


def main(argv):
    if argv:
        text = open(argv[0]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "pvnet_vote.s")
            compile_to_asm(out)
            text = open(out).read()
    funcs = parse_functions(text)
    total, bad = 0, []
    for name, items in funcs.items():
        if not any(k in name for k in KERNELS):
            continue
        n, p = check_function(name, items)
        total += n
        bad += p
    print(f"checked {total} v_mfma instructions in {sum(any(k in f for k in KERNELS) for f in funcs)} scoring kernels")
    for name, i, op, d, j, mn, oo, used, need in bad:
        print(f"HAZARD {name}: {op} {d} (item {i}) -> {mn} {oo} (item {j}) after {used} wait states, {need} needed")
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
