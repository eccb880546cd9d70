"""Round 4: the k2_repro_r4.hip variants with their four highest USED registers moved to the top of the allocation (so that each
uses its allocation to the last register, like the failing original), as code objects for co_runner.   python make_r4_top.py out_dir"""
import re
import subprocess
import sys

import asm_variants as A

out = sys.argv[1]
VARS = {"base256": [], "dual": ["-DDUAL"], "nt128": ["-DNT=128"], "nt64": ["-DNT=64"], "nt64_barrier": ["-DNT=64", "-DFORCE_BARRIER"]}
for name, flags in VARS.items():
    s = f"{out}/{name}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", "-S", "--cuda-device-only", "k2_repro_r4.hip", "-o", s] + flags)
    src = open(s).read()
    nf = max(int(x) for x in re.findall(r"\.amdhsa_next_free_vgpr (\d+)", src))
    alloc = (nf + 7) // 8 * 8
    A.build(out, name + "_asbuilt", src)
    if nf % 8 == 0:
        print(name, "already uses its allocation to the top", nf)
        continue
    shift = alloc - nf
    done = False
    for base in range(nf - 4, 7, -1):   # every register >= base moves up by `shift`: the highest used one lands on the last allocated
        up = {r: r + shift for r in range(base, nf)}
        try:
            A.build(out, name + "_top", A.allocation(A.permute(src, up), alloc))
            print(name, f"used {nf}, allocated {alloc}: v{base}..v{nf - 1} -> v{base + shift}..v{alloc - 1}")
            done = True
            break
        except AssertionError:
            continue
    if not done:
        print(name, "no contiguous shift found")
