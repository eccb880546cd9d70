"""Development aid: a variant of the release library for a same-box A/B (tools/ab.py, tools/experiments/stage_ab.py) -- the named translation
units recompiled with extra compiler flags, the rest taken from pvnet_amd/build/*.o of the last regular build.
    python tools/experiments/variant_lib.py _ab/lib_x.so k5_refine.hip:-DPVNET_RT=256 [unit.hip:-Dflag[,-Dflag]] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pvnet_amd import build as B  # noqa: E402

out = os.path.abspath(sys.argv[1])
over = dict(a.split(":", 1) for a in sys.argv[2:])
os.makedirs(os.path.dirname(out), exist_ok=True)
cflags = [f for f in B.flags() if f != "-shared"]
objs = []
for src in B.SRC:
    name = os.path.basename(src)
    obj = os.path.join(B.OBJ_DIR, name + ".o")
    if name in over:
        obj = out + "." + name + ".o"
        subprocess.check_call([B.hipcc_path()] + cflags + over[name].split(",") + ["-c", src, "-o", obj])
    objs.append(obj)
subprocess.check_call([B.hipcc_path(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
