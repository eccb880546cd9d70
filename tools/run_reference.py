"""Runs one of the reference's OWN scripts, byte for byte unchanged, on top of this repository's HIP voting layer:

    python tools/run_reference.py /path/to/pvnet/tools/demo.py
    python tools/run_reference.py /path/to/pvnet/tools/train_linemod.py --cfg_file configs/linemod_train.json \
        --linemod_cls cat --test_model

What it does before handing over to the script (`runpy`, `__name__ == "__main__"`, cwd = the reference root, which
the scripts assume): puts this repository and the reference root on sys.path, installs the sys.modules shims for the
third-party packages this image lacks (tools/refshim.py) and pins the two overlay modules
`lib.ransac_voting_gpu_layer.{ransac_voting_gpu, ransac_voting}` so that the script's
`from lib.ransac_voting_gpu_layer.ransac_voting_gpu import ransac_voting_layer_v3` (tools/demo.py:8,
tools/train_linemod.py:8-9) binds the HIP layer.  Weights and datasets are the caller's to provide (Appendix B).
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402


def main(argv):
    if not argv:
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    root = os.path.dirname(os.path.dirname(script))
    refshim.install(root)
    refshim.pin_overlay(root)
    os.chdir(root)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
