"""Drop-in module at the reference's import path.

``from lib.ransac_voting_gpu_layer.ransac_voting_gpu import ransac_voting_layer_v3`` (tools/demo.py:8,
tools/train_linemod.py:8-9 of zju3dv/pvnet) resolves here when this repository precedes the reference on
``sys.path``; like the reference tree there is no ``__init__.py`` (implicit namespace packages), so the rest of
``lib.*`` keeps resolving to the reference checkout.

Native (HIP) here: ``ransac_voting_layer_v3`` / ``_v5``, ``estimate_voting_distribution_with_mean``,
``ransac_motion_voting``, ``generate_hypothesis`` -- the functions the reference's tools call.  Every OTHER name of
the reference module (``ransac_voting_layer``, ``_v2`` / ``_v4`` / ``_v6``, ``ransac_voting_center``,
``estimate_voting_distribution``, ``ransac_voting_hypothesis``, ``b_inv``, ...) is resolved lazily by the module
``__getattr__`` below: the reference's own ``ransac_voting_gpu.py`` is loaded from the reference checkout further
down ``sys.path`` with its extension import (``ransac_voting``, :2) bound to the HIP ops, so those functions run their
upstream Python on top of ``generate_hypothesis`` / ``voting_for_hypothesis`` from libpvnet_vote.so -- exactly as well
as upstream's Python runs on the installed torch.  Without a reference checkout on the path the lookup raises an
``AttributeError`` that says so.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pvnet_amd.voting import (estimate_voting_distribution_with_mean, ransac_motion_voting,  # noqa: E402,F401
                              ransac_voting_layer_v3, ransac_voting_layer_v5)
from pvnet_amd.voting import generate_hypothesis_counts as generate_hypothesis  # noqa: E402,F401  (:983-1034)
from pvnet_amd import voting as ransac_voting  # noqa: E402,F401  (the op module the reference imports at :2)


_NATIVE = ("ransac_voting_layer_v3", "ransac_voting_layer_v5", "estimate_voting_distribution_with_mean",
           "ransac_motion_voting", "generate_hypothesis")
_upstream = None


def _load_upstream():
    """the reference's own module of this name, from a `lib/ransac_voting_gpu_layer/` directory that is NOT this one
    (implicit namespace packages: every such directory on sys.path is a portion of the package)."""
    global _upstream
    if _upstream is not None:
        return _upstream
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    for base in sys.path:
        cand = os.path.join(base or ".", "lib", "ransac_voting_gpu_layer", "ransac_voting_gpu.py")
        if os.path.isfile(cand) and os.path.abspath(os.path.dirname(cand)) != here:
            spec = importlib.util.spec_from_file_location(__name__ + "_upstream", cand)
            mod = importlib.util.module_from_spec(spec)
            wb, sys.dont_write_bytecode = sys.dont_write_bytecode, True  # no __pycache__ in somebody else's checkout
            try:
                spec.loader.exec_module(mod)
            finally:
                sys.dont_write_bytecode = wb  # its `import lib.ransac_voting_gpu_layer.ransac_voting` finds the HIP ops
            _upstream = mod
            return mod
    return None


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    up = _load_upstream()
    if up is not None and hasattr(up, name):
        return getattr(up, name)
    raise AttributeError(
        f"{__name__}.{name}: only {', '.join(_NATIVE)} are implemented natively on the HIP layer; other names of the "
        f"reference module are served by the reference's own ransac_voting_gpu.py when a zju3dv/pvnet checkout "
        f"follows this repository on sys.path" + ("" if up is None else f" (it has no attribute {name!r} either)"))
