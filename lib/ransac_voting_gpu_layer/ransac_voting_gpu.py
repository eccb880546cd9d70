"""Drop-in module at the reference's import path.

``from lib.ransac_voting_gpu_layer.ransac_voting_gpu import ransac_voting_layer_v3`` (tools/demo.py:8,
tools/train_linemod.py:8-9 of zju3dv/pvnet) resolves here when this repository precedes the reference on
``sys.path``; like the reference tree there is no ``__init__.py`` (implicit namespace packages), so the rest of
``lib.*`` keeps resolving to the reference checkout.  Everything is forwarded to the HIP implementation.
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
