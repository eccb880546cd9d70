import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def demo_fixture():
    """G1: the reference's demo fixture (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "demo_cat.npz"))
    h, w = (int(x) for x in g["shape"])
    mask = np.unpackbits(g["mask_bits"])[: h * w].reshape(h, w)
    return dict(mask=mask, points_2d=g["points_2d"], points_3d=g["points_3d"], pose=g["pose"], K=g["K"],
                bb8_3d=g["bb8_3d"])
