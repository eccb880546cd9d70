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


# VERDICT r05 "Next" 1: the parity modules of the default mode run under BOTH selections of the exact mode's disc culling -- the
# library's own per-key-point choice and "every key-point culled" -- as a fixture, not as a second job: every equality they assert
# (counts torch.equal to the reference kernel / literal mode, winners, key-points) must hold whichever kernel scored.
CULL_MODULES = ("test_exact_mode", "test_hip_parity")


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in CULL_MODULES and "cull_selection" in metafunc.fixturenames:
        metafunc.parametrize("cull_selection", ["library_selects", "every_keypoint_culled"], indirect=True)


@pytest.fixture
def cull_selection(request, monkeypatch):
    from pvnet_amd import voting
    if getattr(request, "param", "library_selects") == "every_keypoint_culled":
        voting.set_cull_selection("all")   # PVNET_F_CULL_ALL on every call: the RELEASE library under the other selection
        try:
            yield "every_keypoint_culled"
        finally:
            voting.set_cull_selection(None)
    else:
        yield "library_selects"
