"""Host PnP: known-answer tests on the reference's demo fixture -- pose in, key-points out, pose back."""
import numpy as np

from pvnet_amd import pnp as P


def test_rodrigues_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        r = rng.normal(size=3)
        r *= rng.uniform(0, 3.1) / np.linalg.norm(r)
        R = P.rodrigues(r)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(P.rodrigues(P.rodrigues_inv(R)), R, atol=1e-9)


def test_pnp_recovers_demo_pose(demo_fixture):
    f = demo_fixture
    pose = P.pnp(f["points_3d"], f["points_2d"], f["K"])
    np.testing.assert_allclose(pose, f["pose"], atol=1e-5)  # exact key-points -> exact pose (tools/demo.py:176-179)
    tr, rot = P.cm_degree_error(pose, f["pose"].astype(np.float64))
    assert tr < 1e-3 and rot < 0.1  # cat_pose.npy is float32 with ~6 digits: its R is orthonormal to ~1e-6 only
    assert P.projection_2d_error(pose, f["pose"].astype(np.float64), f["bb8_3d"], f["K"]) < 1e-3


def test_pnp_with_noise_and_uncertainty(demo_fixture):
    f = demo_fixture
    rng = np.random.default_rng(1)
    sig = np.array([0.3] * 8 + [6.0])  # one badly localised key-point
    noisy = f["points_2d"] + rng.normal(size=(9, 2)) * 0.3
    noisy[8] = f["points_2d"][8] + [9.0, -7.0]  # the outlier the covariance warns about
    plain = P.pnp(f["points_3d"], noisy, f["K"])
    cov = np.stack([np.eye(2) * s * s for s in sig])
    weighted = P.uncertainty_pnp_v2(noisy, cov, f["points_3d"], f["K"])
    target = f["pose"].astype(np.float64)
    e_plain = P.projection_2d_error(plain, target, f["bb8_3d"], f["K"])
    e_w = P.projection_2d_error(weighted, target, f["bb8_3d"], f["K"])
    assert e_plain < 5.0 and e_w < e_plain  # down-weighting the bad point helps (the paper's uncertainty-driven PnP)
