"""Host PnP: known-answer tests on the reference's demo fixture -- pose in, key-points out, pose back."""
import numpy as np
import pytest

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


# ---- the native LM (libpvnet_pnp.so, include/pvnet_pnp.h) against scipy's MINPACK LM on the same problems ------------
def _random_problem(rng, pn=9, noise=0.5):
    r = rng.normal(size=3)
    r *= rng.uniform(0.1, 2.5) / np.linalg.norm(r)
    t = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.6, 1.5)])
    X = rng.uniform(-0.08, 0.08, size=(pn, 3))
    pose = np.concatenate([P.rodrigues(r), t[:, None]], 1)
    x = P.project(X, pose, P.LINEMOD_K) + rng.normal(size=(pn, 2)) * noise
    return X, x, pose


def test_native_lm_matches_scipy_lm():
    rng = np.random.default_rng(5)
    for _ in range(25):
        X, x, _ = _random_problem(rng)
        a = P.pnp(X, x, P.LINEMOD_K, backend="native")
        b = P.pnp(X, x, P.LINEMOD_K, backend="scipy")
        np.testing.assert_allclose(a, b, atol=2e-7)
        s = rng.uniform(0.2, 3.0, size=9)
        th = rng.uniform(0, np.pi, size=9)
        cov = np.stack([np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]]) @ np.diag([q * q, 4 * q * q]) @
                        np.array([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]]) for q, t in zip(s, th)])
        W = np.stack([np.linalg.inv(c) for c in cov])  # full symmetric 2x2 weights (wxx, wxy, wyy)
        W3 = np.stack([W[:, 0, 0], W[:, 0, 1], W[:, 1, 1]], 1)
        a = P.uncertainty_pnp(x, W3, X, P.LINEMOD_K, backend="native")
        b = P.uncertainty_pnp(x, W3, X, P.LINEMOD_K, backend="scipy")
        np.testing.assert_allclose(a, b, atol=2e-6)


def test_cost_function_and_jacobian_equal_the_references_functor():
    """the cost function the native LM minimises is pinned to the REFERENCE'S OWN
    `ReprojectionErrorArray::operator()` (uncertainty_pnp.cpp:16-35), compiled from the reference tree against its
    vendored header-only ceres/jet.h + ceres/rotation.h (oracle/_ref/libpvnet_refpnp.so; the vendored libceres itself
    cannot be linked here).  Residuals (functor on doubles) AND Jacobians (functor on ceres::Jet<double, 6>, i.e. what
    ceres::AutoDiffCostFunction<ReprojectionErrorArray, 2, 6> at :46-47 hands to the solver) on 120 random poses, including
    rotation angles -> 0 (the Taylor branch of ceres::AngleAxisRotatePoint, rotation.h) and -> pi, full symmetric weights."""
    from oracle import refpnp
    if not refpnp.available():
        pytest.skip("oracle/_ref/libpvnet_refpnp.so not built (needs the reference tree: make -C oracle ref)")
    rng = np.random.default_rng(17)
    K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])
    angles = [0.0, 1e-12, 1e-9, 3e-8, 1e-6, 1e-4, 1e-2, np.pi - 1e-3, np.pi - 1e-7, np.pi, np.pi + 1e-4, 4.0]
    worst_r = worst_j = 0.0
    for trial in range(120):
        pn = int(rng.integers(1, 12))
        X = rng.uniform(-0.1, 0.1, size=(pn, 3))
        th = angles[trial] if trial < len(angles) else rng.uniform(0.0, np.pi)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        rt = np.concatenate([axis * th, [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.4, 2.0)]])
        x = rng.uniform(0, 640, size=(pn, 2))
        W = np.stack([rng.uniform(0.1, 5, pn), rng.uniform(-1, 1, pn), rng.uniform(0.1, 5, pn)], 1)  # (wxx, wxy, wyy)
        r_ref, J_ref = refpnp.jacobian(x, X, W, K, rt)
        np.testing.assert_allclose(refpnp.residuals(x, X, W, K, rt), r_ref, rtol=1e-14, atol=1e-12)  # Jet value part = plain evaluation
        r, J = P.cost_function(x, X, W, K, rt)
        sr = np.abs(r_ref).max()
        np.testing.assert_allclose(r, r_ref, rtol=0, atol=1e-12 * max(1.0, sr))
        sj = np.abs(J_ref).max()
        np.testing.assert_allclose(J, J_ref, rtol=0, atol=2e-9 * max(1.0, sj))
        worst_r = max(worst_r, np.abs(r - r_ref).max() / max(1.0, sr))
        worst_j = max(worst_j, np.abs(J - J_ref).max() / max(1.0, sj))
    print(f"cost function vs the reference's functor: residuals {worst_r:.1e}, Jacobians {worst_j:.1e} (relative to the largest entry)")
    # identity weights (what pvnet_pnp_solve's first stage and `pnp` use) = the functor with (1, 0, 1)
    r_id, J_id = P.cost_function(x, X, None, K, rt)
    r_ref, J_ref = refpnp.jacobian(x, X, np.tile([1.0, 0.0, 1.0], (pn, 1)), K, rt)
    np.testing.assert_allclose(r_id, r_ref, atol=1e-10)
    np.testing.assert_allclose(J_id, J_ref, atol=2e-9 * max(1.0, np.abs(J_ref).max()))


def test_reference_c_signature_uncertainty_pnp(demo_fixture):
    """the symbol the reference's cffi stub binds (uncertainty_pnp.cpp:61-69): void, 6 double* + int"""
    import ctypes as C
    lib = P.load_pnp_library()
    f = demo_fixture
    p2 = np.ascontiguousarray(f["points_2d"], np.float64)
    p3 = np.ascontiguousarray(f["points_3d"], np.float64)
    w = np.ascontiguousarray(np.tile([1.0, 0.0, 1.0], (9, 1)))
    K = np.ascontiguousarray(f["K"], np.float64)
    target = f["pose"].astype(np.float64)
    init = np.concatenate([P.rodrigues_inv(target[:, :3]) + 0.05, target[:, 3] + 0.02])  # a perturbed start
    out = np.zeros(6)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    lib.uncertainty_pnp(dp(p2), dp(p3), dp(w), dp(K), dp(init), dp(out), 9)
    np.testing.assert_allclose(P.rodrigues(out[:3]), target[:, :3], atol=1e-5)
    np.testing.assert_allclose(out[3:], target[:, 3], atol=1e-6)


def test_native_angle_axis_maps():
    import ctypes as C
    lib = P.load_pnp_library()
    rng = np.random.default_rng(2)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    for th in [0.0, 1e-9, 1e-4, 0.3, 1.7, 3.0, np.pi - 1e-7]:
        k = rng.normal(size=3)
        aa = np.ascontiguousarray(k / np.linalg.norm(k) * th)
        R = np.zeros(9)
        lib.pvnet_angle_axis_to_matrix(dp(aa), dp(R))
        np.testing.assert_allclose(R.reshape(3, 3), P.rodrigues(aa), atol=1e-12)
        back = np.zeros(3)
        lib.pvnet_matrix_to_angle_axis(dp(R), dp(back))
        R2 = np.zeros(9)
        lib.pvnet_angle_axis_to_matrix(dp(back), dp(R2))
        np.testing.assert_allclose(R2, R, atol=1e-9)


def test_native_lm_rejects_bad_arguments_and_survives_degenerate_input():
    import ctypes as C
    lib = P.load_pnp_library()
    z = np.zeros(27)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    out = np.zeros(6)
    assert lib.pvnet_pnp_refine(dp(z), dp(z), None, dp(z), dp(z), dp(out), 2, 10, None) == -1  # pn < 3
    assert lib.pvnet_pnp_refine(None, dp(z), None, dp(z), dp(z), dp(out), 9, 10, None) == -1
    # all points at the camera centre: evaluation fails, the start pose comes back untouched
    init = np.array([0.1, 0.2, 0.3, 0.0, 0.0, 0.0])
    assert lib.pvnet_pnp_refine(dp(z), dp(z), None, dp(np.ascontiguousarray(P.LINEMOD_K.ravel())), dp(init), dp(out), 9,
                                10, None) == 0
    np.testing.assert_array_equal(out, init)


def test_native_linear_start_and_batch_solve_match_the_numpy_scipy_path():
    """pvnet_pnp_solve (native DLT + LM) against numpy DLT + scipy LM, and the batched entry against per-image calls"""
    rng = np.random.default_rng(11)
    X = rng.uniform(-0.08, 0.08, size=(9, 3))
    x2, poses = [], []
    for _ in range(16):
        r = rng.normal(size=3)
        r *= rng.uniform(0.1, 2.8) / np.linalg.norm(r)
        pose = np.concatenate([P.rodrigues(r), np.array([[rng.uniform(-0.2, 0.2)], [rng.uniform(-0.2, 0.2)],
                                                         [rng.uniform(0.5, 1.5)]])], 1)
        poses.append(pose)
        x2.append(P.project(X, pose, P.LINEMOD_K) + rng.normal(size=(9, 2)) * 0.4)
    x2 = np.stack(x2)
    batch = P.pnp_batch(X, x2, P.LINEMOD_K)
    for i in range(16):
        ref = P.pnp(X, x2[i], P.LINEMOD_K, backend="scipy")
        np.testing.assert_allclose(P.pnp(X, x2[i], P.LINEMOD_K), ref, atol=2e-7)
        np.testing.assert_allclose(batch[i], ref, atol=2e-7)
        assert P.projection_2d_error(batch[i], poses[i], X, P.LINEMOD_K) < 2.0  # 0.4 px noise on 9 points
    W = np.tile([1.0, 0.0, 1.0], (16, 9, 1))
    W[:, 8] = [0.01, 0.0, 0.01]  # one key-point trusted 100x less
    wb = P.pnp_batch(X, x2, P.LINEMOD_K, weights_2d=W)
    for i in range(16):
        np.testing.assert_allclose(wb[i], P.uncertainty_pnp(x2[i], W[i], X, P.LINEMOD_K, backend="scipy"), atol=2e-6)


def test_native_solve_flags_degenerate_input():
    import ctypes as C
    lib = P.load_pnp_library()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    out = np.zeros(6)
    K = np.ascontiguousarray(P.LINEMOD_K)
    z2, z3 = np.zeros((9, 2)), np.zeros((9, 3))
    assert lib.pvnet_pnp_solve(dp(z2), dp(z3), None, dp(K), dp(out), 5) == -1  # the linear start needs 6 points
    assert lib.pvnet_pnp_solve(dp(z2), dp(z3), None, dp(K), dp(out), 9) < 0    # all points coincide


def test_native_library_is_clean_under_asan_ubsan(tmp_path):
    """pvnet_pnp.cpp + a 200-problem driver compiled with -fsanitize=address,undefined: any out-of-bounds access,
    use of uninitialised stack, signed overflow or misaligned access aborts the run"""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "pnp_san")
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-I", os.path.join(root, "include"),
           os.path.join(root, "pvnet_amd", "csrc", "pvnet_pnp.cpp"),
           os.path.join(root, "tests", "native", "pnp_sanitizer_driver.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert built.returncode == 0, built.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "sanitized run done" in run.stdout


def test_batch_solve_on_threads_equals_the_serial_loop(demo_fixture, monkeypatch):
    """pvnet_pnp_solve_batch shares the poses of a batch out over threads (round 6; the serial loop of
    tools/train_linemod.py:210-218): every pose must be bit-identical to the single-thread result, with and without weights,
    including a degenerate image in the middle of a block"""
    rng = np.random.default_rng(5)
    X, K = demo_fixture["points_3d"], demo_fixture["K"]
    n = 37
    p2 = demo_fixture["points_2d"][None] + rng.normal(0, 0.7, size=(n, X.shape[0], 2))
    p2[11] = 0.0   # degenerate: comes back as zeros
    W = np.abs(rng.normal(1.0, 0.3, size=(n, X.shape[0], 3)))
    W[:, :, 1] *= 0.1
    for weights in (None, W):
        monkeypatch.setenv("PVNET_PNP_THREADS", "1")
        serial = P.pnp_batch(X, p2, K, weights)
        for nt in ("2", "5", "8"):
            monkeypatch.setenv("PVNET_PNP_THREADS", nt)
            assert P.pnp_batch(X, p2, K, weights).tobytes() == serial.tobytes()
        monkeypatch.delenv("PVNET_PNP_THREADS")
        assert P.pnp_batch(X, p2, K, weights).tobytes() == serial.tobytes()
    assert not serial[11].any() and serial[10].any()
