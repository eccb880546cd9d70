"""Host-side pose recovery from the voted 2-D key-points -- SURVEY.md section 8(f) row 1.

The reference does this on the host as well: ``pnp`` = ``cv2.solvePnP(..., SOLVEPNP_ITERATIVE)``
(lib/utils/evaluation_utils.py:19-52), ``uncertainty_pnp`` = P3P initialisation + a Ceres Levenberg-Marquardt on
2x2-weighted reprojection residuals (lib/utils/extend_utils/extend_utils.py:63-114,
lib/utils/extend_utils/src/uncertainty_pnp.cpp:7-92).  Neither OpenCV nor Ceres exists in this image, and this
is a 9-point, 6-parameter problem.  Same algorithm class as OpenCV's ITERATIVE flag: a DLT initialisation (numpy)
followed by Levenberg-Marquardt on the reprojection error over (Rodrigues vector, translation).  The LM is native:
``libpvnet_pnp.so`` (pvnet_amd/csrc/pvnet_pnp.cpp, include/pvnet_pnp.h) exports the reference's own
``uncertainty_pnp(pts2d, pts3d, wgt2d, K, init_rt, result_rt, pn)`` C signature, so the reference's cffi call binds to
it unchanged; ``backend="scipy"`` runs the same problem through scipy's MINPACK LM (the cross-check the tests use).
Also the pose metrics of ``Evaluator`` (evaluation_utils.py:75-134).  Not on the GPU hot path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PNP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpvnet_pnp.so")
_pnp_lib = None


def load_pnp_library() -> C.CDLL:
    """dlopen the in-tree host library; loud failure if it has not been built (python -m pvnet_amd.build)."""
    global _pnp_lib
    if _pnp_lib is None:
        if not os.path.exists(_PNP_LIB_PATH):
            raise RuntimeError(f"pvnet_amd: {_PNP_LIB_PATH} is missing -- build it with `python -m pvnet_amd.build`")
        lib = C.CDLL(_PNP_LIB_PATH)
        dp = C.POINTER(C.c_double)
        lib.uncertainty_pnp.restype = None
        lib.uncertainty_pnp.argtypes = [dp] * 6 + [C.c_int]
        lib.pvnet_pnp_refine.restype = C.c_int
        lib.pvnet_pnp_refine.argtypes = [dp] * 6 + [C.c_int, C.c_int, dp]
        lib.pvnet_pnp_evaluate.argtypes = [dp] * 5 + [C.c_int, dp, dp]
        lib.pvnet_pnp_evaluate.restype = C.c_int
        lib.pvnet_pnp_solve.restype = C.c_int
        lib.pvnet_pnp_solve.argtypes = [dp] * 5 + [C.c_int]
        lib.pvnet_pnp_solve_batch.restype = C.c_int
        lib.pvnet_pnp_solve_batch.argtypes = [dp] * 5 + [C.c_int, C.c_int]
        lib.pvnet_pnp_poses_from_rt.restype = None
        lib.pvnet_pnp_poses_from_rt.argtypes = [dp, dp, C.c_int]
        lib.pvnet_angle_axis_to_matrix.restype = None
        lib.pvnet_angle_axis_to_matrix.argtypes = [dp, dp]
        lib.pvnet_matrix_to_angle_axis.restype = None
        lib.pvnet_matrix_to_angle_axis.argtypes = [dp, dp]
        _pnp_lib = lib
    return _pnp_lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _refine(x0, points_3d, points_2d, K, W, backend):
    """Levenberg-Marquardt on the (optionally 2x2-weighted) reprojection residuals from the pose vector x0[6]."""
    if backend == "scipy":
        from scipy.optimize import least_squares
        return least_squares(_residuals, x0, args=(points_3d, points_2d, K, W), method="lm", xtol=1e-12, ftol=1e-12).x
    if backend != "native":
        raise ValueError(f"unknown backend {backend!r}")
    lib = load_pnp_library()
    p2 = np.ascontiguousarray(points_2d, np.float64)
    p3 = np.ascontiguousarray(points_3d, np.float64)
    Kc = np.ascontiguousarray(K, np.float64)
    x0 = np.ascontiguousarray(x0, np.float64)
    Wc = None if W is None else np.ascontiguousarray(W, np.float64)
    out = np.empty(6, np.float64)
    rc = lib.pvnet_pnp_refine(_dptr(p2), _dptr(p3), None if Wc is None else _dptr(Wc), _dptr(Kc), _dptr(x0),
                              _dptr(out), p2.shape[0], 200, None)
    if rc < 0:
        raise RuntimeError("pvnet_pnp_refine: bad arguments")
    return out

LINEMOD_K = np.array([[572.4114, 0., 325.2611], [0., 573.57043, 242.04899], [0., 0., 1.]])  # base_utils.py:241-243


def cost_function(points_2d, points_3d, weights_2d, camera_matrix, rt, jacobian=True):
    """residuals [2 pn] (and their Jacobian [2 pn, 6] w.r.t. angle-axis + translation) of the native solver's cost function
    at pose ``rt`` [6] -- ``pvnet_pnp_evaluate``: the reference's ``ReprojectionErrorArray`` (uncertainty_pnp.cpp:16-35) and
    what its ``ceres::AutoDiffCostFunction`` derives from it.  ``weights_2d`` [pn, 3] = (wxx, wxy, wyy) or None."""
    lib = load_pnp_library()
    x2 = np.ascontiguousarray(points_2d, np.float64)
    x3 = np.ascontiguousarray(points_3d, np.float64)
    K = np.ascontiguousarray(camera_matrix, np.float64)
    p = np.ascontiguousarray(rt, np.float64)
    W = None if weights_2d is None else np.ascontiguousarray(weights_2d, np.float64)
    pn = x2.shape[0]
    r = np.empty(2 * pn)
    J = np.empty((2 * pn, 6)) if jacobian else None
    rc = lib.pvnet_pnp_evaluate(_dptr(x2), _dptr(x3), None if W is None else _dptr(W), _dptr(K), _dptr(p), pn, _dptr(r),
                                None if J is None else _dptr(J))
    if rc:
        raise RuntimeError(f"pvnet_pnp_evaluate returned {rc}")
    return (r, J) if jacobian else r


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """axis-angle vector -> 3x3 rotation (cv2.Rodrigues)."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3) + _skew(rvec)
    k = rvec / th
    K = _skew(k)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def rodrigues_inv(R: np.ndarray) -> np.ndarray:
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    if np.pi - th < 1e-6:  # near pi: take the axis from the symmetric part
        A = (R + np.eye(3)) / 2
        ax = np.sqrt(np.clip(np.diag(A), 0, None))
        i = int(np.argmax(ax))
        ax = A[i] / ax[i]
        return th * ax / np.linalg.norm(ax)
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], np.float64)


def project(points_3d, pose, K):
    """Projector.project_K (lib/utils/base_utils.py:289-294)."""
    p = points_3d @ pose[:, :3].T + pose[:, 3:].T
    p = p @ K.T
    return p[:, :2] / p[:, 2:]


def _dlt_pose(points_3d, points_2d, K):
    """linear initial guess: DLT on normalised image points, projected onto SO(3)."""
    n = points_3d.shape[0]
    xn = (np.linalg.inv(K) @ np.concatenate([points_2d, np.ones((n, 1))], 1).T).T[:, :2]
    c = points_3d.mean(0)
    s = np.sqrt(((points_3d - c) ** 2).sum(1).mean()) + 1e-12
    X = (points_3d - c) / s  # conditioning
    A = np.zeros((2 * n, 12))
    for i in range(n):
        Xh = np.append(X[i], 1.0)
        A[2 * i, 0:4] = Xh
        A[2 * i, 8:12] = -xn[i, 0] * Xh
        A[2 * i + 1, 4:8] = Xh
        A[2 * i + 1, 8:12] = -xn[i, 1] * Xh
    P = np.linalg.svd(A)[2][-1].reshape(3, 4)
    if np.linalg.det(P[:, :3]) < 0:
        P = -P
    U, S, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = -R
    t = P[:, 3] / S.mean()
    t = t - R @ (c / s) * 1.0  # undo the centring ...
    # ... X = (Xw - c)/s  =>  R X + t' = (R Xw)/s + (t' - R c/s)  => scale whole pose by s
    return R, t * s


def _residuals(x, points_3d, points_2d, K, W=None):
    R = rodrigues(x[:3])
    d = project(points_3d, np.concatenate([R, x[3:, None]], 1), K) - points_2d
    if W is not None:  # uncertainty_pnp.cpp:29-30: r = W d with the symmetric 2x2 W = [[wxx,wxy],[wxy,wyy]]
        d = np.stack([W[:, 0] * d[:, 0] + W[:, 1] * d[:, 1], W[:, 1] * d[:, 0] + W[:, 2] * d[:, 1]], 1)
    return d.ravel()


def pnp(points_3d, points_2d, camera_matrix, method="iterative", init=None, backend="native"):
    """evaluation_utils.py:19-52  ->  [3,4] pose (R | t).  DLT initialisation + LM on the reprojection error."""
    points_3d = np.ascontiguousarray(points_3d, np.float64)
    points_2d = np.ascontiguousarray(points_2d, np.float64)
    K = np.asarray(camera_matrix, np.float64)
    assert points_3d.shape[0] == points_2d.shape[0], "points 3D and points 2D must have same number of vertices"
    if init is None and backend == "native" and points_3d.shape[0] >= 6:  # linear start + LM in one native call
        out = np.empty(6, np.float64)
        Kc = np.ascontiguousarray(K)
        rc = load_pnp_library().pvnet_pnp_solve(_dptr(points_2d), _dptr(points_3d), None, _dptr(Kc), _dptr(out),
                                                points_3d.shape[0])
        if rc >= 0:
            return np.concatenate([rodrigues(out[:3]), out[3:, None]], 1)
    if init is None:
        R0, t0 = _dlt_pose(points_3d, points_2d, K)  # det(R0) > 0 fixes the sign of the homogeneous solution
        x0 = np.concatenate([rodrigues_inv(R0), t0])
    else:
        x0 = np.concatenate([rodrigues_inv(init[:, :3]), init[:, 3]])
    x = _refine(x0, points_3d, points_2d, K, None, backend)
    return np.concatenate([rodrigues(x[:3]), x[3:, None]], 1)


def uncertainty_pnp(points_2d, weights_2d, points_3d, camera_matrix, backend="native"):
    """extend_utils.py:63-114: weights_2d [pn,3] = (wxx, wxy, wyy); initialised from the 4 best-weighted points
    (P3P there; the same 4-point LM fit here), refined by LM on the weighted residuals (uncertainty_pnp.cpp:61-92)."""
    points_2d = np.asarray(points_2d, np.float64)
    points_3d = np.asarray(points_3d, np.float64)
    W = np.asarray(weights_2d, np.float64)
    pose0 = pnp(points_3d, points_2d, camera_matrix, backend=backend)  # robust start (the reference's P3P start needs >= 4 good points)
    x0 = np.concatenate([rodrigues_inv(pose0[:, :3]), pose0[:, 3]])
    x = _refine(x0, points_3d, points_2d, np.asarray(camera_matrix, np.float64), W, backend)
    return np.concatenate([rodrigues(x[:3]), x[3:, None]], 1)


def uncertainty_pnp_v2(points_2d, covars, points_3d, camera_matrix, backend="native"):
    """extend_utils.py:116-165: isotropic weight 1/lambda_max(cov) per key-point (0 when cov[0,0] < 1e-5)."""
    covars = np.asarray(covars, np.float64)
    w = np.array([0.0 if c[0, 0] < 1e-5 else 1.0 / np.max(np.linalg.eigvalsh(c)) for c in covars])
    W = np.stack([w, np.zeros_like(w), w], 1)
    return uncertainty_pnp(points_2d, W, points_3d, camera_matrix, backend=backend)


def pnp_batch(points_3d, points_2d, camera_matrix, weights_2d=None):
    """``pnp`` (or, with ``weights_2d [n,pn,3]``, ``uncertainty_pnp``) for n images that share the object points, in
    one native call: points_2d [n,pn,2] -> poses [n,3,4].  Images whose linear start is degenerate come back as zeros."""
    p2 = np.ascontiguousarray(points_2d, np.float64)
    p3 = np.ascontiguousarray(points_3d, np.float64)
    Kc = np.ascontiguousarray(camera_matrix, np.float64)
    n, pn = p2.shape[0], p2.shape[1]
    assert p3.shape == (pn, 3) and p2.shape == (n, pn, 2)
    Wc = None if weights_2d is None else np.ascontiguousarray(weights_2d, np.float64)
    out = np.empty((n, 6), np.float64)
    rc = load_pnp_library().pvnet_pnp_solve_batch(_dptr(p2), _dptr(p3), None if Wc is None else _dptr(Wc), _dptr(Kc),
                                                  _dptr(out), n, pn)
    if rc < 0:
        raise RuntimeError("pvnet_pnp_solve_batch: bad arguments (needs >= 6 points per image)")
    poses = np.empty((n, 3, 4), np.float64)   # (R | t) per image, zeros where the solve failed: converted natively too (a numpy
    load_pnp_library().pvnet_pnp_poses_from_rt(_dptr(out), _dptr(poses), n)   # loop over 32 poses cost more than solving them)
    return poses


# ---- metrics of Evaluator (evaluation_utils.py:75-134) -----------------------------------------------------
def projection_2d_error(pose_pred, pose_target, model, K, symmetric=False):
    """mean 2-D distance of the projected model points (:75-82); symmetric=True pairs every target point with the
    NEAREST predicted one (projection_2d_sym, :84-91; GPU nearest-neighbour search, pvnet_amd/evaluation.py)"""
    if symmetric:
        from . import evaluation
        return evaluation.projection_2d_error(pose_pred, pose_target, model, K, symmetric=True)
    return float(np.mean(np.linalg.norm(project(model, pose_pred, K) - project(model, pose_target, K), axis=-1)))


def add_error(pose_pred, pose_target, model, symmetric=False):
    """ADD (:95-109); symmetric=True is ADD-S (add_metric_sym, :111-122): nearest-neighbour distances, for the
    symmetric classes (eggbox, glue)"""
    if symmetric:
        from . import evaluation
        return evaluation.add_error(pose_pred, pose_target, model, symmetric=True)
    a = model @ pose_pred[:, :3].T + pose_pred[:, 3]
    b = model @ pose_target[:, :3].T + pose_target[:, 3]
    return float(np.mean(np.linalg.norm(a - b, axis=-1)))


def cm_degree_error(pose_pred, pose_target):
    """(translation error in cm, rotation error in degrees) -- the 5cm5deg metric's two numbers (:126-134)."""
    tr = np.linalg.norm(pose_pred[:, 3] - pose_target[:, 3]) * 100
    c = min(np.trace(pose_pred[:, :3] @ pose_target[:, :3].T), 3.0)
    return float(tr), float(np.rad2deg(np.arccos(np.clip((c - 1.) / 2., -1, 1))))
