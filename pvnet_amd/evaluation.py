"""Host-side mirror of the reference's pose evaluation (lib/utils/evaluation_utils.py) -- SURVEY.md 8(f) row 1.

Same names, argument meaning and recorded quantities as the reference:

* ``find_nearest_point_idx`` / ``find_nearest_point_distance``   extend_utils.py:39-60, evaluation_utils.py:54-62
  -- the brute-force nearest neighbour of the symmetric-object metrics, on the GPU through libpvnet_vote.so
  (``pvnet_nearest_point_idx``, pvnet_amd/csrc/pvnet_nn.hip; replaces nearest_neighborhood.cu:48-117);
* ``Evaluator``   evaluation_utils.py:64-226: ``projection_2d`` / ``projection_2d_sym`` / ``add_metric`` /
  ``add_metric_sym`` / ``cm_degree_5_metric`` / ``evaluate`` / ``evaluate_uncertainty`` / ``evaluate_uncertainty_v2`` /
  ``average_precision`` with the same thresholds (5 px, 10 % of the diameter, 5 cm / 5 deg) and the same recorders.

What differs, and why: the reference's Evaluator pulls object models, diameters and 3-D key-points out of its dataset
classes (``LineModModelDB``, ``VotingType.get_pts_3d``) -- datasets are out of scope here -- so this
Evaluator is constructed with them: ``Evaluator(models={cls: points [n,3]}, diameters={cls: d}, points_3d={cls: [pn,3]},
K=...)``.  PnP is the native host solver of pvnet_amd/pnp.py instead of cv2.solvePnP / Ceres.  No CPU fallback for the
nearest-neighbour search: without a GPU it raises.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import pnp as P

SYMMETRIC_CLASSES = ("eggbox", "glue")  # evaluation_utils.py:153,196,215


def _nn_lib():
    from . import voting
    lib = voting.load_library()
    if not getattr(lib, "_nn_ready", False):
        lib.pvnet_nearest_workspace_bytes.restype = C.c_size_t
        lib.pvnet_nearest_workspace_bytes.argtypes = [C.c_int, C.c_int]
        lib.pvnet_nearest_point_idx.restype = C.c_int
        lib.pvnet_nearest_point_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        lib._nn_ready = True
    return lib


def nearest_point_idx(ref_pts, que_pts, exclude_self=False):
    """torch CUDA tensors in, int32 CUDA tensor out: ref_pts [b,pn1,dim] or [pn1,dim], que_pts likewise (dim 2 or 3);
    for every query the index of the nearest reference point (float32 squared distance, first index on ties).
    Enqueues on the current stream; nothing is copied to the host."""
    import torch
    from .voting import _check
    lib = _nn_lib()
    squeeze = ref_pts.dim() == 2
    if squeeze:
        ref_pts, que_pts = ref_pts[None], que_pts[None]
    if not (ref_pts.is_cuda and que_pts.is_cuda):
        raise RuntimeError("nearest_point_idx: CUDA tensors required (there is no CPU fallback)")
    if ref_pts.dim() != 3 or que_pts.dim() != 3 or ref_pts.shape[0] != que_pts.shape[0] or \
            ref_pts.shape[2] != que_pts.shape[2] or ref_pts.shape[2] not in (2, 3):
        raise RuntimeError("nearest_point_idx: ref_pts [b,pn1,dim], que_pts [b,pn2,dim], dim 2 or 3")
    ref = ref_pts.to(torch.float32).contiguous()
    que = que_pts.to(torch.float32).contiguous()
    b, pn1, dim = ref.shape
    pn2 = que.shape[1]
    idxs = torch.empty((b, pn2), dtype=torch.int32, device=ref.device)
    if pn2 == 0:
        return idxs[0] if squeeze else idxs
    with torch.cuda.device(ref.device):
        nbytes = lib.pvnet_nearest_workspace_bytes(b, pn2)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
        _check(lib.pvnet_nearest_point_idx(ref.data_ptr(), que.data_ptr(), idxs.data_ptr(), b, pn1, pn2, dim,
                                           1 if exclude_self else 0, ws.data_ptr(), nbytes,
                                           torch.cuda.current_stream(ref.device).cuda_stream), "pvnet_nearest_point_idx")
    return idxs[0] if squeeze else idxs


def find_nearest_point_idx(ref_pts, que_pts):
    """the reference's function (extend_utils.py:39-60): numpy [pn1,dim] / [pn2,dim] in, numpy int32 [pn2] out"""
    import torch
    ref_pts, que_pts = np.asarray(ref_pts), np.asarray(que_pts)
    assert ref_pts.shape[1] == que_pts.shape[1] and 1 < que_pts.shape[1] <= 3
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("find_nearest_point_idx needs a GPU (there is no CPU fallback)")
    r = torch.from_numpy(np.ascontiguousarray(ref_pts, np.float32)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(que_pts, np.float32)).to(dev)
    return nearest_point_idx(r, q).cpu().numpy()


def find_nearest_point_distance(pts1, pts2):
    """evaluation_utils.py:54-62: for every point of pts2 its distance to the nearest point of pts1"""
    idxs = find_nearest_point_idx(pts1, pts2)
    return np.linalg.norm(np.asarray(pts1)[idxs] - np.asarray(pts2), 2, 1)


def _transform(model, pose):
    return np.dot(model, pose[:, :3].T) + pose[:, 3]


def add_error(pose_pred, pose_target, model, symmetric=False):
    """mean model-point distance between two poses: ADD (:95-109), or ADD-S with nearest neighbours (:111-122)"""
    a, b = _transform(model, pose_pred), _transform(model, pose_target)
    if symmetric:
        return float(np.mean(find_nearest_point_distance(a, b)))
    return float(np.mean(np.linalg.norm(a - b, axis=-1)))


def projection_2d_error(pose_pred, pose_target, model, K, symmetric=False):
    a, b = P.project(model, pose_pred, K), P.project(model, pose_target, K)
    if symmetric:
        return float(np.mean(find_nearest_point_distance(a, b)))
    return float(np.mean(np.linalg.norm(a - b, axis=-1)))


# the reference's `Projector.intrinsic_matrix` (lib/utils/base_utils.py:240-250), which `Evaluator.evaluate*` index by
# `intri_type` (evaluation_utils.py:146-149, :182-185, :204): 'blender' (the DEFAULT of all three) is fx = fy = 700,
# c = (320, 240) -- NOT the LINEMOD camera
INTRINSIC_MATRIX = {
    "linemod": np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]]),
    "blender": np.array([[700.0, 0.0, 320.0], [0.0, 700.0, 240.0], [0.0, 0.0, 1.0]]),
    "pascal": np.array([[-3000.0, 0.0, 0.0], [0.0, 3000.0, 0.0], [0.0, 0.0, 1.0]]),
}


class Evaluator(object):
    """evaluation_utils.py:64-226 with the dataset look-ups replaced by constructor arguments (see module docstring).
    ``K`` (optional) REPLACES the whole intrinsics table: every ``intri_type`` then resolves to it -- for callers whose
    camera is none of the reference's three; without it the table is the reference's own."""

    def __init__(self, models=None, diameters=None, points_3d=None, K=None):
        self.models = dict(models or {})
        self.diameters = dict(diameters or {})
        self.points_3d = dict(points_3d or {})
        self.K = None if K is None else np.asarray(K, np.float64)
        self.intrinsic_matrix = {k: v.copy() for k, v in INTRINSIC_MATRIX.items()}
        self.projection_2d_recorder = []
        self.add_recorder = []
        self.cm_degree_5_recorder = []
        self.proj_mean_diffs = []
        self.add_dists = []
        self.uncertainty_pnp_cost = []

    # ---- the five metric recorders, names and thresholds as the reference ------------------------------------
    def projection_2d(self, pose_pred, pose_targets, model, K, threshold=5):
        d = projection_2d_error(pose_pred, pose_targets, model, K)
        self.proj_mean_diffs.append(d)
        self.projection_2d_recorder.append(d < threshold)

    def projection_2d_sym(self, pose_pred, pose_targets, model, K, threshold=5):
        d = projection_2d_error(pose_pred, pose_targets, model, K, symmetric=True)
        self.proj_mean_diffs.append(d)
        self.projection_2d_recorder.append(d < threshold)

    def add_metric(self, pose_pred, pose_targets, model, diameter, percentage=0.1):
        d = add_error(pose_pred, pose_targets, model)
        self.add_recorder.append(d < diameter * percentage)
        self.add_dists.append(d)

    def add_metric_sym(self, pose_pred, pose_targets, model, diameter, percentage=0.1):
        d = add_error(pose_pred, pose_targets, model, symmetric=True)
        self.add_recorder.append(d < diameter * percentage)
        self.add_dists.append(d)

    def cm_degree_5_metric(self, pose_pred, pose_targets):
        tr, rot = P.cm_degree_error(pose_pred, pose_targets)
        self.cm_degree_5_recorder.append(tr < 5 and rot < 5)

    # ---- evaluate* : PnP + the metrics of one image (:136-217) ------------------------------------------------
    def _intrinsics(self, intri_type, intri_matrix=None):
        """the camera matrix as the reference resolves it (:146-149): the caller's matrix for 'use_intrinsic', else the
        table entry of `intri_type` ('blender' by default); an unknown type raises, as the reference's dict look-up does"""
        if intri_type == "use_intrinsic" and intri_matrix is not None:
            return np.asarray(intri_matrix, np.float64)
        if self.K is not None:
            return self.K
        if intri_type not in self.intrinsic_matrix:
            raise KeyError(f"unknown intri_type {intri_type!r} (known: {sorted(self.intrinsic_matrix)} or 'use_intrinsic' "
                           f"with intri_matrix)")
        return self.intrinsic_matrix[intri_type]

    def _record(self, pose_pred, pose_targets, class_type, K, sym_projection=False):
        model, diameter = self.models[class_type], self.diameters[class_type]
        sym = class_type in SYMMETRIC_CLASSES
        if sym:
            self.add_metric_sym(pose_pred, pose_targets, model, diameter)
        else:
            self.add_metric(pose_pred, pose_targets, model, diameter)
        if sym and sym_projection:
            self.projection_2d_sym(pose_pred, pose_targets, model, K)
        else:
            self.projection_2d(pose_pred, pose_targets, model, K)
        self.cm_degree_5_metric(pose_pred, pose_targets)

    def evaluate(self, points_2d, pose_targets, class_type, intri_type="blender", vote_type=None, intri_matrix=None):
        K = self._intrinsics(intri_type, intri_matrix)
        pose_pred = P.pnp(self.points_3d[class_type], np.asarray(points_2d, np.float64), K)
        self._record(pose_pred, np.asarray(pose_targets, np.float64), class_type, K)
        return pose_pred

    def evaluate_uncertainty(self, mean_pts2d, covar, pose_targets, class_type, intri_type="blender", vote_type=None,
                             intri_matrix=None):
        begin = time.time()
        covar = np.asarray(covar, np.float64)
        cov_invs = []
        for vi in range(covar.shape[0]):  # :169-177: inverse matrix square root of every 2x2 covariance
            if covar[vi, 0, 0] < 1e-6 or np.isnan(covar[vi]).any():
                cov_invs.append(np.zeros((2, 2)))
                continue
            w, v = np.linalg.eigh(covar[vi])
            cov_invs.append((v / np.sqrt(np.maximum(w, 1e-30))) @ v.T)
        weights = np.asarray(cov_invs).reshape(-1, 4)[:, (0, 1, 3)]
        K = self._intrinsics(intri_type, intri_matrix)
        pose_pred = P.uncertainty_pnp(np.asarray(mean_pts2d, np.float64), weights, self.points_3d[class_type], K)
        self.uncertainty_pnp_cost.append(time.time() - begin)
        self._record(pose_pred, np.asarray(pose_targets, np.float64), class_type, K)
        return pose_pred

    def evaluate_uncertainty_v2(self, mean_pts2d, covar, pose_targets, class_type, intri_type="blender", vote_type=None):
        K = self._intrinsics(intri_type)  # :204: the table entry of intri_type (no 'use_intrinsic' branch here upstream)
        pose_pred = P.uncertainty_pnp_v2(np.asarray(mean_pts2d, np.float64), np.asarray(covar, np.float64),
                                         self.points_3d[class_type], K)
        self._record(pose_pred, np.asarray(pose_targets, np.float64), class_type, K, sym_projection=True)
        return pose_pred

    def average_precision(self, verbose=True):
        """:219-226 (the reference also dumps proj_mean_diffs to ./tmp.npy; not reproduced)"""
        r = (float(np.mean(self.projection_2d_recorder)), float(np.mean(self.add_recorder)),
             float(np.mean(self.cm_degree_5_recorder)))
        if verbose:
            print("2d projections metric: {}".format(r[0]))
            print("ADD metric: {}".format(r[1]))
            print("5 cm 5 degree metric: {}".format(r[2]))
        return r
