"""numpy restatement of the reference's farthest-point sampling -- TEST INFRASTRUCTURE (oracle).

lib/utils/extend_utils/src/farthest_point_sampling.cpp:124-160 (`sample_farthest_points_init_center`), :41-71 (the two
helpers).  Float32 throughout, one rounding per operation.  Pinned by the reference's own source compiled where it lies
(oracle/Makefile `ref` -> oracle/_ref/libpvnet_reffps.so; tests/test_evaluation.py)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libpvnet_reffps.so")


def _sqdist(pts, c):
    d = pts - c[None].astype(np.float32)
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]  # :25, left to right


def farthest_point_sampling_init_center(pts, sn):
    pts = np.ascontiguousarray(pts, np.float32)
    pn = pts.shape[0]
    center = (pts.max(0) + pts.min(0)) * np.float32(0.5)  # :138 (operator/ multiplies by the reciprocal, :21)
    min_dist = np.minimum(_sqdist(pts, center), np.float32(np.finfo(np.float32).max))  # :140-141
    taken = np.zeros(pn, bool)
    idxs = np.zeros(sn, np.int32)

    def next_idx():  # :55-71: strict > from 0, first index on ties, index 0 when nothing is left above 0
        m = np.where(taken, np.float32(-1), min_dist)
        best = int(np.argmax(m))
        return best if m[best] > 0 else 0

    cur = next_idx()  # :149
    for s in range(sn):
        taken[cur] = True
        idxs[s] = cur
        if s < sn - 1:
            d = _sqdist(pts, pts[cur])
            upd = ~taken & (d < min_dist)  # :41-53
            min_dist = np.where(upd, d, min_dist)
            cur = next_idx()
    return idxs


def reference_available():
    return os.path.exists(REF_SO)


def reference_init_center(pts, sn):
    """the reference's own farthest_point_sampling_init_center (its source compiled by g++, oracle/_ref)"""
    L = C.CDLL(REF_SO)
    pts = np.ascontiguousarray(pts, np.float32)
    idxs = np.zeros(sn, np.int32)
    L.farthest_point_sampling_init_center(pts.ctypes.data_as(C.c_void_p), idxs.ctypes.data_as(C.c_void_p), pts.shape[0], sn)
    return idxs
