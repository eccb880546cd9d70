"""ctypes binding of oracle/_ref/libpvnet_refpnp.so -- the reference's uncertainty-PnP COST FUNCTOR
(lib/utils/extend_utils/src/uncertainty_pnp.cpp:7-35, `ReprojectionErrorArray::operator()`), compiled from the reference tree
against its vendored header-only ceres/jet.h + ceres/rotation.h by `make -C oracle ref` -- TEST INFRASTRUCTURE.

Only tests/ may import this.  `residuals()` evaluates the functor with doubles, `jacobian()` with ceres::Jet<double, 6>: the
very numbers ceres::AutoDiffCostFunction<ReprojectionErrorArray, 2, 6> (:46-47) hands to the reference's solver."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libpvnet_refpnp.so")
_lib = None


def available() -> bool:
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_pnp_build_info.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _args(pts2d, pts3d, wgt2d, K, pose):
    a = [np.ascontiguousarray(x, np.float64) for x in (pts2d, pts3d, wgt2d, K, pose)]
    pn = a[0].shape[0]
    assert a[0].shape == (pn, 2) and a[1].shape == (pn, 3) and a[2].shape == (pn, 3) and a[3].shape == (3, 3) and a[4].shape == (6,)
    return a, pn


def residuals(pts2d, pts3d, wgt2d, K, pose):
    a, pn = _args(pts2d, pts3d, wgt2d, K, pose)
    r = np.empty(2 * pn, np.float64)
    lib().ref_pnp_residuals(*[_p(x) for x in a], C.c_int(pn), _p(r))
    return r


def jacobian(pts2d, pts3d, wgt2d, K, pose):
    a, pn = _args(pts2d, pts3d, wgt2d, K, pose)
    r = np.empty(2 * pn, np.float64)
    J = np.empty((2 * pn, 6), np.float64)
    lib().ref_pnp_jacobian(*[_p(x) for x in a], C.c_int(pn), _p(r), _p(J))
    return r, J
