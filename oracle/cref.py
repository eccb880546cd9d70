"""ctypes binding of the plain-C oracle (oracle/oracle_c/pvnet_vote_ref.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpvnet_vote_ref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ref_rng_u32.restype = C.c_uint32
        _lib.ref_rng_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.ref_num_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def generate_hypothesis(direct, coords, idxs):
    direct = np.ascontiguousarray(direct, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    idxs = np.ascontiguousarray(idxs, np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    hyp = np.zeros((hn, vn, 2), np.float32)
    lib().ref_generate_hypothesis(_p(direct, C.c_float), _p(coords, C.c_float), _p(idxs, C.c_int32),
                                  _p(hyp, C.c_float), tn, vn, hn)
    return hyp


def voting_counts(direct, coords, hyp, thresh):
    direct = np.ascontiguousarray(direct, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    hyp = np.ascontiguousarray(hyp, np.float32)
    tn, vn, _ = direct.shape
    hn = hyp.shape[0]
    counts = np.zeros((hn, vn), np.int32)
    lib().ref_voting_counts(_p(direct, C.c_float), _p(coords, C.c_float), _p(hyp, C.c_float),
                            _p(counts, C.c_int32), tn, vn, hn, C.c_float(thresh))
    return counts


def voting_for_hypothesis(direct, coords, hyp, inliers, thresh):
    direct = np.ascontiguousarray(direct, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    hyp = np.ascontiguousarray(hyp, np.float32)
    assert inliers.dtype == np.uint8 and inliers.flags.c_contiguous
    tn, vn, _ = direct.shape
    lib().ref_voting_for_hypothesis(_p(direct, C.c_float), _p(coords, C.c_float), _p(hyp, C.c_float),
                                    _p(inliers, C.c_uint8), tn, vn, hyp.shape[0], C.c_float(thresh))
    return inliers


def generate_hypothesis_vanishing_point(direct, coords, idxs):
    direct = np.ascontiguousarray(direct, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    idxs = np.ascontiguousarray(idxs, np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    hyp = np.zeros((hn, vn, 3), np.float32)
    lib().ref_generate_hypothesis_vanishing_point(_p(direct, C.c_float), _p(coords, C.c_float), _p(idxs, C.c_int32),
                                                  _p(hyp, C.c_float), tn, vn, hn)
    return hyp


def voting_for_hypothesis_vanishing_point(direct, coords, hyp, inliers, thresh):
    direct = np.ascontiguousarray(direct, np.float32)
    coords = np.ascontiguousarray(coords, np.float32)
    hyp = np.ascontiguousarray(hyp, np.float32)
    assert inliers.dtype == np.uint8 and inliers.flags.c_contiguous and hyp.shape[2] == 3
    tn, vn, _ = direct.shape
    lib().ref_voting_for_hypothesis_vanishing_point(_p(direct, C.c_float), _p(coords, C.c_float), _p(hyp, C.c_float),
                                                    _p(inliers, C.c_uint8), tn, vn, hyp.shape[0], C.c_float(thresh))
    return inliers


def vote_v3(fg, vertex, hn, thresh=0.999, min_num=5, max_num=30000, seed=0, idxs=None, return_winners=False,
            image_base=0):
    """fg [b,h,w] bool/uint8, vertex [b,h,w,vn,2] float32 with ANY strides (multiples of 4 bytes); image_base = global
    index of image 0 (its RNG stream), as the product's argument of the same name."""
    fg = np.ascontiguousarray(fg, np.uint8)
    assert vertex.dtype == np.float32
    b, h, w, vn, _ = vertex.shape
    vs = (C.c_int64 * 5)(*[s // 4 for s in vertex.strides])
    out = np.zeros((b, vn, 2), np.float32)
    wi = np.zeros((b, vn), np.int32)
    wc = np.zeros((b, vn), np.int32)
    ip = None
    if idxs is not None:
        idxs = np.ascontiguousarray(np.broadcast_to(idxs, (b, hn, vn, 2)), np.int32)
        ip = _p(idxs, C.c_int32)
    base = C.cast(vertex.ctypes.data, C.POINTER(C.c_float))
    lib().ref_vote_v3_base(_p(fg, C.c_uint8), base, vs, b, h, w, vn, int(hn), C.c_float(thresh), int(min_num),
                           int(max_num), C.c_uint64(seed), int(image_base), ip, _p(out, C.c_float), _p(wi, C.c_int32),
                           _p(wc, C.c_int32))
    return (out, wi, wc) if return_winners else out


def num_threads():
    return lib().ref_num_threads()


def set_num_threads(n):
    lib().ref_set_num_threads(int(n))
