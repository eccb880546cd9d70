"""ctypes binding of oracle/_ref/libpvnet_refkernels*.so -- the reference's OWN CUDA kernels
(lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu:11-351, all four kernels), compiled for gfx950 from the reference tree by
`make -C oracle ref` through the header shim in oracle/ref_kernels/ -- TEST INFRASTRUCTURE.

Only tests/ (GPU parity tests) and tools/ may import this; the product never does.  The libraries are built in the
build container (where /root/reference exists) and travel to the GPU box as git-ignored files; `available()` is
False when they are absent."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = {"off": os.path.join(_HERE, "_ref", "libpvnet_refkernels.so"),
       "fast": os.path.join(_HERE, "_ref", "libpvnet_refkernels_fma.so")}
_libs = {}
REFERENCE = os.environ.get("PVNET_REFERENCE", "/root/reference")


def can_build() -> bool:
    return os.path.exists(os.path.join(REFERENCE, "lib", "ransac_voting_gpu_layer", "src", "ransac_voting_kernel.cu"))


def build():
    """Compile the reference kernels where they lie (needs the reference tree; a no-op request elsewhere)."""
    if can_build():
        subprocess.check_call(["make", "-C", _HERE, "ref", f"REFERENCE={REFERENCE}"], stdout=subprocess.DEVNULL)


def available(contract: str = "off") -> bool:
    return os.path.exists(_SO[contract])


def lib(contract: str = "off"):
    if contract not in _libs:
        L = C.CDLL(_SO[contract])
        L.ref_build_info.restype = C.c_char_p
        L.ref_generate_hypothesis.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
        L.ref_voting_for_hypothesis.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float]
        L.ref_generate_hypothesis_vanishing_point.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
        L.ref_voting_for_hypothesis_vanishing_point.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float]
        _libs[contract] = L
    return _libs[contract]


def _check(t, dtype):
    import torch
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, "contiguous CUDA tensor of the right dtype expected"
    assert torch.cuda.current_device() == t.device.index


def generate_hypothesis(direct, coords, idxs, contract: str = "off"):
    """direct [tn,vn,2] f32, coords [tn,2] f32, idxs [hn,vn,2] i32 (CUDA tensors) -> [hn,vn,2] f32."""
    import torch
    _check(direct, torch.float32); _check(coords, torch.float32); _check(idxs, torch.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    out = torch.empty((hn, vn, 2), dtype=torch.float32, device=direct.device)
    torch.cuda.synchronize()
    rc = lib(contract).ref_generate_hypothesis(direct.data_ptr(), coords.data_ptr(), idxs.data_ptr(), out.data_ptr(),
                                               tn, vn, hn)
    assert rc == 0, f"hip error {rc}"
    return out


def voting_for_hypothesis(direct, coords, hypo_pts, inlier_thresh, contract: str = "off"):
    """-> inliers [hn,vn,tn] uint8 (zero-initialised here, as the reference's caller does)."""
    import torch
    _check(direct, torch.float32); _check(coords, torch.float32); _check(hypo_pts, torch.float32)
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    inl = torch.zeros((hn, vn, tn), dtype=torch.uint8, device=direct.device)
    torch.cuda.synchronize()
    rc = lib(contract).ref_voting_for_hypothesis(direct.data_ptr(), coords.data_ptr(), hypo_pts.data_ptr(),
                                                 inl.data_ptr(), tn, vn, hn, float(inlier_thresh))
    assert rc == 0, f"hip error {rc}"
    return inl


def generate_hypothesis_vanishing_point(direct, coords, idxs, contract: str = "off"):
    """ransac_voting_kernel.cu:231-266 through its own launcher -> [hn,vn,3] f32."""
    import torch
    _check(direct, torch.float32); _check(coords, torch.float32); _check(idxs, torch.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    out = torch.empty((hn, vn, 3), dtype=torch.float32, device=direct.device)
    torch.cuda.synchronize()
    rc = lib(contract).ref_generate_hypothesis_vanishing_point(direct.data_ptr(), coords.data_ptr(), idxs.data_ptr(),
                                                               out.data_ptr(), tn, vn, hn)
    assert rc == 0, f"hip error {rc}"
    return out


def voting_for_hypothesis_vanishing_point(direct, coords, hypo_pts, inlier_thresh, contract: str = "off"):
    """ransac_voting_kernel.cu:313-351 through its own launcher -> inliers [hn,vn,tn] uint8 (zero-initialised here)."""
    import torch
    _check(direct, torch.float32); _check(coords, torch.float32); _check(hypo_pts, torch.float32)
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    assert hypo_pts.shape[2] == 3
    inl = torch.zeros((hn, vn, tn), dtype=torch.uint8, device=direct.device)
    torch.cuda.synchronize()
    rc = lib(contract).ref_voting_for_hypothesis_vanishing_point(direct.data_ptr(), coords.data_ptr(),
                                                                 hypo_pts.data_ptr(), inl.data_ptr(), tn, vn, hn,
                                                                 float(inlier_thresh))
    assert rc == 0, f"hip error {rc}"
    return inl


# ---- the reference's nearest-neighbour launcher (lib/utils/extend_utils/src/nearest_neighborhood.cu:120-160) --------
_SO_NN = {"off": os.path.join(_HERE, "_ref", "libpvnet_refnn.so"), "fast": os.path.join(_HERE, "_ref", "libpvnet_refnn_fma.so")}
_nn_libs = {}


def nn_available(contract: str = "off") -> bool:
    return os.path.exists(_SO_NN[contract])


def find_nearest_point_idx(ref_pts, que_pts, exclude_self: bool = False, contract: str = "off"):
    """numpy [pn1,dim] / [pn2,dim] float32 -> int32 [pn2] through the REFERENCE'S OWN `findNearestPointIdxLauncher`
    (host pointers; it allocates and copies itself), exactly as extend_utils.py:51-58 calls it."""
    import numpy as np
    if contract not in _nn_libs:
        L = C.CDLL(_SO_NN[contract])
        L.ref_nn_build_info.restype = C.c_char_p
        L.findNearestPointIdxLauncher.restype = None
        L.findNearestPointIdxLauncher.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5
        _nn_libs[contract] = L
    ref = np.ascontiguousarray(ref_pts[None], np.float32)
    que = np.ascontiguousarray(que_pts[None], np.float32)
    idxs = np.zeros((1, que.shape[1]), np.int32)
    _nn_libs[contract].findNearestPointIdxLauncher(ref.ctypes.data, que.ctypes.data, idxs.ctypes.data, 1, ref.shape[1],
                                                   que.shape[1], ref.shape[2], 1 if exclude_self else 0)
    return idxs[0]
