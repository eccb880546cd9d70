"""Runs the REFERENCE'S OWN Python driver (lib/ransac_voting_gpu_layer/ransac_voting_gpu.py, imported from the
reference tree where it lies) on CPU tensors -- TEST INFRASTRUCTURE, build container only (the GPU box has no
reference tree; it gets the fixtures this produces: tests/golden/make_golden.py, G6).

The driver imports a compiled CUDA extension (`lib.ransac_voting_gpu_layer.ransac_voting`).  Here that module is a
stub whose two functions are the plain-C restatement of the kernels (oracle/oracle_c) -- itself checked bit-for-bit
against the reference's device code on the MI355X (tests/test_reference_kernels.py) -- so that every line of the
driver's own logic (gates :531-540, compaction :542-546, arg-max :557-569, confidence loop :571-576, refinement
:579-594) executes as written.  Two names this torch (2.10) dropped are mapped to their replacements while the
driver runs: torch.gesv(b, A) -> torch.linalg.solve(A, b), and uint8 masks in Tensor.masked_select -> bool.
Nothing in the reference tree is modified or copied."""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types

import numpy as np

REFERENCE = os.environ.get("PVNET_REFERENCE", "/root/reference")
_DRIVER = os.path.join(REFERENCE, "lib", "ransac_voting_gpu_layer", "ransac_voting_gpu.py")
_NAMES = ("lib", "lib.ransac_voting_gpu_layer", "lib.ransac_voting_gpu_layer.ransac_voting")


def available() -> bool:
    return os.path.exists(_DRIVER)


class Capture:
    """What the driver handed to the extension: per image the idxs it drew (:547) and the pixels it kept (:542)."""
    def __init__(self):
        self.idxs, self.coords, self.tn, self.rounds = [], [], [], []


@contextlib.contextmanager
def reference_driver(capture: Capture | None = None):
    import torch
    from oracle import cref
    stub = types.ModuleType(_NAMES[2])

    def generate_hypothesis(direct, coords, idxs):
        if capture is not None:
            i_np, c_np = idxs.numpy().copy(), coords.numpy().copy()
            again = bool(capture.idxs) and np.array_equal(capture.idxs[-1], i_np) and \
                np.array_equal(capture.coords[-1], c_np)
            if again:  # the driver's while-loop (:552-576) came round again with the very same idxs (:547)
                capture.rounds[-1] += 1
            else:
                capture.idxs.append(i_np)
                capture.coords.append(c_np)
                capture.tn.append(int(c_np.shape[0]))
                capture.rounds.append(1)
        return torch.from_numpy(cref.generate_hypothesis(direct.contiguous().numpy(), coords.contiguous().numpy(),
                                                         idxs.contiguous().numpy()))

    def voting_for_hypothesis(direct, coords, hypo_pts, inliers, inlier_thresh):
        assert inliers.is_contiguous() and inliers.dtype == torch.uint8
        cref.voting_for_hypothesis(direct.contiguous().numpy(), coords.contiguous().numpy(),
                                   hypo_pts.contiguous().numpy(), inliers.numpy(), float(inlier_thresh))

    stub.generate_hypothesis = generate_hypothesis
    stub.voting_for_hypothesis = voting_for_hypothesis
    pk, pk2 = types.ModuleType(_NAMES[0]), types.ModuleType(_NAMES[1])
    pk.__path__, pk2.__path__ = [], []
    pk.ransac_voting_gpu_layer, pk2.ransac_voting = pk2, stub
    saved = {k: sys.modules.get(k) for k in _NAMES}
    saved_ms, had_gesv = torch.Tensor.masked_select, hasattr(torch, "gesv")
    sys.modules.update(dict(zip(_NAMES, (pk, pk2, stub))))
    torch.Tensor.masked_select = lambda self, mask: saved_ms(self, mask.bool() if mask.dtype == torch.uint8 else mask)
    if not had_gesv:
        torch.gesv = lambda b, A: (torch.linalg.solve(A, b), None)
    try:
        spec = importlib.util.spec_from_file_location("pvnet_reference_ransac_voting_gpu", _DRIVER)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        yield mod
    finally:
        torch.Tensor.masked_select = saved_ms
        if not had_gesv:
            del torch.gesv
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def run_v3(mask: np.ndarray, vertex: np.ndarray, hn: int, *, torch_seed: int = 0, **kw):
    """reference ransac_voting_layer_v3 on numpy inputs -> (key-points [b,vn,2] float32, Capture)."""
    import torch
    cap = Capture()
    with reference_driver(cap) as ref:
        torch.manual_seed(torch_seed)
        out = ref.ransac_voting_layer_v3(torch.from_numpy(np.ascontiguousarray(mask)),
                                         torch.from_numpy(np.ascontiguousarray(vertex)), hn, **kw)
    return out.numpy().astype(np.float32), cap


def run(fn_name: str, mask: np.ndarray, vertex: np.ndarray, *args, torch_seed: int = 0, **kw):
    """Any of the driver's sibling functions (ransac_voting_layer_v5, estimate_voting_distribution_with_mean,
    ransac_motion_voting, generate_hypothesis, ...) on numpy inputs -> (tuple of numpy outputs, Capture).
    numpy positional arguments after ``vertex`` (e.g. ``mean``) are converted to tensors."""
    import torch
    cap = Capture()
    with reference_driver(cap) as ref:
        torch.manual_seed(torch_seed)
        targs = [torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a for a in args]
        out = getattr(ref, fn_name)(torch.from_numpy(np.ascontiguousarray(mask)),
                                    torch.from_numpy(np.ascontiguousarray(vertex)), *targs, **kw)
    if not isinstance(out, (tuple, list)):
        out = (out,)
    return tuple(o.numpy() for o in out), cap
