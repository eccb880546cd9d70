"""Host-side mirror of the reference's voting-layer interface on top of the C-ABI HIP library.

Mirrors (same names, argument meaning, defaults and error behaviour):

* ``ransac_voting_layer_v3``  -- lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:514-598
* ``generate_hypothesis`` / ``voting_for_hypothesis`` ops -- lib/ransac_voting_gpu_layer/src/ransac_voting.cpp:20-55,
  exported by the pybind module ``ransac_voting`` (:102-107)

PyTorch is plumbing only here: device memory, the current stream, dtype/stride bookkeeping.  All compute is
in ``libpvnet_vote.so`` (pvnet_amd/csrc/*.hip -- the stage map is at the top of vote_host.hip -- C ABI in include/pvnet_vote.h) reached through ctypes
(which releases the GIL for the duration of the call).  There is NO CPU fallback: without the library, or with
CPU tensors, these functions raise ``RuntimeError`` exactly like the reference's ``CHECK_CUDA`` does
(src/ransac_voting.cpp:7-9).
"""
from __future__ import annotations

import ctypes as C
import struct
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvnet_vote.so")          # release build: the tuning knobs are constants
DEV_LIB_PATH = os.path.join(_HERE, "libpvnet_vote_dev.so")  # development build (-DPVNET_DEV): PVNET_* environment knobs + every kernel variant
# the knobs of the development build (vote_host.hip, load_tuning): with one of them in the environment the Python front end loads
# libpvnet_vote_dev.so instead of the release library -- the knob tests, the fuzz matrix and the tuning tools work through that
TUNING_KNOBS = ("PVNET_SCORE_MODE", "PVNET_SCORE_WGS_PER_CU", "PVNET_SCORE_HPL", "PVNET_SCORE_CHUNK", "PVNET_COMPACT_KG",
                "PVNET_SCORE_XCD", "PVNET_SCORE_ATOMIC", "PVNET_SCORE_LDS_KB", "PVNET_SCORE_ACC", "PVNET_EXACT_FOLD",
                "PVNET_SCORE_RUNS", "PVNET_SCORE_CULL", "PVNET_CULL_Q_MILLI", "PVNET_DEV_STAGES")

F_LITERAL = 1
F_NO_REFINE = 2
F_VERTEX_F16, F_VERTEX_BF16, F_LOGITS_F16, F_LOGITS_BF16 = 4, 8, 16, 32
F_APPROX = 64        # the round-1/2 "fast" mode: matrix-pipe scoring without the rounding-band re-evaluation
F_BAND_STATS = 128   # development aid: count the re-evaluated cells / literal tests (exact mode)
F_CONCURRENT = 256   # hint: other batches are in flight on other streams (see concurrent_hint)
F_CULL_ALL, F_CULL_NONE = 512, 1024   # exact mode: disc-cull every key-point / none (default: K3 selects per image on the device)
S_SKIPPED, S_SINGULAR, S_NO_INLIER, S_OVERFLOW = 1, 2, 4, 8
NUM_STAGES = 6
STAGE_NAMES = ("mask_bits", "subsample", "compact", "hypotheses", "score", "select_refine")


class Layout(C.Structure):
    """ctypes image of ``PvnetVoteLayout`` (include/pvnet_vote.h)."""
    _fields_ = [(n, C.c_int32) for n in ("b", "h", "w", "vn", "hn", "cap", "words", "chunk", "max_chunks", "hpl",
                                          "hgroups", "hn_pad")] + \
               [(n, C.c_size_t) for n in ("off_ctrl", "off_bits", "off_pix", "off_rec", "off_hyp",
                                          "off_partial", "off_counts", "off_win", "off_seg", "off_items", "off_hypb",
                                          "total_bytes")] + \
               [("nseg", C.c_int32), ("wg_g", C.c_int32), ("wg_s", C.c_int32), ("reserved_", C.c_int32), ("cull", C.c_int32)] + \
               [(n, C.c_size_t) for n in ("off_perm", "off_hyps", "off_cnts", "off_hypc")]


_lib = None
_libs = {}   # path -> loaded library


def _wanted_library() -> str:
    if os.environ.get("PVNET_VOTE_LIB"):   # development aid: an experimental build of the same ABI
        return os.environ["PVNET_VOTE_LIB"]
    return DEV_LIB_PATH if any(os.environ.get(k) not in (None, "") for k in TUNING_KNOBS) else LIB_PATH


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; loud failure if it has not been built (python -m pvnet_amd.build).  The release library unless
    a tuning knob is set in the environment (see TUNING_KNOBS; `reload_tuning()` re-decides after the environment changed)."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = _load(_wanted_library())
    return _lib


def _load(lib_path: str) -> C.CDLL:
    if lib_path in _libs:
        return _libs[lib_path]
    if not os.path.exists(lib_path):
        raise RuntimeError(f"pvnet_amd: HIP library {lib_path} is missing -- build it with "
                           f"`python -m pvnet_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(lib_path)
    i64p, f32p, i32p, u8p = C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p
    lib.pvnet_vote_abi_version.restype = C.c_int
    lib.pvnet_vote_build_info.restype = C.c_char_p
    lib.pvnet_vote_layout.restype = C.c_int
    lib.pvnet_vote_layout.argtypes = [C.c_int] * 6 + [C.POINTER(Layout)]
    lib.pvnet_vote_workspace_bytes.restype = C.c_size_t
    lib.pvnet_vote_workspace_bytes.argtypes = [C.c_int] * 6
    v3_args = [C.c_void_p, C.c_int, i64p, f32p, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
               C.c_int, C.c_int, C.c_uint64, C.c_int, i32p, C.c_uint32, f32p, i32p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pvnet_vote_v3.restype = C.c_int
    lib.pvnet_vote_v3.argtypes = v3_args
    lib.pvnet_vote_v3_logits.restype = C.c_int
    lib.pvnet_vote_v3_logits.argtypes = [f32p, i64p, C.c_int, f32p, i64p] + v3_args[5:]
    lib.pvnet_vote_v3_profiled.restype = C.c_int
    lib.pvnet_vote_v3_profiled.argtypes = v3_args + [C.POINTER(C.c_float)]
    lib.pvnet_vote_v3_stage_repeat.restype = C.c_int
    lib.pvnet_vote_v3_stage_repeat.argtypes = v3_args + [C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.pvnet_generate_hypothesis.restype = C.c_int
    lib.pvnet_generate_hypothesis.argtypes = [f32p, f32p, i32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.pvnet_voting_for_hypothesis.restype = C.c_int
    lib.pvnet_voting_for_hypothesis.argtypes = [f32p, f32p, f32p, u8p, C.c_int, C.c_int, C.c_int, C.c_float,
                                                C.c_void_p]
    lib.pvnet_generate_hypothesis_vanishing_point.restype = C.c_int
    lib.pvnet_generate_hypothesis_vanishing_point.argtypes = lib.pvnet_generate_hypothesis.argtypes
    lib.pvnet_voting_for_hypothesis_vanishing_point.restype = C.c_int
    lib.pvnet_voting_for_hypothesis_vanishing_point.argtypes = lib.pvnet_voting_for_hypothesis.argtypes
    ws_tail = [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pvnet_motion_workspace_bytes.restype = C.c_size_t
    lib.pvnet_motion_workspace_bytes.argtypes = [C.c_int] * 4
    lib.pvnet_motion_voting.restype = C.c_int
    lib.pvnet_motion_voting.argtypes = [C.c_void_p, C.c_int, i64p, f32p, i64p, C.c_int, C.c_int, C.c_int, C.c_int, f32p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pvnet_motion_voting_typed.restype = C.c_int
    lib.pvnet_motion_voting_typed.argtypes = [C.c_void_p, C.c_int, i64p, C.c_void_p, i64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_uint32, f32p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.pvnet_vote_confidence.restype = C.c_int
    lib.pvnet_vote_confidence.argtypes = [f32p, C.c_float, f32p, C.c_uint32] + ws_tail
    lib.pvnet_vote_distribution.restype = C.c_int
    lib.pvnet_vote_distribution.argtypes = [f32p, f32p] + ws_tail
    lib.pvnet_vote_band_margin.restype = C.c_int
    lib.pvnet_vote_band_margin.argtypes = [C.c_float, C.c_void_p] + ws_tail
    lib.pvnet_vote_tuning_reload.restype = None
    lib.pvnet_vote_tuning_reload.argtypes = []
    if lib.pvnet_vote_abi_version() != 9:
        raise RuntimeError("pvnet_amd: libpvnet_vote.so ABI version mismatch; rebuild it")
    _libs[lib_path] = lib
    return lib


def reload_tuning():
    """the PVNET_* tuning environment changed: pick the library again (release without knobs, the development build with) and have
    the development build re-read them (it reads them once, at its first call; the release build's knobs are constants)."""
    global _lib
    _lib = _load(_wanted_library())
    _lib.pvnet_vote_tuning_reload()


def _check(rc: int, what: str):
    if rc == 0:
        return
    names = {-1: "PVNET_E_BADARG", -2: "PVNET_E_WORKSPACE", -3: "PVNET_E_UNSUPPORTED"}
    raise RuntimeError(f"{what} failed: {names.get(rc, 'hipError_t ' + str(rc))}")


_FIELD_FLAGS = {torch.float32: 0, torch.float16: F_VERTEX_F16, torch.bfloat16: F_VERTEX_BF16}
_LOGITS_FLAGS = {torch.float32: 0, torch.float16: F_LOGITS_F16, torch.bfloat16: F_LOGITS_BF16}
_MASK_CODES = {torch.uint8: 0, torch.int8: 0, torch.bool: 0, torch.int16: 1, torch.int32: 2, torch.int64: 3,
               torch.float32: 4}


def vote_layout(b, h, w, vn, hn, max_num) -> Layout:
    L = Layout()
    _check(load_library().pvnet_vote_layout(b, h, w, vn, hn, max_num, C.byref(L)), "pvnet_vote_layout")
    return L


def _prepare(mask, vertex, round_hyp_num, max_num, idxs, convert_mask=True):
    if not (isinstance(mask, torch.Tensor) and isinstance(vertex, torch.Tensor)):
        raise TypeError("mask and vertex must be torch tensors")
    if not vertex.is_cuda:
        raise RuntimeError("vertex must be a CUDA tensor")  # CHECK_CUDA, ransac_voting.cpp:7
    if not mask.is_cuda:
        raise RuntimeError("mask must be a CUDA tensor")
    if mask.device != vertex.device:
        raise RuntimeError("mask and vertex must live on the same device")
    if vertex.dim() != 5 or vertex.shape[-1] != 2:
        raise RuntimeError(f"vertex must be [b,h,w,vn,2], got {tuple(vertex.shape)}")
    b, h, w, vn, _ = vertex.shape
    if tuple(mask.shape) != (b, h, w):
        raise RuntimeError(f"mask must be [b,h,w]={(b, h, w)}, got {tuple(mask.shape)}")
    if vertex.dtype not in _FIELD_FLAGS:  # float32 / float16 / bfloat16 are read in place (autocast backbones emit the latter)
        vertex = vertex.float()
    if convert_mask and mask.dtype not in _MASK_CODES:
        mask = mask.float()  # .byte() of any other float type truncates the same way
    hn = int(round_hyp_num)
    if hn <= 0:
        raise RuntimeError("round_hyp_num must be positive")
    max_num = int(min(max(int(max_num), 0), 2 ** 31 - 1))
    if idxs is not None:
        if not idxs.is_cuda or idxs.device != vertex.device:
            raise RuntimeError("idxs must be a CUDA tensor on the inputs' device")
        if idxs.dim() == 3:
            idxs = idxs.unsqueeze(0).expand(b, -1, -1, -1)
        if tuple(idxs.shape) != (b, hn, vn, 2):
            raise RuntimeError(f"idxs must be [b,hn,vn,2]={(b, hn, vn, 2)}, got {tuple(idxs.shape)}")
        idxs = idxs.to(torch.int32).contiguous()
    return mask, vertex, b, h, w, vn, hn, max_num, idxs


def _strides(t, n):
    return (C.c_int64 * n)(*[int(s) for s in t.stride()])


class _DebugViews(dict):
    """the debug dict of a call; "cull" (did the library disc-cull any key-point?) is worked out on first access"""

    def __missing__(self, key):
        if key == "cull":
            self[key] = bool(self["cull_bits"].any().item())
            return self[key]
        raise KeyError(key)


def _debug_views(ws: torch.Tensor, L: Layout):
    """typed views into the workspace (see PvnetVoteLayout) for tests / visualisation."""
    def view(off, nbytes, dtype, shape):
        return ws[off:off + nbytes].view(dtype).view(*shape)
    b, vn, cap, hp = L.b, L.vn, L.cap, L.hn_pad
    ctrl = view(L.off_ctrl, 4 * 8 * (b + 1), torch.int32, (b + 1, 8))
    cull = {}
    if L.cull:  # disc culling: sorted position -> caller's hypothesis index, the hypotheses in sorted order (culled key-points only)
        cull = dict(perm=view(L.off_perm, 4 * b * vn * hp, torch.int32, (b, vn, hp)),
                    hyps=view(L.off_hyps, 8 * b * vn * hp, torch.float32, (b, vn, hp, 2)))
    return _DebugViews(
        **cull,
        layout=L, ctrl=ctrl, tn0=ctrl[:b, 0], tn=ctrl[:b, 1], nchunks=ctrl[:b, 4], total_items=ctrl[b, 0],
        bits=view(L.off_bits, 8 * b * L.words, torch.int64, (b, L.words)),
        pix=view(L.off_pix, 4 * b * cap, torch.int32, (b, cap)),
        rec=view(L.off_rec, 16 * b * vn * cap, torch.float32, (b, vn, cap, 4)),
        hyp=view(L.off_hyp, 8 * b * vn * hp, torch.float32, (b, vn, hp, 2))[:, :, :L.hn],
        counts=view(L.off_counts, 4 * b * vn * hp, torch.int32, (b, vn, hp))[:, :, :L.hn],
        win=view(L.off_win, 8 * b * vn, torch.int32, (b, vn, 2)),
        # exact mode: the origin of the rounding band per (image, key-point), behind the ctrl rows (band_origin_ptr())
        band_origin=view(L.off_ctrl + 4 * 8 * (b + 1), 8 * b * vn, torch.int32, (b, vn, 2)),
        # which (image, key-point)s the disc-culling kernel scored (kp_cull_ptr(); written by exact-mode calls only)
        cull_bits=view(L.off_ctrl + 4 * 8 * (b + 1) + 8 * b * vn, 4 * b * vn, torch.int32, (b, vn)),
    )


def effective_literal(literal: bool, inlier_thresh: float) -> bool:
    """the scoring mode a call really runs in: the matrix-pipe modes fold tan(acos(thresh)) into their operands and need
    1e-3 <= thresh < 1; outside that range the library scores literally (fill_params, vote_host.hip)."""
    t = float(inlier_thresh)
    t32 = struct.unpack('f', struct.pack('f', t))[0]  # the library compares the float32 it receives
    return bool(literal) or not (struct.unpack('f', struct.pack('f', 1e-3))[0] <= t32 < 1.0)


def mode_flags(literal: bool, approx: bool, inlier_thresh: float) -> int:
    """scoring-mode bits of a call.  Default (neither): EXACT mode -- matrix-pipe scoring whose inlier counts and winners
    equal the reference kernels' (pairs inside the float32 rounding band of kernel.cu:107-125 are re-evaluated in the
    reference's operation order); ``literal``: the reference's order for every pair on the VALU (~10x slower);
    ``approx``: matrix-pipe scoring without the re-evaluation (counts within a few votes of the reference's)."""
    if effective_literal(literal, inlier_thresh):
        return F_LITERAL
    return F_APPROX if approx else _CULL_FLAG


_CULL_FLAG = 0


def set_cull_selection(which):
    """Which key-points the exact mode disc-culls in the calls that follow: None (the default) -- the library selects per image on
    the device; "all" / "none" -- PVNET_F_CULL_ALL / PVNET_F_CULL_NONE.  The inlier counts are the same integers under every
    selection (tests/conftest.py runs the parity modules under two of them); only the time differs."""
    global _CULL_FLAG
    _CULL_FLAG = {None: 0, "all": F_CULL_ALL, "none": F_CULL_NONE}[which]


_last_stream = {}  # device index -> the stream of the previous voting call on that device


def reset_concurrent_hint(dev=None):
    """forget the stream history ``concurrent_hint`` keeps (all devices, or one): the next call with ``concurrent=None`` on a
    device counts as its first.  For callers that switch streams once (a warm-up on a side stream, a profiler) and do not want
    the one call after the switch to run the multi-batch variant; ``concurrent=False`` / ``True`` never consult the history."""
    if dev is None:
        _last_stream.clear()
    else:
        dev = torch.device(dev)
        _last_stream.pop(dev.index if dev.index is not None else torch.cuda.current_device(), None)


def concurrent_hint(dev, concurrent: Optional[bool]) -> int:
    """PVNET_F_CONCURRENT or 0 for a call on the current stream of ``dev``.  The flag never changes a result; it picks the
    variant of the scoring kernel that leaves registers to the small stages of OTHER batches in flight (+3 % throughput with
    six batches on six streams, re-measured in round 5: profiles/r05_ab_runs.txt; -5 % for a batch alone: profiles/r04_ab_runs.txt).  ``concurrent=None`` (the default of the callers): set when this
    call's stream differs from the stream of the previous call on the device -- a caller that alternates streams keeps
    batches in flight; one that stays on a stream (the reference's call sites, DataParallel replicas: one stream per device)
    does not."""
    if concurrent is not None:
        return F_CONCURRENT if concurrent else 0
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    prev = _last_stream.get(key)
    _last_stream[key] = stream
    return F_CONCURRENT if (prev is not None and prev != stream) else 0


def _workspace(workspace, L: Layout, dev) -> torch.Tensor:
    if workspace is None:
        return torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
    if not (isinstance(workspace, torch.Tensor) and workspace.is_cuda and workspace.device == dev and
            workspace.dtype == torch.uint8 and workspace.is_contiguous() and workspace.numel() >= L.total_bytes):
        raise RuntimeError(f"workspace must be a contiguous uint8 CUDA tensor of >= {L.total_bytes} bytes on {dev}")
    if workspace.data_ptr() % 256:
        raise RuntimeError("workspace must be 256-byte aligned")
    return workspace


_RETURN_KW = ("return_status", "return_debug", "stage_times")


def _strip_return_kw(kw: dict) -> dict:
    """the sibling wrappers call v3 with return_debug=True themselves: a caller's return_* keyword would change the
    shape of what comes back, so it is dropped here (their own return values are fixed by the reference)."""
    return {k: v for k, v in kw.items() if k not in _RETURN_KW}


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=30000, *, idxs: Optional[torch.Tensor] = None,
                           seed: Optional[int] = None, image_offset: int = 0, literal: bool = False, approx: bool = False,
                           refine: bool = True, return_status: bool = False, return_debug: bool = False,
                           stage_times: bool = False, band_stats: bool = False,
                           workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                           concurrent: Optional[bool] = None):
    """Drop-in for the reference's ``ransac_voting_layer_v3`` (ransac_voting_gpu.py:514-598).

    :param mask:      [b,h,w]  any integer / bool / float dtype; foreground <=> ``mask.byte() != 0``
    :param vertex:    [b,h,w,vn,2] float32, any strides (the planar permuted view of tools/demo.py:48-50 is
                      read in place, never copied)
    :param round_hyp_num: hypotheses per key-point
    :param inlier_thresh, min_num, max_num: as the reference
    :param confidence, max_iter: accepted and ignored -- the reference re-uses one ``idxs`` draw for every round
                      (:547 is outside the loop at :552), so rounds after the first never change its result.
    :return: [b,vn,2] float32 on ``mask.device``

    Extra keyword-only arguments (never required):
      idxs    int tensor [b,hn,vn,2] (or [hn,vn,2]): pixel-pair indices into each image's raster-ordered
              foreground list, replacing the internal counter RNG (parity runs against the oracle)
      seed    RNG seed (default: drawn from torch's CPU generator, so ``torch.manual_seed`` makes runs repeatable)
      image_offset  global index of this call's first image (RNG stream = image_offset + i): lets a batch sharded
              over GPUs reproduce the unsharded draw exactly
      (default scoring mode: EXACT -- matrix pipe, inlier counts and winners equal to the reference kernels', see mode_flags)
      literal score with the reference's float32 operation order for every pair (bit-exact as well, ~10x slower)
      approx  matrix-pipe scoring without the rounding-band re-evaluation (~10 % faster; counts may differ from the
              reference's float32 kernels by a few votes where the reference's own rounding decides a pair)
      band_stats  (exact mode, with return_debug) also count the re-evaluated cells / literal tests: debug["band_stats"]
      refine  False skips the least-squares refinement (:579-595) and returns the winning hypotheses
      return_status / return_debug / stage_times: also return the per-(image,kp) status bits / typed views of
              the workspace / per-stage GPU milliseconds (synchronises; for bench.py)
      out     a caller-owned float32 CUDA tensor [b,vn,2] (contiguous) to receive the key-points instead of a fresh one
              (e.g. a slot of a staging buffer that a later collective sends)
      workspace  a caller-owned uint8 CUDA tensor of >= ``vote_layout(...).total_bytes`` bytes to use instead of a
              fresh allocation (the caller then guarantees that no other call in flight on another stream uses it);
              ``VotePlan`` wraps this for repeated calls of one shape
      concurrent  throughput hint, never changes a result (``concurrent_hint``): None = set when consecutive calls
              alternate streams
    """
    lib = load_library()
    mask, vertex, b, h, w, vn, hn, max_num, idxs = _prepare(mask, vertex, round_hyp_num, max_num, idxs)
    dev = vertex.device
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    literal = effective_literal(literal, inlier_thresh)
    flags = mode_flags(literal, approx, inlier_thresh) | (0 if refine else F_NO_REFINE) | _FIELD_FLAGS[vertex.dtype] | \
        (F_BAND_STATS if band_stats else 0)
    L = vote_layout(b, h, w, vn, hn, max_num)
    with torch.cuda.device(dev):
        flags |= concurrent_hint(dev, concurrent)
        ws = _workspace(workspace, L, dev)
        if out is None:
            out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
        elif not (out.is_cuda and out.device == dev and out.dtype == torch.float32 and out.is_contiguous() and
                  tuple(out.shape) == (b, vn, 2)):
            raise RuntimeError(f"out must be a contiguous float32 CUDA tensor of shape {(b, vn, 2)} on {dev}")
        status = torch.empty((b, vn), dtype=torch.int32, device=dev) if (return_status or return_debug) else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        args = [C.c_void_p(mask.data_ptr()), _MASK_CODES[mask.dtype], _strides(mask, 3),
                C.c_void_p(vertex.data_ptr()), _strides(vertex, 5), b, h, w, vn, hn, C.c_float(inlier_thresh),
                int(min_num), max_num, C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(image_offset),
                C.c_void_p(idxs.data_ptr()) if idxs is not None else None, flags, C.c_void_p(out.data_ptr()),
                C.c_void_p(status.data_ptr()) if status is not None else None, C.c_void_p(ws.data_ptr()),
                C.c_size_t(L.total_bytes), C.c_void_p(stream)]
        times = None
        if stage_times:
            ms = (C.c_float * NUM_STAGES)()
            _check(lib.pvnet_vote_v3_profiled(*args, ms), "pvnet_vote_v3_profiled")
            times = dict(zip(STAGE_NAMES, [float(x) for x in ms]))
        else:
            _check(lib.pvnet_vote_v3(*args), "pvnet_vote_v3")
    extras = []
    if return_status:
        extras.append(status)
    if return_debug:
        d = _debug_views(ws, L)
        d["literal"] = bool(literal)
        # what the library really ran: without the matrix-pipe buffers (PVNET_SCORE_MODE=0) the default mode is scored literally
        d["mode"] = "literal" if (literal or (not approx and not L.reserved_)) else ("approx" if approx else "exact")
        d["concurrent"] = bool(flags & F_CONCURRENT)
        # disc culling is decided on the device per image: d["cull_bits"] is what the library recorded (not re-derived
        # from the layout); d["cull"] = any key-point culled -- evaluated when it is READ (it synchronises: v5 and the py-level
        # generate_hypothesis go through this dict on every call and lost 37 us to it, profiles/r06p_bench_configs.txt)
        if not (d["mode"] == "exact" and L.cull):
            d["cull"] = False
            d["cull_bits"] = torch.zeros((b, vn), dtype=torch.int32, device=dev)
        if band_stats:  # (cells re-evaluated, literal tests made) of this call; synchronises
            d["band_stats"] = tuple(int(x) for x in d["ctrl"][b, 4:6].tolist())
            # disc culling: (fine steps executed, steps the full exact kernel would have executed); (0, 0) when the call did not cull
            d["cull_stats"] = (int(d["ctrl"][b, 1]), int(d["ctrl"][b, 7]))
        d["status"] = status
        d["seed"] = seed
        d["workspace"] = ws
        d["max_num"] = max_num
        extras.append(d)
    if stage_times:
        extras.append(times)
    return (out, *extras) if extras else out


def band_margin(dbg, inlier_thresh):
    """development aid: the measured safety margin of the exact mode's rounding band on the workspace of a completed
    default-mode call (``dbg`` = its ``return_debug`` dict; same ``inlier_thresh``): every test re-evaluated on the matrix
    pipe as the scoring kernel does it and with the reference's arithmetic.  Returns a dict: ``worst`` = the largest |x|
    among the tests whose matrix-pipe vote differs from the reference's (the kernel trusts x only where |x| >= 1: must be
    < 1), ``disagree`` their number, ``band`` the tests with |x| < 1, ``tests``.  Synchronises."""
    L, ws = dbg["layout"], dbg["workspace"]
    dev = ws.device
    out = torch.zeros((L.b * L.vn, 4), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _check(load_library().pvnet_vote_band_margin(
            C.c_float(inlier_thresh), C.c_void_p(out.data_ptr()), L.b, L.h, L.w, L.vn, L.hn, int(dbg["max_num"]),
            C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
            "pvnet_vote_band_margin")
    o = out.cpu()
    u = o.to(torch.int64) & 0xFFFFFFFF
    return {"worst": float(o[:, 0].contiguous().view(torch.float32).max()), "disagree": int(u[:, 1].sum()),
            "band": int(u[:, 2].sum()), "tests": int(u[:, 3].sum())}


def stage_repeat_ms(mask, vertex, round_hyp_num, inlier_thresh=0.999, min_num=5, max_num=30000, *, stage="score",
                    repeats=200, seed=0, image_offset=0, literal=False, approx=False, both=False):
    """profiling: average GPU milliseconds of ONE stage (a name of STAGE_NAMES) re-launched ``repeats`` times after one
    complete pass (``pvnet_vote_v3_stage_repeat``); synchronises.  For the fast-mode scoring stage the value is the
    kernel's own duration from device clock stamps; ``both=True`` also returns the back-to-back event average."""
    lib = load_library()
    mask, vertex, b, h, w, vn, hn, max_num, _ = _prepare(mask, vertex, round_hyp_num, max_num, None)
    dev = vertex.device
    flags = mode_flags(literal, approx, inlier_thresh) | _FIELD_FLAGS[vertex.dtype]
    L = vote_layout(b, h, w, vn, hn, max_num)
    with torch.cuda.device(dev):
        ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
        ms = (C.c_float * 2)()
        _check(lib.pvnet_vote_v3_stage_repeat(
            C.c_void_p(mask.data_ptr()), _MASK_CODES[mask.dtype], _strides(mask, 3), C.c_void_p(vertex.data_ptr()),
            _strides(vertex, 5), b, h, w, vn, hn, C.c_float(inlier_thresh), int(min_num), max_num,
            C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(image_offset), None, flags, C.c_void_p(out.data_ptr()), None,
            C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes),
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), STAGE_NAMES.index(stage), int(repeats),
            ms), "pvnet_vote_v3_stage_repeat")
    return (float(ms[0]), float(ms[1])) if both else float(ms[0])


class VotePlan:
    """A voting call of ONE shape prepared once: layout, workspace, output tensor and the ctypes argument block are
    built here, so a repeated call costs one ctypes call (five kernel launches) and nothing else on the host -- the
    reference's real call sites vote one image at a time (tools/demo.py:55: hn 512; tools/train_linemod.py:106:
    hn 128, max_num 100), where per-call host work is what the caller feels.

        plan = VotePlan(mask, vertex, 512, inlier_thresh=0.99)     # example tensors fix shape / dtype / strides
        kpts = plan(mask, vertex, seed=3)                          # [b, vn, 2]; valid until the next plan() call

    The plan owns its workspace and its output: calls of one plan must be issued on one stream at a time, and the
    returned tensor is overwritten by the next call (clone it to keep it)."""

    def __init__(self, mask, vertex, round_hyp_num, inlier_thresh=0.999, min_num=5, max_num=30000, *, literal=False,
                 approx=False, refine=True, concurrent=False):
        self.lib = load_library()
        mask, vertex, b, h, w, vn, hn, max_num, _ = _prepare(mask, vertex, round_hyp_num, max_num, None)
        self.key = (mask.dtype, tuple(mask.shape), tuple(mask.stride()), vertex.dtype, tuple(vertex.shape),
                    tuple(vertex.stride()), vertex.device)
        self.dev = vertex.device
        self.layout = vote_layout(b, h, w, vn, hn, max_num)
        self.literal = effective_literal(literal, inlier_thresh)
        flags = mode_flags(self.literal, approx, inlier_thresh) | (0 if refine else F_NO_REFINE) | _FIELD_FLAGS[vertex.dtype] | \
            (F_CONCURRENT if concurrent else 0)  # (a plan is used on one stream at a time: the caller says if others run beside it)
        with torch.cuda.device(self.dev):
            self.workspace = torch.empty(self.layout.total_bytes, dtype=torch.uint8, device=self.dev)
            self.out = torch.empty((b, vn, 2), dtype=torch.float32, device=self.dev)
        self._ms, self._vs = _strides(mask, 3), _strides(vertex, 5)
        self._head = (_MASK_CODES[mask.dtype], self._ms)
        self._mid = (self._vs, b, h, w, vn, hn, C.c_float(inlier_thresh), int(min_num), max_num)
        self._tail = (None, flags, C.c_void_p(self.out.data_ptr()), None, C.c_void_p(self.workspace.data_ptr()),
                      C.c_size_t(self.layout.total_bytes))

    def __call__(self, mask, vertex, seed: int = 0, image_offset: int = 0) -> torch.Tensor:
        if (mask.dtype, tuple(mask.shape), tuple(mask.stride()), vertex.dtype, tuple(vertex.shape),
                tuple(vertex.stride()), vertex.device) != self.key:
            raise RuntimeError("VotePlan: tensors differ in dtype / shape / strides / device from the planned call")
        with torch.cuda.device(self.dev):  # the library launches on the CURRENT device (as ransac_voting_layer_v3 does)
            rc = self.lib.pvnet_vote_v3(mask.data_ptr(), *self._head, vertex.data_ptr(), *self._mid,
                                        seed & 0xFFFFFFFFFFFFFFFF, image_offset, *self._tail,
                                        torch.cuda.current_stream(self.dev).cuda_stream)
        if rc:
            _check(rc, "pvnet_vote_v3")
        return self.out


def debug_dir(dbg) -> torch.Tensor:
    """raw directions [b,vn,cap,2] as the records of a ``return_debug`` result carry them: (x, y, ux, uy) in both
    scoring modes (fast mode stores a zero direction for |u| < 1e-6, which never votes)."""
    return dbg["rec"][..., 2:4]


def ransac_voting_layer_v3_from_logits(seg_pred, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99,
                                       max_iter=20, min_num=5, max_num=30000, *, idxs=None, seed=None,
                                       image_offset=0, literal=False, approx=False, refine=True, workspace=None,
                                       out=None, concurrent=None):
    """``ransac_voting_layer_v3(torch.argmax(seg_pred, 1), vertex, ...)`` with the arg-max fused into the first
    kernel: the class logits ``seg_pred [b,C,h,w]`` float32 are read in place and the int64 mask the reference
    materialises (tools/demo.py:52) never exists.  Same result as the two-step call."""
    lib = load_library()
    if not seg_pred.is_cuda or seg_pred.dim() != 4:
        raise RuntimeError("seg_pred must be a CUDA tensor [b,C,h,w]")
    if seg_pred.dtype not in _LOGITS_FLAGS:
        seg_pred = seg_pred.float()
    b, nc, h, w = seg_pred.shape
    fake_mask = seg_pred[:, 0]  # shape/device checks only
    _, vertex, b, h, w, vn, hn, max_num, idxs = _prepare(fake_mask, vertex, round_hyp_num, max_num, idxs,
                                                         convert_mask=False)
    dev = vertex.device
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    flags = mode_flags(literal, approx, inlier_thresh) | (0 if refine else F_NO_REFINE) | \
        _FIELD_FLAGS[vertex.dtype] | _LOGITS_FLAGS[seg_pred.dtype]
    L = vote_layout(b, h, w, vn, hn, max_num)
    with torch.cuda.device(dev):
        flags |= concurrent_hint(dev, concurrent)
        ws = _workspace(workspace, L, dev)  # caller-owned (reused across calls) or a fresh 171 MB at batch 32
        if out is None:
            out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
        elif not (out.is_cuda and out.device == dev and out.dtype == torch.float32 and out.is_contiguous() and
                  tuple(out.shape) == (b, vn, 2)):
            raise RuntimeError(f"out must be a contiguous float32 CUDA tensor of shape {(b, vn, 2)} on {dev}")
        _check(lib.pvnet_vote_v3_logits(
            C.c_void_p(seg_pred.data_ptr()), _strides(seg_pred, 4), nc, C.c_void_p(vertex.data_ptr()),
            _strides(vertex, 5), b, h, w, vn, hn, C.c_float(inlier_thresh), int(min_num), max_num,
            C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), int(image_offset),
            C.c_void_p(idxs.data_ptr()) if idxs is not None else None, flags, C.c_void_p(out.data_ptr()), None,
            C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes),
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "pvnet_vote_v3_logits")
    return out


class EvalWrapper(torch.nn.Module):
    """The reference's ``EvalWrapper`` (tools/demo.py:46-55, tools/train_linemod.py:94-106) on the HIP layer:
    backbone outputs in, key-points out; with ``use_argmax`` the arg-max is fused into the voting launch."""

    def __init__(self, round_hyp_num=512, inlier_thresh=0.99, max_num=30000):
        super().__init__()
        self.round_hyp_num, self.inlier_thresh, self.max_num = round_hyp_num, inlier_thresh, max_num

    def forward(self, seg_pred, vertex_pred, use_argmax=True):
        vertex_pred = vertex_pred.permute(0, 2, 3, 1)
        b, h, w, vn_2 = vertex_pred.shape
        vertex_pred = vertex_pred.view(b, h, w, vn_2 // 2, 2)
        if use_argmax:
            return ransac_voting_layer_v3_from_logits(seg_pred, vertex_pred, self.round_hyp_num,
                                                      inlier_thresh=self.inlier_thresh, max_num=self.max_num)
        return ransac_voting_layer_v3(seg_pred, vertex_pred, self.round_hyp_num, inlier_thresh=self.inlier_thresh,
                                      max_num=self.max_num)


def _ws_tail(L: Layout, max_num: int, ws: torch.Tensor):
    return [L.b, L.h, L.w, L.vn, L.hn, max_num, C.c_void_p(ws.data_ptr()), C.c_size_t(L.total_bytes),
            C.c_void_p(torch.cuda.current_stream(ws.device).cuda_stream)]


def ransac_voting_layer_v5(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=100, *, conf_thresh=0.999, **kw):
    """Drop-in for the reference's ``ransac_voting_layer_v5`` (ransac_voting_gpu.py:763-858): v3 plus a per-key-point
    confidence = fraction of the (sub-sampled) foreground pixels voting for the refined point at 0.999 (:846-850).
    :return: ([b,vn,2], [b,vn]) float32"""
    max_num = int(min(max(int(max_num), 0), 2 ** 31 - 1))
    out, dbg = ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh, confidence, max_iter, min_num,
                                      max_num, return_debug=True, **_strip_return_kw(kw))
    return out, vote_confidence(dbg, out, conf_thresh)


def vote_confidence(dbg, pts, conf_thresh=0.999):
    """ransac_voting_layer_v5's second output for ANY points ``pts`` [b,vn,2] (ransac_voting_gpu.py:846-850): the fraction of the
    kept pixels of a completed call (``dbg`` = its ``return_debug`` dict) whose direction points at ``pts`` within
    ``conf_thresh``, in the reference's float32 operation order.  Enqueued on the current stream, like the call itself."""
    L, ws = dbg["layout"], dbg["workspace"]
    pts = pts.to(device=ws.device, dtype=torch.float32).contiguous()
    if tuple(pts.shape) != (L.b, L.vn, 2):
        raise RuntimeError(f"pts must be [b,vn,2]={(L.b, L.vn, 2)}, got {tuple(pts.shape)}")
    conf = torch.empty((L.b, L.vn), dtype=torch.float32, device=ws.device)
    with torch.cuda.device(ws.device):
        _check(load_library().pvnet_vote_confidence(C.c_void_p(pts.data_ptr()), C.c_float(conf_thresh),
                                                    C.c_void_p(conf.data_ptr()), F_LITERAL if dbg["literal"] else 0,
                                                    *_ws_tail(L, int(dbg["max_num"]), ws)),
               "pvnet_vote_confidence")
    return conf


def estimate_voting_distribution_with_mean(mask, vertex, mean, round_hyp_num=256, min_hyp_num=4096, topk=128,
                                           inlier_thresh=0.99, min_num=5, max_num=30000, output_hyp=False, **kw):
    """Drop-in for the reference's function of the same name (ransac_voting_gpu.py:333-406): ``min_hyp_num``
    fresh hypotheses per key-point (the reference draws them in ceil(min_hyp_num/round_hyp_num) rounds of
    independent pairs -- one draw of the same total here), their inlier ratios, and the ratio-weighted 2x2
    covariance about ``mean`` [b,vn,2].  Returns ``(mean, cov [b,vn,2,2])``.  ``topk`` is unused upstream too.
    (The reference selects ``mask == 1`` here rather than ``mask.byte() != 0``; identical for 0/1 masks.)"""
    hn = -(-int(min_hyp_num) // int(round_hyp_num)) * int(round_hyp_num)
    max_num = int(min(max(int(max_num), 0), 2 ** 31 - 1))
    _, dbg = ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh, min_num=min_num, max_num=max_num, refine=False,
                                    return_debug=True, **_strip_return_kw(kw))
    L, ws = dbg["layout"], dbg["workspace"]
    mean_c = mean.to(device=ws.device, dtype=torch.float32).contiguous()
    cov = torch.empty((L.b, L.vn, 2, 2), dtype=torch.float32, device=ws.device)
    with torch.cuda.device(ws.device):
        _check(load_library().pvnet_vote_distribution(C.c_void_p(mean_c.data_ptr()), C.c_void_p(cov.data_ptr()),
                                                      *_ws_tail(L, max_num, ws)), "pvnet_vote_distribution")
    if output_hyp:
        return mean, cov, dbg["hyp"].permute(0, 2, 1, 3), dbg["counts"].permute(0, 2, 1)
    return mean, cov


def generate_hypothesis_counts(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                               min_num=5, max_num=30000, **kw):
    """The reference's Python-level ``generate_hypothesis`` (ransac_voting_gpu.py:983-1034, used by the hypothesis
    visualiser of tools/demo.py:120-134): all hypotheses and their inlier counts.
    :return: ([b,hn,vn,2] float32, [b,hn,vn] int64)   (skipped images: zeros)"""
    _, dbg = ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh, min_num=min_num, max_num=max_num,
                                    refine=False, return_debug=True, **_strip_return_kw(kw))
    return dbg["hyp"].permute(0, 2, 1, 3).contiguous(), dbg["counts"].permute(0, 2, 1).contiguous().long()


def ransac_motion_voting(mask, vertex):
    """Drop-in for ransac_voting_gpu.py:960-981: per image, the mean over foreground pixels of (vertex + pixel
    coordinate); zeros for an image without foreground.  HIP: the mask kernel's bit mask, then only the foreground
    vectors of the (strided) field are read and summed in float64 (``pvnet_motion_voting_typed``: float16 / bfloat16
    fields are read in place and widened element by element -- the result equals the float32 call on ``vertex.float()``)."""
    lib = load_library()
    mask, vertex, b, h, w, vn, _, _, _ = _prepare(mask, vertex, 1, 0, None)
    dev = vertex.device
    with torch.cuda.device(dev):
        nbytes = lib.pvnet_motion_workspace_bytes(b, h, w, vn)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
        _check(lib.pvnet_motion_voting_typed(C.c_void_p(mask.data_ptr()), _MASK_CODES[mask.dtype], _strides(mask, 3),
                                             C.c_void_p(vertex.data_ptr()), _strides(vertex, 5), b, h, w, vn,
                                             _FIELD_FLAGS[vertex.dtype], C.c_void_p(out.data_ptr()),
                                             C.c_void_p(ws.data_ptr()), C.c_size_t(nbytes),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "pvnet_motion_voting")
    return out


# ---------------------------------------------------------------------------------------------------------
# op-level functions of the reference's pybind module `ransac_voting` (src/ransac_voting.cpp:102-107)
# ---------------------------------------------------------------------------------------------------------
def _check_input(x, name, dtype):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # CHECK_CUDA
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")  # CHECK_CONTIGUOUS
    if x.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}")  # implicit in .data<T>() of the reference


def generate_hypothesis(direct, coords, idxs):
    """direct [tn,vn,2] f32, coords [tn,2] f32, idxs [hn,vn,2] i32 -> new [hn,vn,2] f32 (ransac_voting.cpp:20-31)."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(idxs, "idxs", torch.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    if coords.shape != (tn, 2) or idxs.shape != (hn, vn, 2) or direct.shape[2] != 2:
        raise RuntimeError("generate_hypothesis: shape mismatch")
    out = torch.empty((hn, vn, 2), dtype=torch.float32, device=direct.device)
    with torch.cuda.device(direct.device):
        _check(load_library().pvnet_generate_hypothesis(
            C.c_void_p(direct.data_ptr()), C.c_void_p(coords.data_ptr()), C.c_void_p(idxs.data_ptr()),
            C.c_void_p(out.data_ptr()), tn, vn, hn, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "pvnet_generate_hypothesis")
    return out


def voting_for_hypothesis(direct, coords, hypo_pts, inliers, inlier_thresh):
    """in place: inliers [hn,vn,tn] uint8 gets a 1 wherever the pixel votes (ransac_voting.cpp:41-55)."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(hypo_pts, "hypo_pts", torch.float32)
    _check_input(inliers, "inliers", torch.uint8)
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    if coords.shape != (tn, 2) or hypo_pts.shape != (hn, vn, 2) or inliers.shape != (hn, vn, tn):
        raise RuntimeError("voting_for_hypothesis: shape mismatch")
    with torch.cuda.device(direct.device):
        _check(load_library().pvnet_voting_for_hypothesis(
            C.c_void_p(direct.data_ptr()), C.c_void_p(coords.data_ptr()), C.c_void_p(hypo_pts.data_ptr()),
            C.c_void_p(inliers.data_ptr()), tn, vn, hn, C.c_float(inlier_thresh),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pvnet_voting_for_hypothesis")
    return None


def generate_hypothesis_vanishing_point(direct, coords, idxs):
    """direct [tn,vn,2] f32, coords [tn,2] f32, idxs [hn,vn,2] i32 -> new [hn,vn,3] f32: homogeneous intersections
    (x, y, z) of the two pixels' rays; z = 0 for parallel rays, (0, 0, 0) where the rays do not meet
    (ransac_voting.cpp:57-75 -> ransac_voting_kernel.cu:170-266)."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(idxs, "idxs", torch.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    if coords.shape != (tn, 2) or idxs.shape != (hn, vn, 2) or direct.shape[2] != 2:
        raise RuntimeError("generate_hypothesis_vanishing_point: shape mismatch")
    out = torch.empty((hn, vn, 3), dtype=torch.float32, device=direct.device)
    with torch.cuda.device(direct.device):
        _check(load_library().pvnet_generate_hypothesis_vanishing_point(
            C.c_void_p(direct.data_ptr()), C.c_void_p(coords.data_ptr()), C.c_void_p(idxs.data_ptr()),
            C.c_void_p(out.data_ptr()), tn, vn, hn, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "pvnet_generate_hypothesis_vanishing_point")
    return out


def voting_for_hypothesis_vanishing_point(direct, coords, hypo_pts, inliers, inlier_thresh):
    """in place: inliers [hn,vn,tn] uint8 gets a 1 wherever the pixel votes for the homogeneous hypothesis
    hypo_pts [hn,vn,3] (ransac_voting.cpp:84-99 -> ransac_voting_kernel.cu:268-351)."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(hypo_pts, "hypo_pts", torch.float32)
    _check_input(inliers, "inliers", torch.uint8)
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    if coords.shape != (tn, 2) or hypo_pts.shape != (hn, vn, 3) or inliers.shape != (hn, vn, tn):
        raise RuntimeError("voting_for_hypothesis_vanishing_point: shape mismatch")
    with torch.cuda.device(direct.device):
        _check(load_library().pvnet_voting_for_hypothesis_vanishing_point(
            C.c_void_p(direct.data_ptr()), C.c_void_p(coords.data_ptr()), C.c_void_p(hypo_pts.data_ptr()),
            C.c_void_p(inliers.data_ptr()), tn, vn, hn, C.c_float(inlier_thresh),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pvnet_voting_for_hypothesis_vanishing_point")
    return None
