"""Multi-GPU voting: images shard over ranks, one all-gather of 2-D key-points.

The reference's only parallelism on this path is ``torch.nn.DataParallel`` around the parameter-free
``EvalWrapper`` (tools/train_linemod.py:183-184, tools/demo.py:174): scatter along the batch, vote per device,
gather ``[b_i, vn, 2]`` to device 0 -- single process, one GIL-bound Python thread per GPU.  Here it is one
process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm, over xGMI inside a node): every rank
votes its own contiguous block of images with no data-path collective, and the single exchange is an all-gather
of ``B/G x vn x 2`` float32 (2.3 KB per rank at B/G = 32 -- latency-bound, so ring-vs-tree and the 7 x 153 GB/s
link budget are irrelevant).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous block of images owned by ``rank`` (blocks differ by at most one image)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class RcclGather:
    """An RCCL communicator owned by the LIBRARY's binding (include/pvnet_vote.h: pvnet_rccl_* / pvnet_vote_allgather;
    pvnet_amd/csrc/pvnet_rccl.hip): the all-gather of key-points is one ``ncclAllGather`` on the CURRENT stream -- the stream
    the votes were issued on -- instead of a ``torch.distributed`` collective on the ProcessGroup's internal stream (a fourth
    hardware queue beside the voting streams: -3.5 % on every rank, profiles/r05c_gather_stream_ab.txt).  ``torch.distributed``
    (any backend, here only the bootstrap) broadcasts the 128-byte ncclUniqueId; with no process group the communicator has one rank.

        comm = RcclGather(device)                       # collective over the ranks of `group`
        comm.all_gather(out, local)                     # out [world * n] <- local [n] of every rank, float32, current stream
    """

    def __init__(self, device, group=None, lib_path: Optional[str] = None):
        from . import voting
        self.lib = voting.load_library()
        self.lib.pvnet_rccl_load.restype = C.c_int
        self.lib.pvnet_rccl_load.argtypes = [C.c_char_p]
        self.lib.pvnet_rccl_unique_id.restype = C.c_int
        self.lib.pvnet_rccl_unique_id.argtypes = [C.c_void_p]
        self.lib.pvnet_rccl_comm_init.restype = C.c_int
        self.lib.pvnet_rccl_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]
        self.lib.pvnet_rccl_comm_ranks.restype = C.c_int
        self.lib.pvnet_rccl_comm_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self.lib.pvnet_rccl_comm_destroy.restype = C.c_int
        self.lib.pvnet_rccl_comm_destroy.argtypes = [C.c_void_p]
        self.lib.pvnet_vote_allgather.restype = C.c_int
        self.lib.pvnet_vote_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        self.device = torch.device(device)
        if lib_path is None:  # the librccl the process already has (PyTorch-ROCm ships one), else the system's
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            lib_path = cand if os.path.exists(cand) else ""
        rc = self.lib.pvnet_rccl_load(lib_path.encode() if lib_path else None)
        if rc:
            raise RuntimeError(f"pvnet_rccl_load({lib_path or 'default search'}) failed: {rc} (no librccl on this host?)")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            voting._check(self.lib.pvnet_rccl_unique_id(C.c_void_p(uid.data_ptr())), "pvnet_rccl_unique_id")
        if self.world > 1:
            on_dev = dist.get_backend(group) == "nccl"
            t = uid.to(self.device) if on_dev else uid
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = t.cpu()
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            voting._check(self.lib.pvnet_rccl_comm_init(C.byref(self.comm), self.world, C.c_void_p(uid.data_ptr()), self.rank),
                          "pvnet_rccl_comm_init")

    def ranks(self) -> int:
        n = C.c_int(0)
        from . import voting
        voting._check(self.lib.pvnet_rccl_comm_ranks(self.comm, C.byref(n)), "pvnet_rccl_comm_ranks")
        return int(n.value)

    def all_gather(self, out: torch.Tensor, local: torch.Tensor) -> torch.Tensor:
        if not (out.is_cuda and local.is_cuda and out.dtype == local.dtype == torch.float32 and out.is_contiguous() and
                local.is_contiguous() and out.numel() == self.world * local.numel()):
            raise RuntimeError("RcclGather.all_gather: contiguous float32 CUDA tensors, out.numel() == world * local.numel()")
        rc = self.lib.pvnet_vote_allgather(local.data_ptr(), out.data_ptr(), local.numel(), self.comm,
                                           torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            raise RuntimeError(f"pvnet_vote_allgather failed: {rc}")
        return out

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.lib.pvnet_rccl_comm_destroy(self.comm)
            self.comm = C.c_void_p()


def gather_keypoints(local: torch.Tensor, total: int, group=None, comm: Optional[RcclGather] = None) -> torch.Tensor:
    """all-gather ``[b_local, vn, 2]`` blocks into ``[total, vn, 2]`` (every rank gets the full result).  With ``comm`` (an
    ``RcclGather``) the collective is the library's own ``ncclAllGather`` on the current stream; without, ``torch.distributed``'s
    (any backend: the gloo path of the CPU tests)."""
    world = comm.world if comm is not None else dist.get_world_size(group)
    rank = comm.rank if comm is not None else dist.get_rank(group)
    if world == 1:
        return local

    def all_gather(out, inp):
        if comm is not None:
            comm.all_gather(out, inp)
        else:
            dist.all_gather_into_tensor(out, inp, group=group)  # one collective
    vn = local.shape[1]
    sizes = [shard_range(total, world, r) for r in range(world)]
    if all(e - s == sizes[0][1] - sizes[0][0] for s, e in sizes):
        out = torch.empty((total, vn, 2), dtype=local.dtype, device=local.device)
        all_gather(out, local.contiguous())
        return out
    pad = max(e - s for s, e in sizes)  # ragged tail: pad to the largest block, still one collective
    buf = torch.zeros((pad, vn, 2), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * pad, vn, 2), dtype=local.dtype, device=local.device)
    all_gather(out, buf)
    assert sizes[rank][1] - sizes[rank][0] == local.shape[0]
    return torch.cat([out[r * pad: r * pad + (e - s)] for r, (s, e) in enumerate(sizes)], 0)


def sharded_ransac_voting_layer_v3(mask: torch.Tensor, vertex: torch.Tensor, round_hyp_num: int, *args,
                                   total: Optional[int] = None, group=None, voter: Optional[Callable] = None,
                                   seed: Optional[int] = None, comm: Optional[RcclGather] = None, **kw) -> torch.Tensor:
    """Vote this rank's images and return the key-points of the WHOLE batch on every rank.

    ``mask`` / ``vertex`` hold only this rank's shard (``shard_range(total, world, rank)``); ``total`` defaults to
    ``world * b_local``.  ``voter`` defaults to the HIP layer; (CPU tests inject a checker here).  The per-image
    RNG stream is the GLOBAL image index, so results do not depend on how the batch was sharded -- given ONE seed for
    the whole batch: an explicit ``seed`` is used as it is; with ``seed=None`` rank 0 draws one from its CPU generator
    and broadcasts it (one int64), so the default call is shard-invariant too and ``torch.manual_seed`` on rank 0
    makes it repeatable at any world size."""
    if voter is None:
        from .voting import ransac_voting_layer_v3 as voter
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b_local = mask.shape[0]
    total = world * b_local if total is None else total
    start, end = shard_range(total, world, rank)
    assert end - start == b_local, f"rank {rank} holds {b_local} images but owns [{start},{end})"
    if seed is None:
        t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)  # drawn on every rank (keeps the generators in step) ...
        if world > 1:
            if dist.get_backend(group) == "nccl":
                t = t.to(mask.device)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        seed = int(t.item())  # ... rank 0's draw is the one every rank uses
    kw["seed"] = seed
    kw["image_offset"] = start
    local = voter(mask, vertex, round_hyp_num, *args, **kw)
    return gather_keypoints(local, total, group, comm) if world > 1 else local   # (comm: the library's RCCL on the voting stream)
