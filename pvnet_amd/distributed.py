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

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous block of images owned by ``rank`` (blocks differ by at most one image)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_keypoints(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """all-gather ``[b_local, vn, 2]`` blocks into ``[total, vn, 2]`` (every rank gets the full result)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    vn = local.shape[1]
    sizes = [shard_range(total, world, r) for r in range(world)]
    if all(e - s == sizes[0][1] - sizes[0][0] for s, e in sizes):
        out = torch.empty((total, vn, 2), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # one RCCL collective
        return out
    pad = max(e - s for s, e in sizes)  # ragged tail: pad to the largest block, still one collective
    buf = torch.zeros((pad, vn, 2), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * pad, vn, 2), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    assert sizes[rank][1] - sizes[rank][0] == local.shape[0]
    return torch.cat([out[r * pad: r * pad + (e - s)] for r, (s, e) in enumerate(sizes)], 0)


def sharded_ransac_voting_layer_v3(mask: torch.Tensor, vertex: torch.Tensor, round_hyp_num: int, *args,
                                   total: Optional[int] = None, group=None, voter: Optional[Callable] = None,
                                   seed: Optional[int] = None, **kw) -> torch.Tensor:
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
    return gather_keypoints(local, total, group) if world > 1 else local
