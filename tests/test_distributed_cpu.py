"""world_size-2 (and one world_size-8) gloo tests of the sharding / gather logic (pvnet_amd/distributed.py).  No GPU here, so the voter
is injected: the numpy oracle plays the voter *as the checker's stand-in* -- this tests the N>1 plumbing
(shard ownership, global RNG streams, the single all-gather, ragged tails), not the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ransac_voting_oracle as O
from pvnet_amd import distributed as D
from pvnet_amd import synth


def test_shard_ranges_partition_the_batch():
    for total in (1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            r = [D.shard_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in r]
            assert max(sizes) - min(sizes) <= 1


def _oracle_voter(mask, vertex, hn, *a, seed=0, image_offset=0, **kw):
    out = O.ransac_voting_layer_v3(mask.numpy(), vertex.numpy(), hn, *a, seed=seed, image_offset=image_offset, **kw)
    return torch.from_numpy(out)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mask, planar, _ = synth.make_batch(total, first_index=900, h=64, w=80, radius=9, noise=True)
        vertex = synth.planar_to_vertex_view(planar)
        s, e = D.shard_range(total, world, rank)
        full = D.sharded_ransac_voting_layer_v3(torch.from_numpy(mask[s:e]), torch.from_numpy(vertex[s:e].copy()), 32,
                                                0.99, total=total, voter=_oracle_voter, seed=5)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total", [4, 5])  # even split -> all_gather_into_tensor; ragged tail -> padded gather
def test_two_ranks_reproduce_the_unsharded_batch(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mask, planar, _ = synth.make_batch(total, first_index=900, h=64, w=80, radius=9, noise=True)
    ref = O.ransac_voting_layer_v3(mask, synth.planar_to_vertex_view(planar), 32, 0.99, seed=5)
    for r in range(2):
        assert res[r].shape == (total, 9, 2)
        np.testing.assert_array_equal(res[r], ref)  # same global RNG streams, same order, on every rank


def _worker_cfg3(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, e = D.shard_range(total, world, rank)
        # every rank builds ONLY its own block (global image indices s .. e - 1), as a data loader would hand it over
        mask, planar, _ = synth.make_batch(e - s, first_index=4000 + s, h=48, w=64, radius=7, noise=True)
        vertex = synth.planar_to_vertex_view(planar)
        full = D.sharded_ransac_voting_layer_v3(torch.from_numpy(mask), torch.from_numpy(vertex.copy()), 16, 0.99,
                                                total=total, voter=_oracle_voter, seed=77)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_split_baseline_config3_like_the_north_star():
    """BASELINE configs[3]: batch 256 sharded over 8 ranks -- 32 images per rank, contiguous blocks, ONE all-gather of
    [32, vn, 2] -- must return, on every rank, exactly what the unsharded call returns.  World size 8 on
    gloo / CPU; the voter is the numpy oracle standing in for the HIP layer: this covers the N = 8 plumbing the driver's
    `bench.py --gpus 8` run relies on (shard ownership, RNG streams by GLOBAL image index, the single collective)."""
    world, total = 8, 256
    assert [D.shard_range(total, world, r) for r in range(world)] == [(32 * r, 32 * r + 32) for r in range(world)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg3, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    mask, planar, _ = synth.make_batch(total, first_index=4000, h=48, w=64, radius=7, noise=True)
    ref = O.ransac_voting_layer_v3(mask, synth.planar_to_vertex_view(planar), 16, 0.99, seed=77)
    assert ref.shape == (total, 9, 2)
    for r in range(world):
        np.testing.assert_array_equal(res[r], ref)   # every rank holds the unsharded result, byte for byte


def _seed_echo_voter(mask, vertex, hn, *a, seed=0, image_offset=0, **kw):
    """stub voter: every 'key-point' carries (seed mod 2^20, global image index) -- shows which seed a rank voted with"""
    b, vn = mask.shape[0], vertex.shape[3]
    out = torch.empty((b, vn, 2), dtype=torch.float32)
    out[..., 0] = float(seed % (1 << 20))
    out[..., 1] = torch.arange(image_offset, image_offset + b, dtype=torch.float32)[:, None]
    return out


def _worker_default_seed(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # the ranks' own generators DIFFER: only rank 0's draw may be used
        mask = torch.ones((2, 4, 4), dtype=torch.int64)
        vertex = torch.zeros((2, 4, 4, 3, 2))
        full = D.sharded_ransac_voting_layer_v3(mask, vertex, 8, voter=_seed_echo_voter)  # seed=None
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_default_seed_is_rank0s_draw_on_every_rank():
    """with seed=None every rank used to draw its own seed, so the default call was not shard-invariant;
    now rank 0's draw is broadcast"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_default_seed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0], res[1])
    assert len(np.unique(res[0][..., 0])) == 1  # ONE seed for the whole batch
    np.testing.assert_array_equal(res[0][:, 0, 1], np.arange(4))  # global image indices: RNG streams by global index
    g = torch.Generator().manual_seed(100)
    want = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, generator=g).item()) % (1 << 20)
    assert int(res[0][0, 0, 0]) == want  # and it is rank 0's
