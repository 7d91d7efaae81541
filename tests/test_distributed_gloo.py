"""world_size-2 test of the batch-mode sharding + gather (plade_amd/batch.py) on CPU with gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plade_amd.batch import shard, gather_results, sharded_overlap_counts


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_T(i):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [i, 2 * i, -i]
    T[0, 0] = np.float32(np.cos(i))
    return T


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard(n_items, rank, world)
    T = np.stack([_fake_T(i) for i in mine]) if mine else np.zeros((0, 4, 4), np.float32)
    ok = np.array([i % 3 != 0 for i in mine], bool)
    dist.barrier()
    Tg, okg = gather_results(T, ok, n_items, rank, world)
    if rank == 0:
        q.put((Tg, okg))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partitions_everything():
    for n in (0, 1, 5, 64):
        for w in (1, 2, 3, 8):
            allidx = sorted(i for r in range(w) for i in shard(n, r, w))
            assert allidx == list(range(n))


def test_gather_world2_gloo():
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    Tg, okg = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for i in range(n_items):
        assert np.array_equal(Tg[i], _fake_T(i))
        assert okg[i] == (i % 3 != 0)


def test_gather_world1():
    T = np.stack([_fake_T(i) for i in range(3)])
    Tg, okg = gather_results(T, np.array([True, False, True]), 3, 0, 1)
    assert np.array_equal(Tg, T) and okg.tolist() == [True, False, True]


def _fake_counts(src, tgt, T, centers, radius, dist_):
    """Stand-in for seam S3 on CPU: a deterministic function of each candidate alone."""
    return np.array([int(abs(t[0, 3]) * 10) - (1 if c[0] > 50 else 0) for t, c in zip(T, centers)], np.int32)


def _overlap_worker(rank, world, port, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = np.stack([_fake_T(i) for i in range(K)])
    centers = np.stack([[i, 0, 0] for i in range(K)]).astype(np.float32)
    got = sharded_overlap_counts(None, None, None, T, centers, 1.0, 0.1, rank, world, counter=_fake_counts)
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_candidate_sharding_world2_gloo():
    """Single-pair axis: candidates split over 2 ranks, counts all-gathered = the unsharded counts, on every rank."""
    world, K = 2, 101
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    T = np.stack([_fake_T(i) for i in range(K)])
    centers = np.stack([[i, 0, 0] for i in range(K)]).astype(np.float32)
    want = _fake_counts(None, None, T, centers, 1.0, 0.1)
    assert np.array_equal(res[0], want) and np.array_equal(res[1], want)


class _GlooArrayComm:
    """The interface of plade_amd.rccl_comm.RcclComm (one all-gather of equally sized blocks per exchange step) over gloo: what the
    RCCL branches of plade_amd/batch.py see on a multi-GPU node, on CPU."""
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def all_gather_array(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t)
        return [p.numpy() for p in parts]


def _array_comm_worker(rank, world, port, n_items, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = _GlooArrayComm(rank, world)
    mine = shard(n_items, rank, world)
    T = np.stack([_fake_T(i) for i in mine]) if mine else np.zeros((0, 4, 4), np.float32)
    ok = np.array([i % 3 != 0 for i in mine], bool)
    Tg, okg = gather_results(T, ok, n_items, rank, world, comm=comm)       # shards of unequal length: padded to one block size
    Tc = np.stack([_fake_T(i) for i in range(K)])
    centers = np.stack([[i, 0, 0] for i in range(K)]).astype(np.float32)
    counts = sharded_overlap_counts(None, None, None, Tc, centers, 1.0, 0.1, rank, world, counter=_fake_counts, comm=comm)
    q.put((rank, Tg, okg, counts))
    dist.barrier()
    dist.destroy_process_group()


def test_the_rccl_branches_with_equal_blocks_world3_gloo():
    """plade_amd/batch.py over an all_gather_array communicator (RCCL on GPUs: plade_amd/rccl_comm.py): 3 ranks, 7 pairs (shards of 3, 2
    and 2: padded), 11 candidates."""
    world, n_items, K = 3, 7, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_array_comm_worker, args=(r, world, port, n_items, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, Tg, okg, counts = q.get(timeout=120)
        res[r] = (Tg, okg, counts)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    Tg, okg, _ = res[0]
    for i in range(n_items):
        assert np.array_equal(Tg[i], _fake_T(i)) and okg[i] == (i % 3 != 0)
    assert res[1][0] is None and res[2][0] is None
    Tc = np.stack([_fake_T(i) for i in range(K)])
    centers = np.stack([[i, 0, 0] for i in range(K)]).astype(np.float32)
    want = _fake_counts(None, None, Tc, centers, 1.0, 0.1)
    for r in range(world):
        assert np.array_equal(res[r][2], want)
