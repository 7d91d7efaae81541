"""Second sharding axis (plade_set_candidate_shard): the candidates of ONE pair's verification
(code/PLADE/plade.cpp:547-564) split over the ranks of a process group.  Two processes share the one GPU of the box and
exchange over gloo; each runs the whole registration with the REAL verification kernel on its half of the candidates, the
counts are all-reduced inside the library's exchange callback, and both ranks must end with the bits of the unsharded run."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import plade_amd
    from plade_amd.synth import make_pair
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tg, sr, Tgt = make_pair(150000, seed=21)
    ctx = plade_amd.Context(0, orient_normals=1, dump=1, max_candidates=2000)
    ok0, T0 = ctx.registration(tg, sr)                       # unsharded
    d0 = ctx.dump()
    calls = []

    def exchange(values, r, w):                              # every word is filled by exactly one rank, the others hold 0
        t = torch.from_numpy(values.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        values[:] = t.numpy()
        calls.append(len(values))
    ctx.set_candidate_shard(rank, world, exchange, min_candidates=100)
    ok1, T1 = ctx.registration(tg, sr)
    d1 = ctx.dump()
    st = ctx.stats()
    ctx.set_candidate_shard(0, 1)
    ok2, T2 = ctx.registration(tg, sr)
    q.put((rank, bool(ok0), bool(ok1), bool(ok2), T0, T1, T2, d0["overlap_counts"], d1["overlap_counts"], calls,
           st.get("n_candidates_scored_here", -1), st["n_candidates_verified"], "n_candidates_scored_here" in ctx.stats()))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_candidates_of_one_pair_sharded_over_two_ranks():
    import torch.multiprocessing as mp
    world = 2
    c = mp.get_context("spawn")
    q = c.Queue()
    port = _free_port()
    procs = [c.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=500)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        _, ok0, ok1, ok2, T0, T1, T2, c0, c1, calls, here, verified, still = res[r]
        assert ok0 and ok1 and ok2
        assert np.array_equal(T0, T1) and np.array_equal(T0, T2)
        assert np.array_equal(c0, c1)                       # integer overlap counts: bit-exact
        assert verified >= 100 and calls == [2 * int(verified)]
        assert here in (int(verified) // 2, (int(verified) + 1) // 2)
        assert not still                                     # switched off again: nothing was sharded in the third run
    assert np.array_equal(res[0][4], res[1][4])
