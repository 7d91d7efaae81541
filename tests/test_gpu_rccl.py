"""The native RCCL exchange (include/plade_hip.h: plade_comm_*, plade_amd/csrc/comm.hip: librccl opened with dlopen, no torch in
the process; plade_amd/rccl_comm.py) on the one GPU of the test box: a communicator of ONE rank -- unique id, ncclCommInitRank,
ncclAllGather, destroy -- through every use the multi-GPU paths make of it: barrier, reductions, the batch-mode result gather
(code/PLADE/main.cpp:122-148 sharded pair i -> rank i % world) and the candidate shard of one pair (code/PLADE/plade.cpp:547-564)
with the library all-gathering the device-resident counts itself.  (RCCL refuses two ranks on one device, so world > 1 runs
only on the driver's multi-GPU node; the world-2 control flow is covered over gloo / the rendezvous in
tests/test_distributed_gloo.py, tests/test_rendezvous.py, tests/test_gpu_shard.py, tests/test_gpu_bench_world2.py.)"""
import sys

import numpy as np
import pytest

import plade_amd
from plade_amd import rccl_comm
from plade_amd.batch import gather_results
from plade_amd.synth import make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    c, why = rccl_comm.connect(0, 1, 0, None)
    assert c is not None, why
    assert why == "rccl"
    yield c
    c.close()


def test_rccl_communicator_of_one_rank(comm):
    a = np.arange(17 * 5, dtype=np.float32).reshape(5, 17)
    parts = comm.all_gather_array(a)
    assert len(parts) == 1 and np.array_equal(parts[0], a)
    comm.barrier()
    assert comm.all_reduce_max([3.5, -1.0]) == [3.5, -1.0] and comm.all_reduce_sum([4, 2.5]) == [4, 2.5]
    big = np.random.default_rng(0).integers(0, 255, 3 << 20, dtype=np.uint8)        # a second, larger block through the same buffers
    assert np.array_equal(comm.all_gather_array(big)[0], big)
    T = np.stack([np.eye(4, dtype=np.float32) * (i + 1) for i in range(3)])
    Tg, okg = gather_results(T, np.array([True, False, True]), 3, 0, 1, comm=comm)
    assert np.array_equal(Tg, T) and okg.tolist() == [True, False, True]


def test_candidate_shard_over_rccl_returns_the_unsharded_bits(comm):
    """plade_set_candidate_shard_comm: the verification's counts stay in device memory, ONE ncclAllGather on the context's stream
    behind k_overlap, one read-back -- and the registration returns the bits of the unsharded run, every candidate scored."""
    tg, sr, _ = make_pair(120000, seed=7)
    c = plade_amd.Context(0, orient_normals=1, dump=1, max_candidates=2000)
    ok0, T0 = c.registration(tg, sr)
    d0 = c.dump()
    rccl_comm.set_candidate_shard(c, comm)
    ok1, T1 = c.registration(tg, sr)
    d1 = c.dump()
    st = c.stats()
    assert ok0 and ok1 and np.array_equal(T0, T1)
    assert np.array_equal(d0["overlap_counts"], d1["overlap_counts"]) and np.array_equal(d0["scores"], d1["scores"])
    assert st["n_candidates_scored_here"] == len(d1["overlap_counts"]) > 0
    # a group of several pairs switches the axis off for the call (collectives from concurrent threads would mismatch) ...
    res = c.registration_pairs([(tg, sr), (tg, sr)])
    assert all(ok and np.array_equal(T, T0) for ok, T in res)
    assert "n_candidates_scored_here" not in c.stats() and "n_candidates_scored_here" not in c.stats(pair=1)
    # ... and the next single-pair call has it again
    ok2, T2 = c.registration(tg, sr)
    assert ok2 and np.array_equal(T2, T0) and c.stats()["n_candidates_scored_here"] > 0
    rccl_comm.set_candidate_shard(c, None)
    ok3, T3 = c.registration(tg, sr)
    assert ok3 and np.array_equal(T3, T0) and "n_candidates_scored_here" not in c.stats()
    c.close()


def test_candidate_shard_over_rccl_in_a_pairs_call_split_into_one_pair_parts(comm):
    """Advisor r5: a pairs call whose clouds exceed plade_params.group_max_points is registered in consecutive parts; a part of
    ONE pair keeps the candidate shard, and from the second part on it runs on a peer context behind a one-member Combiner --
    the all-gather then has to take its place behind the queued upload and overlap kernels on the lead's stream
    (pipeline.hip: raw_launch).  Every pair returns the bits of the unsharded pair alone, all its candidates scored."""
    pairs = [make_pair(120000, seed=7)[:2], make_pair(120000, seed=8)[:2], make_pair(120000, seed=9)[:2]]
    c = plade_amd.Context(0, orient_normals=1, max_candidates=2000)
    alone = [c.registration(tg, sr) for tg, sr in pairs]
    rccl_comm.set_candidate_shard(c, comm)
    c.set_params(group_max_points=250000)          # 240 000 points per pair: every part holds one pair
    res = c.registration_pairs(pairs)
    assert c.stats()["group_parts"] == 3
    for q, ((ok, T), (ok0, T0)) in enumerate(zip(res, alone)):
        assert ok and ok0 and np.array_equal(T, T0), q
        assert c.stats(pair=q)["n_candidates_scored_here"] > 0, q
    rccl_comm.set_candidate_shard(c, None)
    c.close()


def test_bad_communicator_arguments():
    L = plade_amd.load_library()
    rccl_comm._bind(L)
    import ctypes as C
    h = C.c_void_p()
    ident = np.zeros(128, np.uint8)
    assert L.plade_comm_create(0, 3, 2, ident.ctypes.data_as(C.c_void_p), C.byref(h)) == plade_amd.PLADE_EINVAL   # rank >= world
    assert L.plade_comm_create(0, 0, 1, None, C.byref(h)) == plade_amd.PLADE_EINVAL
    assert L.plade_comm_unique_id(None) == plade_amd.PLADE_EINVAL
