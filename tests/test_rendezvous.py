"""The ranks' torch-free exchange (plade_amd/rendezvous.py) with world sizes 2 and 3 on CPU: rendezvous through the port
file, barrier, all_gather / gather / reductions, the batch-mode result gather (plade_amd/batch.py; pairs are sharded round
robin like code/PLADE/main.cpp:97-158's loop would be split) and candidate sharding -- without torch in the ranks."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

from plade_amd.batch import gather_results, shard, sharded_overlap_counts
from plade_amd.rendezvous import Rendezvous


def _fake_T(i):
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [i, 2 * i, -i]
    T[0, 0] = np.float32(np.cos(i))
    return T


def _rank(rank, world, key, n_items, q):
    try:
        if rank == world - 1:
            time.sleep(0.3)                      # a late rank: the others wait for it
        comm = Rendezvous(rank, world, key, timeout=60.0)
        comm.barrier()
        parts = comm.all_gather({"rank": rank, "a": np.arange(rank + 1)})
        assert [p["rank"] for p in parts] == list(range(world)) and all(len(p["a"]) == r + 1 for r, p in enumerate(parts))
        mx = comm.all_reduce_max([float(rank), -float(rank)])
        sm = comm.all_reduce_sum([rank, 1])
        assert mx == [float(world - 1), 0.0] and sm == [world * (world - 1) // 2, world]
        mine = shard(n_items, rank, world)
        T = np.stack([_fake_T(i) for i in mine]) if mine else np.zeros((0, 4, 4), np.float32)
        ok = np.array([i % 3 != 0 for i in mine], bool)
        Tg, okg = gather_results(T, ok, n_items, rank, world, comm=comm)
        # candidate sharding with a stand-in counter: the count of candidate i is a function of its transform alone
        K = 11
        cand = np.stack([_fake_T(i) for i in range(K)])
        counts = sharded_overlap_counts(None, None, None, cand, np.zeros((K, 3), np.float32), 1.0, 0.1, rank, world, comm=comm,
                                        counter=lambda s, t, TT, c, r, d: (TT[:, 0, 3] * 7).astype(np.int32))
        assert np.array_equal(counts, np.arange(K, dtype=np.int32) * 7)
        comm.close()
        q.put((rank, "torch" in sys.modules, Tg, okg))
    except Exception as e:   # noqa: BLE001 -- reported to the parent, which fails the test
        q.put((rank, repr(e), None, None))


@pytest.mark.parametrize("world,n_items", [(2, 7), (3, 8), (2, 1), (8, 64), (8, 5)])   # 8: the ranks of one MI355X node; 5 items: ranks without work
def test_ranks_meet_and_gather_without_torch(world, n_items, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"test_{os.getpid()}_{world}_{n_items}"
    ps = [ctx.Process(target=_rank, args=(r, world, key, n_items, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, torch_loaded, Tg, okg = q.get(timeout=120)
        got[r] = (torch_loaded, Tg, okg)
    for p in ps:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][0] is False, got[r][0]     # neither an exception text nor torch in the process
    Tg, okg = got[0][1], got[0][2]
    assert Tg.shape == (n_items, 4, 4)
    for i in range(n_items):
        assert np.array_equal(Tg[i], _fake_T(i)) and okg[i] == (i % 3 != 0)
    for r in range(1, world):
        assert got[r][1] is None and got[r][2] is None


def test_world_of_one_needs_no_peer():
    comm = Rendezvous(0, 1, "unused")
    assert comm.all_gather(5) == [5] and comm.all_reduce_max([2.0]) == [2.0] and comm.gather("x") == ["x"]
    comm.barrier()
    comm.close()


def test_a_stale_port_file_is_ignored(tmp_path):
    """A port file left by a launch that died (nobody listens there, or somebody else does) must not wedge the next one:
    rank 1 keeps retrying until the rank 0 of ITS launch has replaced the file."""
    from plade_amd.rendezvous import _private_dir
    key = f"stale_{os.getpid()}"
    path = os.path.join(_private_dir(), f"plade_rendezvous_{key}")
    with open(path, "w") as f:
        f.write("1 deadbeef\n")                  # port 1: connection refused
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p1 = ctx.Process(target=_rank, args=(1, 2, key, 3, q))
    p1.start()
    time.sleep(0.5)
    p0 = ctx.Process(target=_rank, args=(0, 2, key, 3, q))
    p0.start()
    res = [q.get(timeout=120) for _ in range(2)]
    p0.join(30); p1.join(30)
    assert all(r[1] is False for r in res), res
    assert not os.path.exists(path)


def test_the_port_file_is_private_and_nothing_is_unpickled():
    """Advisor r4: the port file lives in a directory only this user can enter (mode 0700, owner checked, no symlink) with mode
    0600; the handshake is a fixed-format HMAC proof checked before anything else is read; payloads are JSON + raw numeric
    arrays -- the module does not import pickle, refuses object arrays and arbitrary objects, and bounds every length."""
    import socket
    import stat
    import struct
    import threading
    import plade_amd.rendezvous as rz
    src = open(rz.__file__).read()
    assert "import pickle" not in src and "pickle.loads" not in src
    d = rz._private_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0
    # round trip of what the ranks exchange
    a, b = socket.socketpair()
    obj = {"rank": 3, "a": np.arange(5, dtype=np.int32), "l": [1.5, None, "x", [np.float32(2.0)]], "ok": True}
    th = threading.Thread(target=rz._send, args=(a, obj))
    th.start()
    got = rz._recv(b)
    th.join()
    assert got["rank"] == 3 and np.array_equal(got["a"], obj["a"]) and got["l"][:3] == [1.5, None, "x"] and got["ok"] is True
    with pytest.raises(TypeError):
        rz._encode(np.array([object()], dtype=object))
    with pytest.raises(TypeError):
        rz._encode(object())
    # malformed / oversized headers are refused before anything is allocated
    a.sendall(struct.pack("<IIQ", 0x12345678, 0, 4))
    with pytest.raises(ConnectionError):
        rz._recv(b)
    a.sendall(struct.pack("<IIQ", rz._MAGIC, 0, 1 << 40))
    with pytest.raises(ConnectionError):
        rz._recv(b)
    a.close(); b.close()


def _rank_with_a_stranger(rank, world, key, q):
    try:
        comm = Rendezvous(rank, world, key, timeout=60.0)
        comm.barrier()
        comm.close()
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))


def test_a_stranger_without_the_token_is_turned_away():
    """Somebody who finds the port (but cannot read the 0600 port file) and connects with garbage or a wrong proof does not
    become a rank, does not crash rank 0, and does not keep the real rank 1 out."""
    import socket
    import struct
    from plade_amd.rendezvous import _private_dir, _MAGIC
    key = f"stranger_{os.getpid()}"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_rank_with_a_stranger, args=(0, 2, key, q))
    p0.start()
    path = os.path.join(_private_dir(), f"plade_rendezvous_{key}")
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < 30:
        time.sleep(0.02)
    port = int(open(path).read().split()[0])
    for payload in (b"\x80\x04\x95" + b"A" * 200,                                    # a pickle, for a listener that would unpickle
                    struct.pack("<II", _MAGIC, 1) + b"\0" * 32):                      # right format, wrong proof
        s = socket.create_connection(("127.0.0.1", port), timeout=5)
        s.sendall(payload)
        time.sleep(0.1)
        s.close()
    p1 = ctx.Process(target=_rank_with_a_stranger, args=(1, 2, key, q))
    p1.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    p0.join(30); p1.join(30)
    assert res == [(0, "ok"), (1, "ok")], res
