"""Exchange between the ranks of ONE node without torch: the registration path shards whole pairs across ranks and has no
data-path collective (code/PLADE/main.cpp:97-158 is a plain loop over independent pairs); what the ranks do exchange is a
barrier around the timed region, a few numbers to reduce, and 68 bytes of result per pair for rank 0 to write out.  That
does not need a collective library in the process -- and `import torch` costs the registration ~5 % (it brings the HIP
runtime bundled with the wheel into the process ahead of the system's, bench.py) -- so the ranks meet on a TCP socket of
the loopback interface instead: rank 0 listens on an ephemeral port and publishes it (with a random token) in a file named
after MASTER_PORT; ranks 1.. connect and present the token; every operation is a gather to rank 0 followed by
a broadcast of the result.

    comm = Rendezvous.from_env()           # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them
    comm.barrier(); parts = comm.all_gather(obj); comm.close()

Objects are pickled; the peers are the ranks of one launch on one host, not a network service."""
import os
import pickle
import socket
import struct
import tempfile
import time


def _send(sock, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(b)) + b)


def _recv(sock):
    def exactly(n):
        parts = []
        while n:
            c = sock.recv(min(n, 1 << 20))
            if not c:
                raise ConnectionError("rendezvous: a peer closed its connection")
            parts.append(c)
            n -= len(c)
        return b"".join(parts)
    (n,) = struct.unpack("<Q", exactly(8))
    return pickle.loads(exactly(n))


class Rendezvous:
    def __init__(self, rank, world, key, addr="127.0.0.1", timeout=300.0, directory=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: the sockets of ranks 1.., in rank order
        self.sock = None         # ranks 1..: the connection to rank 0
        self._listener = None
        self._path = os.path.join(directory or tempfile.gettempdir(), f"plade_rendezvous_{key}")
        if self.world <= 1:
            return
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind((addr, 0))
            ls.listen(self.world)
            self._listener = ls
            token = os.urandom(8).hex()
            tmp = self._path + f".{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(f"{ls.getsockname()[1]} {token}\n")
            os.replace(tmp, self._path)          # atomically: a reader sees the whole line or the previous file
            got = {}
            ls.settimeout(1.0)
            while len(got) < self.world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: {self.world - 1 - len(got)} rank(s) did not arrive")
                try:
                    c, _ = ls.accept()
                except socket.timeout:
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                try:
                    hello = _recv(c)
                    ok = hello.get("token") == token and hello.get("rank") not in got and 0 < hello.get("rank", 0) < self.world
                except Exception:                # noqa: BLE001 -- not one of ours: whatever it sent, it is turned away
                    ok = False
                if not ok:
                    c.close()                    # a straggler of another launch that read a stale file
                    continue
                got[hello["rank"]] = c
            self.peers = [got[r] for r in range(1, self.world)]
            for c in self.peers:
                _send(c, "welcome")
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank 0 did not appear")
                try:
                    port, token = open(self._path).read().split()
                    s = socket.create_connection((addr, int(port)), timeout=2.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    _send(s, {"rank": self.rank, "token": token})
                    if _recv(s) == "welcome":
                        self.sock = s
                        break
                    s.close()
                except Exception:                 # noqa: BLE001 -- no file yet, a stale one (nobody, or somebody else, listens
                    pass                          # there), or rank 0 not listening yet: read the file again
                time.sleep(0.02)

    @classmethod
    def from_env(cls, timeout=300.0):
        """The ranks `python -m torch.distributed.run` (or anything that sets the same variables) started on this host."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        # MASTER_PORT identifies the launch among those running on this host at the same time (two cannot share it); a file
        # left by an earlier launch on the same port is harmless: nobody listens there (or its token is refused) and the
        # ranks keep reading until the new rank 0 has replaced it
        key = os.environ.get("PLADE_RENDEZVOUS_KEY") or f"{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        return cls(rank, world, key, os.environ.get("PLADE_RENDEZVOUS_ADDR", "127.0.0.1"), timeout)

    def all_gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            parts = [obj] + [_recv(c) for c in self.peers]
            for c in self.peers:
                _send(c, parts)
            return parts
        _send(self.sock, obj)
        return _recv(self.sock)

    def gather(self, obj):
        """The list of all ranks' objects on rank 0, None elsewhere (the other ranks do not wait for rank 0)."""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(c) for c in self.peers]
        _send(self.sock, obj)
        return None

    def barrier(self):
        self.all_gather(None)

    def all_reduce_max(self, values):
        parts = self.all_gather([float(v) for v in values])
        return [max(p[i] for p in parts) for i in range(len(values))]

    def all_reduce_sum(self, values):
        parts = self.all_gather(list(values))
        return [sum(p[i] for p in parts) for i in range(len(values))]

    def close(self):
        if self.world > 1:
            try:
                self.barrier()                    # nobody leaves while a peer still reads
            except (OSError, ConnectionError):
                pass
        for c in self.peers:
            c.close()
        if self.sock:
            self.sock.close()
        if self._listener:
            self._listener.close()
            try:
                os.remove(self._path)
            except OSError:
                pass
        self.peers, self.sock, self._listener = [], None, None
