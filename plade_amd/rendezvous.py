"""Exchange between the ranks of ONE node without torch: the registration path shards whole pairs across ranks and has no
data-path collective (code/PLADE/main.cpp:97-158 is a plain loop over independent pairs); what the ranks do exchange is a
barrier around the timed region, a few numbers to reduce, and 68 bytes of result per pair for rank 0 to write out.  `import
torch` costs the registration ~5 % (it brings the HIP runtime bundled with the wheel into the process ahead of the
system's, bench.py), so the ranks meet on a TCP socket of the loopback interface instead: rank 0 listens on an ephemeral
port and publishes it, with a random token, in a file named after MASTER_PORT; ranks 1.. connect and prove they know the
token; every operation is a gather to rank 0 followed by a broadcast of the result.  It also bootstraps the RCCL
communicator of plade_amd/rccl_comm.py (the 128-byte unique id travels over it) and is the fallback where RCCL cannot
serve (two ranks on one device, no librccl).

    comm = Rendezvous.from_env()           # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them
    comm.barrier(); parts = comm.all_gather(obj); comm.close()

Security (advisor r4): nothing a peer sends is ever unpickled.
  * The port file lives in a PER-USER directory with mode 0700 whose owner and mode are verified (XDG_RUNTIME_DIR, else
    <tmp>/plade-<uid> created by us -- a pre-created foreign directory or a symlink is refused), is written with
    O_CREAT | O_EXCL, mode 0600, and moved into place inside that directory, so no other local user can plant, read or
    replace it.
  * The handshake is fixed-format: 4 bytes magic, 4 bytes rank, 32 bytes HMAC-SHA256(token, "hello" | rank), compared with
    hmac.compare_digest before anything else is read; rank 0 answers with HMAC(token, "welcome" | rank), which the client
    checks -- whoever listens on a stale port cannot impersonate rank 0.
  * Messages are a JSON skeleton (None, bool, int, float, str, list, dict with string keys) plus the raw bytes of numpy
    arrays of plain numeric dtypes; every length field is bounded before anything is allocated.
"""
import hashlib
import hmac
import json
import os
import socket
import stat
import struct
import tempfile
import time

import numpy as np

_MAGIC = 0x504C4452                      # "PLDR"
_MAX_JSON = 16 << 20
_MAX_BLOB = 256 << 20
_MAX_BLOBS = 4096
_DTYPES = {"bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float16", "float32", "float64"}


def _exactly(sock, n):
    parts = []
    while n:
        c = sock.recv(min(n, 1 << 20))
        if not c:
            raise ConnectionError("rendezvous: a peer closed its connection")
        parts.append(c)
        n -= len(c)
    return b"".join(parts)


def _encode(obj):
    blobs = []

    def skel(o):
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        if isinstance(o, (np.bool_, np.integer)):
            return int(o)
        if isinstance(o, np.floating):
            return float(o)
        if isinstance(o, np.ndarray):
            if o.dtype.name not in _DTYPES:
                raise TypeError(f"rendezvous: arrays of dtype {o.dtype} are not exchanged")
            a = np.ascontiguousarray(o)
            blobs.append(a.tobytes())
            return {"__nd__": len(blobs) - 1, "dtype": a.dtype.name, "shape": list(a.shape)}
        if isinstance(o, (list, tuple)):
            return [skel(x) for x in o]
        if isinstance(o, dict):
            if not all(isinstance(k, str) for k in o):
                raise TypeError("rendezvous: dict keys must be strings")
            return {"__map__": {k: skel(v) for k, v in o.items()}}
        raise TypeError(f"rendezvous: objects of type {type(o).__name__} are not exchanged (numbers, strings, lists, dicts, numpy arrays)")
    js = json.dumps(skel(obj), allow_nan=True).encode()
    out = [struct.pack("<IIQ", _MAGIC, len(blobs), len(js)), js]
    for b in blobs:
        out.append(struct.pack("<Q", len(b)))
        out.append(b)
    return b"".join(out)


def _send(sock, obj):
    sock.sendall(_encode(obj))


def _recv(sock):
    magic, n_blobs, n_js = struct.unpack("<IIQ", _exactly(sock, 16))
    if magic != _MAGIC or n_blobs > _MAX_BLOBS or n_js > _MAX_JSON:
        raise ConnectionError("rendezvous: malformed message header")
    sk = json.loads(_exactly(sock, n_js).decode())
    blobs, total = [], 0
    for _ in range(n_blobs):
        (n,) = struct.unpack("<Q", _exactly(sock, 8))
        total += n
        if total > _MAX_BLOB:
            raise ConnectionError("rendezvous: message too large")
        blobs.append(_exactly(sock, n))

    def build(o):
        if isinstance(o, list):
            return [build(x) for x in o]
        if isinstance(o, dict):
            if "__map__" in o:
                return {k: build(v) for k, v in o["__map__"].items()}
            if "__nd__" in o:
                if o["dtype"] not in _DTYPES:
                    raise ConnectionError("rendezvous: array dtype not allowed")
                dt = np.dtype(o["dtype"])
                shape = tuple(int(x) for x in o["shape"])
                b = blobs[int(o["__nd__"])]
                if int(np.prod(shape, dtype=np.int64)) * dt.itemsize != len(b):
                    raise ConnectionError("rendezvous: array size does not match its header")
                return np.frombuffer(b, dt).reshape(shape).copy()
            raise ConnectionError("rendezvous: malformed message")
        return o
    return build(sk)


def _private_dir(directory=None):
    """A directory only this user can enter: XDG_RUNTIME_DIR when it is one, else <tmp>/plade-<uid> (created 0700).  Owner and
    mode are checked on the directory itself (lstat: a symlink planted by somebody else is refused)."""
    uid = os.getuid()
    cands = []
    if directory:
        cands.append(directory)
    else:
        x = os.environ.get("XDG_RUNTIME_DIR")
        if x:
            cands.append(x)
        cands.append(os.path.join(tempfile.gettempdir(), f"plade-{uid}"))
    for d in cands:
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        except OSError:
            continue
        try:
            st = os.lstat(d)
        except OSError:
            continue
        if stat.S_ISDIR(st.st_mode) and st.st_uid == uid and (st.st_mode & 0o077) == 0:
            return d
        if directory:       # a caller-chosen directory (the tests' tmp_path) need not be 0700, but must be ours and no symlink
            if stat.S_ISDIR(st.st_mode) and st.st_uid == uid:
                return d
    raise PermissionError("rendezvous: no private directory for the port file (XDG_RUNTIME_DIR or <tmp>/plade-<uid> must be a "
                          "directory owned by this user with mode 0700)")


def _mac(token, what, rank):
    return hmac.new(token, what + struct.pack("<I", rank), hashlib.sha256).digest()


class Rendezvous:
    def __init__(self, rank, world, key, addr="127.0.0.1", timeout=300.0, directory=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: the sockets of ranks 1.., in rank order
        self.sock = None         # ranks 1..: the connection to rank 0
        self._listener = None
        self._path = None
        if self.world <= 1:
            return
        safe_key = "".join(c if c.isalnum() or c in "-_." else "_" for c in str(key))
        self._path = os.path.join(_private_dir(directory), f"plade_rendezvous_{safe_key}")
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            ls.bind((addr, 0))
            ls.listen(self.world)
            self._listener = ls
            token = os.urandom(32)
            tmp = self._path + f".{os.getpid()}.{os.urandom(4).hex()}"
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as f:
                f.write(f"{ls.getsockname()[1]} {token.hex()}\n")
            os.replace(tmp, self._path)          # atomically, inside our own directory: a reader sees the whole line or the previous file
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
                c.settimeout(10.0)
                try:
                    magic, r = struct.unpack("<II", _exactly(c, 8))
                    proof = _exactly(c, 32)
                    ok = (magic == _MAGIC and 0 < r < self.world and r not in got
                          and hmac.compare_digest(proof, _mac(token, b"hello", r)))
                except Exception:                # noqa: BLE001 -- not one of ours: whatever it sent, it is turned away
                    ok = False
                if not ok:
                    c.close()                    # a straggler of another launch that read a stale file, or a stranger
                    continue
                c.settimeout(timeout)
                got[r] = c
            self.peers = [got[r] for r in range(1, self.world)]
            for r, c in enumerate(self.peers, start=1):
                c.sendall(_mac(token, b"welcome", r))
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank 0 did not appear")
                s = None
                try:
                    st = os.lstat(self._path)
                    if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
                        raise PermissionError("rendezvous: the port file is not ours")
                    port, token_hex = open(self._path).read().split()
                    token = bytes.fromhex(token_hex)
                    s = socket.create_connection((addr, int(port)), timeout=2.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    s.sendall(struct.pack("<II", _MAGIC, self.rank) + _mac(token, b"hello", self.rank))
                    if hmac.compare_digest(_exactly(s, 32), _mac(token, b"welcome", self.rank)):
                        self.sock = s
                        break
                    s.close()
                except Exception:                 # noqa: BLE001 -- no file yet, a stale one (nobody, or somebody else, listens
                    if s is not None:             # there), or rank 0 not listening yet: read the file again
                        s.close()
                time.sleep(0.02)

    @classmethod
    def from_env(cls, timeout=300.0):
        """The ranks `python -m torch.distributed.run` (or anything that sets the same variables) started on this host."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        # MASTER_PORT identifies the launch among those running on this host at the same time (two cannot share it); a file
        # left by an earlier launch on the same port is harmless: nobody listens there (or the proof fails) and the
        # ranks keep reading until the new rank 0 has replaced it
        key = os.environ.get("PLADE_RENDEZVOUS_KEY") or f"{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        return cls(rank, world, key, os.environ.get("PLADE_RENDEZVOUS_ADDR", "127.0.0.1"), timeout,
                   os.environ.get("PLADE_RENDEZVOUS_DIR"))

    def all_gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            parts = [obj] + [_recv(c) for c in self.peers]
            msg = _encode(parts)
            for c in self.peers:
                c.sendall(msg)
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

    def broadcast(self, obj):
        """rank 0's obj on every rank."""
        return self.all_gather(obj if self.rank == 0 else None)[0]

    def barrier(self):
        self.all_gather(None)

    def all_reduce_max(self, values):
        parts = self.all_gather([float(v) for v in values])
        return [max(p[i] for p in parts) for i in range(len(values))]

    def all_reduce_sum(self, values):
        parts = self.all_gather([v.item() if hasattr(v, "item") else v for v in values])
        return [sum(p[i] for p in parts) for i in range(len(values))]

    def close(self):
        if self.world > 1 and (self.peers or self.sock):
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
