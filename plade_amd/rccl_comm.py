"""The ranks' exchange over RCCL / xGMI without torch in the process (include/plade_hip.h: plade_comm_*; the library opens
librccl with dlopen).  One process per GPU; scan pairs are independent (code/PLADE/main.cpp:122-148 is a plain loop), so the
only exchange steps of a multi-GPU run are a barrier around the timed region, a few numbers to reduce, 68 bytes of result
per pair once per batch, and -- on the candidate axis of one pair (code/PLADE/plade.cpp:547-564) -- 8 bytes per candidate per
registration.  Every one of them is ONE ncclAllGather of a fixed-size block.

    boot = Rendezvous.from_env()                       # carries the 128-byte unique id, and is the fallback
    comm, why = rccl_comm.connect(rank, world, device, boot)
    if comm is None: comm = boot                        # same interface: barrier / all_reduce_max / all_reduce_sum / gather

RCCL refuses two ranks on one device (the one-GPU test hook of bench.py) and can be absent; `connect` therefore initialises it on
a helper thread with a deadline, runs one all-gather as a self-test, and lets the ranks AGREE over the bootstrap whether all of
them came up -- if not, every rank falls back to the bootstrap and says so."""
import ctypes as C
import os
import threading

import numpy as np

from . import load_library, PladeError

ID_BYTES = 128


class RcclComm:
    def __init__(self, L, handle, rank, world):
        self.L, self.h, self.rank, self.world = L, handle, int(rank), int(world)

    # -- the one primitive -------------------------------------------------------------------------------------------
    def all_gather_array(self, a):
        """Every rank passes an array of the SAME dtype and shape; returns the list of all ranks' arrays, in rank order."""
        a = np.ascontiguousarray(a)
        out = np.empty((self.world,) + a.shape, a.dtype)
        if a.nbytes:
            rc = self.L.plade_comm_all_gather(self.h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.nbytes)
            if rc != 0:
                raise PladeError(rc, self.L.plade_comm_last_error(self.h).decode(errors="replace"))
        return [out[r] for r in range(self.world)]

    # -- what bench.py / batch.py need (same names as plade_amd.rendezvous.Rendezvous) ----------------------------------
    def barrier(self):
        self.all_gather_array(np.zeros(1, np.int32))

    def all_reduce_max(self, values):
        parts = self.all_gather_array(np.asarray(values, np.float64))
        return [float(max(p[i] for p in parts)) for i in range(len(values))]

    def all_reduce_sum(self, values):
        parts = self.all_gather_array(np.asarray(values, np.float64))
        sums = [float(sum(p[i] for p in parts)) for i in range(len(values))]
        return [int(round(v)) if isinstance(values[i], (int, np.integer)) else v for i, v in enumerate(sums)]

    def close(self):
        if self.h:
            self.L.plade_comm_destroy(self.h)
            self.h = None


def _bind(L):
    p = C.c_void_p
    L.plade_comm_unique_id.argtypes = [p]
    L.plade_comm_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, p, C.POINTER(p)]
    L.plade_comm_all_gather.argtypes = [p, p, p, C.c_uint64]
    L.plade_comm_destroy.argtypes = [p]
    L.plade_comm_last_error.argtypes = [p]
    L.plade_comm_last_error.restype = C.c_char_p
    L.plade_set_candidate_shard_comm.argtypes = [p, p, C.c_uint32]


def connect(rank, world, device, bootstrap, timeout=None):
    """(RcclComm, "rccl") when every rank of the job came up on RCCL, else (None, reason).  `bootstrap`: a Rendezvous of the
    same ranks (it carries the unique id and the agreement; world 1 needs none)."""
    timeout = float(timeout if timeout is not None else os.environ.get("PLADE_RCCL_TIMEOUT", "90"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
    L = load_library()
    _bind(L)
    ident = np.zeros(ID_BYTES, np.uint8)
    reason = ""
    if rank == 0:
        rc = L.plade_comm_unique_id(ident.ctypes.data_as(C.c_void_p))
        if rc != 0:
            reason = "rank 0: " + L.plade_comm_last_error(None).decode(errors="replace")
    if world > 1:
        got = bootstrap.broadcast({"id": ident, "reason": reason})
        ident, reason = np.ascontiguousarray(got["id"], np.uint8), got["reason"]
    state = {"comm": None, "err": reason}

    def init():
        if state["err"]:
            return
        h = C.c_void_p()
        rc = L.plade_comm_create(int(device), int(rank), int(world), ident.ctypes.data_as(C.c_void_p), C.byref(h))
        if rc != 0:
            state["err"] = L.plade_comm_last_error(None).decode(errors="replace") or f"plade_comm_create: {rc}"
            return
        c = RcclComm(L, h, rank, world)
        try:
            parts = c.all_gather_array(np.array([rank, 7 * rank + 1], np.int64))     # self-test: one collective, checked
            if not all(int(p[0]) == r and int(p[1]) == 7 * r + 1 for r, p in enumerate(parts)):
                state["err"] = "self-test all-gather returned wrong data"
                return
        except PladeError as e:
            state["err"] = str(e)
            return
        state["comm"] = c
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        state["err"] = f"RCCL initialisation did not finish within {timeout:.0f} s"
    mine_ok = state["comm"] is not None and not th.is_alive()
    if world > 1:
        flags = bootstrap.all_gather({"ok": bool(mine_ok), "err": state["err"]})
        all_ok = all(f["ok"] for f in flags)
        if not all_ok:
            why = "; ".join(f"rank {r}: {f['err']}" for r, f in enumerate(flags) if not f["ok"])
            return None, why       # a communicator that came up on some ranks only is abandoned (destroying it could wait for the others)
    elif not mine_ok:
        return None, state["err"]
    return state["comm"], "rccl"


def set_candidate_shard(ctx, comm, min_candidates=0):
    """plade_set_candidate_shard_comm: ctx (a plade_amd.Context) scores candidates k % world == rank and the library all-gathers
    the device-resident counts over `comm` itself.  comm=None switches the axis off."""
    _bind(ctx.L)
    ctx._check(ctx.L.plade_set_candidate_shard_comm(ctx.h, comm.h if comm is not None else None, int(min_candidates)))
