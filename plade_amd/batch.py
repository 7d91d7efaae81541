"""Multi-GPU batch mode: scan pairs are independent (code/PLADE/main.cpp:97-158 is a plain loop), so
pair i goes to rank i % world, each rank registers its shard on its own GPU with no data-path
collective, and the 4x4 results (+ status) are gathered and assembled on rank 0 in input order -- 68 bytes per pair, over
the ranks' loopback rendezvous (`comm`, plade_amd/rendezvous.py: no torch in the process) or, for callers that live in a
torch.distributed job anyway, with one all_gather (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard(n_items, rank, world):
    """Indices of the items rank `rank` of `world` processes owns (round robin)."""
    return list(range(rank, n_items, world))


def _assemble(parts, n_items, world):
    T = np.zeros((n_items, 4, 4), np.float32)
    ok = np.zeros(n_items, bool)
    for r in range(world):
        a = parts[r]
        for k, i in enumerate(shard(n_items, r, world)):
            T[i] = a[k, :16].reshape(4, 4)
            ok[i] = a[k, 16] > 0.5
    return T, ok


def gather_results(local_T, local_ok, n_items, rank, world, device=None, comm=None):
    """local_T: (len(shard), 4, 4) float32, local_ok: (len(shard),) bool.  Returns on rank 0
    (T (n_items,4,4), ok (n_items,)) in input order, on other ranks (None, None).  `comm`: a plade_amd.rendezvous.Rendezvous
    of the ranks (then nothing here touches torch); without it the exchange is torch.distributed's."""
    mine = shard(n_items, rank, world)
    assert len(mine) == len(local_T) == len(local_ok)
    if world == 1:   # nothing to exchange (and no torch in a single-GPU process: see bench.py)
        return np.asarray(local_T, np.float32).reshape(n_items, 4, 4).copy(), np.asarray(local_ok, bool).copy()
    if comm is not None and hasattr(comm, "all_gather_array"):
        # RCCL (plade_amd/rccl_comm.py): ONE ncclAllGather of equally sized blocks, 68 bytes per pair, shards padded to the same length
        per = (n_items + world - 1) // world
        buf = np.zeros((per, 17), np.float32)
        if len(mine):
            buf[: len(mine), :16] = np.asarray(local_T, np.float32).reshape(len(mine), 16)
            buf[: len(mine), 16] = np.asarray(local_ok, np.float32)
        parts = comm.all_gather_array(buf)
        return _assemble(parts, n_items, world) if rank == 0 else (None, None)
    if comm is not None:
        buf = np.zeros((len(mine), 17), np.float32)
        if len(mine):
            buf[:, :16] = np.asarray(local_T, np.float32).reshape(len(mine), 16)
            buf[:, 16] = np.asarray(local_ok, np.float32)
        parts = comm.gather(buf)
        return _assemble(parts, n_items, world) if rank == 0 else (None, None)
    import torch
    import torch.distributed as dist
    per = (n_items + world - 1) // world  # pad every shard to the same length for the collective
    buf = np.zeros((per, 17), np.float32)
    if len(mine):
        buf[: len(mine), :16] = np.asarray(local_T, np.float32).reshape(len(mine), 16)
        buf[: len(mine), 16] = np.asarray(local_ok, np.float32)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    if world == 1:
        parts = [t]
    else:
        # all_gather: the one collective every backend implements natively (RCCL ring over xGMI); 68 bytes
        # per pair, so the extra copies on ranks > 0 cost nothing
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
    if rank != 0:
        return None, None
    return _assemble([p.cpu().numpy() for p in parts], n_items, world)


def sharded_overlap_counts(ctx, src_ds, tgt_ds, T, centers, src_radius, inlier_dist, rank, world, device=None, counter=None, comm=None):
    """Second sharding axis: the K candidate transforms of ONE pair are independent in the verification
    step (code/PLADE/plade.cpp:547-564), so rank r scores candidates r, r + world, ... on its own GPU against its
    own copy of the two downsampled clouds (seam S3, plade_overlap_counts) and the K int32 counts are all-gathered
    (4 B per candidate; the clouds are <= 15 MB and are handed to every rank by the caller).  Returns the K counts in
    candidate order on every rank.  `counter(src, tgt, T, centers, radius, dist)` defaults to ctx.overlap_counts (the
    CPU tests pass a stand-in).  The default pipeline does not use this: with the reference's K <= 201 the
    verification kernel takes ~0.2 ms of a 7 ms registration; it pays when the candidate cap is lifted."""
    T = np.ascontiguousarray(T, np.float32).reshape(-1, 4, 4)
    centers = np.ascontiguousarray(centers, np.float32).reshape(-1, 3)
    K = len(T)
    mine = shard(K, rank, world)
    counter = counter or ctx.overlap_counts
    local = (np.asarray(counter(src_ds, tgt_ds, T[mine], centers[mine], src_radius, inlier_dist), np.int32)
             if mine else np.zeros(0, np.int32))
    if world == 1:
        return local
    if comm is not None and hasattr(comm, "all_gather_array"):
        per = (K + world - 1) // world
        buf = np.full(per, -2, np.int32)
        buf[: len(mine)] = local
        out = np.zeros(K, np.int32)
        for r, a in enumerate(comm.all_gather_array(buf)):
            idx = shard(K, r, world)
            out[idx] = a[: len(idx)]
        return out
    if comm is not None:
        out = np.zeros(K, np.int32)
        for r, a in enumerate(comm.all_gather(local)):
            out[shard(K, r, world)] = a
        return out
    import torch
    import torch.distributed as dist
    per = (K + world - 1) // world
    buf = np.full(per, -2, np.int32)
    buf[: len(mine)] = local
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = np.zeros(K, np.int32)
    for r in range(world):
        a = parts[r].cpu().numpy()
        idx = shard(K, r, world)
        out[idx] = a[: len(idx)]
    return out


def write_result_file(path, pairs, T, ok):
    """The batch result grammar of code/PLADE/main.cpp:134-146 (matrix in Eigen's default format is
    produced by the C++ CLI; this python writer is used by tools/tests with %g formatting)."""
    with open(path, "w") as f:
        for (tg, sr), m, s in zip(pairs, T, ok):
            f.write(f"target: {tg}\nsource: {sr}\n")
            f.write("transformation:\n" if s else "registration failed, an identity matrix is recorded:\n")
            mm = m if s else np.eye(4, dtype=np.float32)
            cells = [[f"{v:.6g}" for v in row] for row in mm]
            w = max(len(c) for row in cells for c in row)
            f.write("\n".join(" ".join(c.rjust(w) for c in row) for row in cells) + "\n\n")
