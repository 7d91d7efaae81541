"""Multi-GPU batch mode: scan pairs are independent (code/PLADE/main.cpp:97-158 is a plain loop), so
pair i goes to rank i % world, each rank registers its shard on its own GPU with no data-path
collective, and the 4x4 results (+ status) are gathered (one all_gather) and assembled on rank 0 in input
order over torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard(n_items, rank, world):
    """Indices of the items rank `rank` of `world` processes owns (round robin)."""
    return list(range(rank, n_items, world))


def gather_results(local_T, local_ok, n_items, rank, world, device=None):
    """local_T: (len(shard), 4, 4) float32, local_ok: (len(shard),) bool.  Returns on rank 0
    (T (n_items,4,4), ok (n_items,)) in input order, on other ranks (None, None)."""
    import torch
    import torch.distributed as dist
    mine = shard(n_items, rank, world)
    assert len(mine) == len(local_T) == len(local_ok)
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
    T = np.zeros((n_items, 4, 4), np.float32)
    ok = np.zeros(n_items, bool)
    for r in range(world):
        a = parts[r].cpu().numpy()
        for k, i in enumerate(shard(n_items, r, world)):
            T[i] = a[k, :16].reshape(4, 4)
            ok[i] = a[k, 16] > 0.5
    return T, ok


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
