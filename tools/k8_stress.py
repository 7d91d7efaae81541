"""K8 (candidate verification, plade.cpp:545-564 / util.h:611-647) at the size BASELINE configs[4] names: K = 10^4 candidate
transforms on the voxel-downsampled clouds of a 10M-point pair (n_ds ~ 6e5), through the seam plade_overlap_counts.
SURVEY.md 8d: B_verify = K n_s 12 B + n_t 12 B = 72 GB, T_verify = 27 K n_s cell probes.

    python tools/k8_stress.py [points=10000000] [K=10000] [oracle_samples=64]

Prints one JSON line: sizes, seconds of the whole seam call (upload + grid + kernels), and -- with the kernel's duration
from rocprofv3 (tools/prof_k8.sh) -- the figures DESIGN.md quotes.  A sample of candidates is checked against the oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair, CONFIG4


def stress_inputs(points, K, seed=0):
    tg, sr, Tgt = make_pair(points, seed=seed, **CONFIG4)
    ctx = plade_amd.Context(0)
    s = ctx.average_spacing(sr)
    leaf = np.float32(4) * s
    tds, sds = ctx.voxel_downsample(tg, leaf), ctx.voxel_downsample(sr, leaf)
    ctx.close()
    rng = np.random.default_rng(11)
    # candidates: the true transform, perturbations of it that still overlap (what the clusters around the truth look
    # like), and unrelated poses
    T = np.tile(Tgt.astype(np.float64), (K, 1, 1))
    for k in range(1, K):
        ang = rng.normal(0, 0.02 if k % 3 else 0.4, 3)
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        dR = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        D = np.eye(4)
        D[:3, :3] = dR
        D[:3, 3] = rng.normal(0, 0.1 if k % 3 else 3.0, 3)
        T[k] = D @ Tgt
    T = T.astype(np.float32)
    c_s = ((sds.min(0) + sds.max(0)) / 2).astype(np.float32)
    centers = (np.einsum("kij,j->ki", T[:, :3, :3], c_s) + T[:, :3, 3]).astype(np.float32)   # plade.cpp:555
    radius = np.float32(np.max(sds.max(0) - sds.min(0)) / 2)
    return tds, sds, T, centers, radius, leaf


if __name__ == "__main__":
    points = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n_check = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    t0 = time.perf_counter()
    tds, sds, T, centers, radius, leaf = stress_inputs(points, K)
    t_gen = time.perf_counter() - t0
    ctx = plade_amd.Context(0)
    ctx.overlap_counts(sds, tds, T[:8], centers[:8], radius, leaf)      # first-use allocations
    t0 = time.perf_counter()
    counts = ctx.overlap_counts(sds, tds, T, centers, radius, leaf)
    dt = time.perf_counter() - t0
    out = {"points_per_cloud": points, "K": K, "n_s_ds": len(sds), "n_t_ds": len(tds), "leaf": float(leaf), "src_radius": float(radius),
           "seam_call_seconds": dt, "generation_seconds": t_gen,
           "B_verify_GB": (K * len(sds) * 12 + len(tds) * 12) / 1e9, "T_verify_cell_probes": 27.0 * K * len(sds),
           "counts_min_max_mean": [int(counts.min()), int(counts.max()), float(counts.mean())],
           "candidates_with_overlap_over_half": int((counts > 0.5 * min(len(sds), len(tds))).sum())}
    if n_check:
        from oracle.oracle import Oracle
        orc = Oracle()
        ids = np.unique(np.concatenate([[0, 1, 2], np.linspace(0, K - 1, n_check).astype(int)]))
        bad = 0
        t0 = time.perf_counter()
        for i in ids:
            want = orc.overlap_count(sds, tds, T[i], centers[i], radius, leaf)
            bad += int(want != counts[i])
        out.update({"oracle_checked": len(ids), "oracle_mismatches": bad, "oracle_seconds": time.perf_counter() - t0})
    print(json.dumps(out))
    ctx.close()
