"""k_overlap at the bench's shape through the seam plade_overlap_counts: K candidates near the true transform on the
downsampled clouds of a 1M-point pair.  Run under rocprofv3 --kernel-trace --stats to read the kernel's duration per K."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

tg, sr, Tgt = make_pair(1000000, seed=0)
ctx = plade_amd.Context(0)
s = ctx.average_spacing(sr)
leaf = np.float32(4) * s
tds, sds = ctx.voxel_downsample(tg, leaf), ctx.voxel_downsample(sr, leaf)
rng = np.random.default_rng(3)
c_s = ((sds.min(0) + sds.max(0)) / 2).astype(np.float32)
radius = np.float32(np.max(sds.max(0) - sds.min(0)) / 2)
print("n_s", len(sds), "n_t", len(tds), "leaf", leaf)
for K in [int(v) for v in (sys.argv[1:] or ["2", "18", "64"])]:
    T = np.tile(Tgt.astype(np.float64), (K, 1, 1))
    for k in range(1, K):
        T[k, :3, 3] += rng.normal(0, 0.05, 3)
    T = T.astype(np.float32)
    centers = (np.einsum("kij,j->ki", T[:, :3, :3], c_s) + T[:, :3, 3]).astype(np.float32)
    ctx.overlap_counts(sds, tds, T, centers, radius, leaf)
    t0 = time.perf_counter()
    for _ in range(5):
        cnt = ctx.overlap_counts(sds, tds, T, centers, radius, leaf)
    print("K", K, "seam call ms", (time.perf_counter() - t0) / 5 * 1e3, "counts", cnt[:3])
