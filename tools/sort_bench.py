"""Per-kernel timing of the device-wide radix sort (run under rocprofv3 --kernel-trace by tools/sort_bench.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
ctx = plade_amd.Context(0)
rng = np.random.default_rng(0)
for n in (60000, 250000, 1000000, 4000000):
    for dt, bits in ((np.uint32, 24), (np.uint64, 30)):
        k = rng.integers(0, (1 << bits) - 1, n, dtype=np.uint64).astype(dt)
        v = np.arange(n, dtype=np.uint32)
        for rep in range(3):
            ctx.sort_pairs(k, v, bits)
