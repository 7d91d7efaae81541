"""BASELINE configs[4]-style run: one large pair (default 10M points, ~100 planes, up to 1e4 candidates)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair, CONFIG4

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
boxes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t0 = time.perf_counter()
# a 32 x 28 x 12 m hall with `boxes` pieces of furniture in general position (no two faces parallel): 6 + 3 x 32 = 102 planes
tg, sr, Tgt = make_pair(n, seed=0, **CONFIG4)
print(f"generated {len(tg)} + {len(sr)} points in {time.perf_counter() - t0:.1f} s", flush=True)
ctx = plade_amd.Context(0, orient_normals=1, max_planes=100, max_candidates=10000, init_min_support=int(os.environ.get("MIN_SUPPORT", "10000")))
ct, cs = ctx.upload(tg), ctx.upload(sr)
for it in range(3):
    t0 = time.perf_counter()
    try:
        ok, T = ctx.registration_dev(ct, cs)
        dt = time.perf_counter() - t0
        print(f"iter {it} ok {ok} sec {dt:.4f} frob {np.linalg.norm(T - Tgt):.3e}", flush=True)
    except plade_amd.PladeError as e:
        print("error:", e, flush=True)
        break
for k, v in ctx.stats().items():
    if k.startswith(("t_", "n_", "ransac_", "pen_")):
        print(f"  {k:28s} {v:.6g}")
