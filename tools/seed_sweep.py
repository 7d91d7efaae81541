"""Registers synthetic pairs of many seeds and prints the error against the ground truth (development helper: a change of
the extraction's search order must not turn a correct registration into a symmetric alignment of the room)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, plade_amd
from plade_amd.synth import make_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
seeds = range(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
ctx = plade_amd.Context(0, orient_normals=1)
bad = 0
for s in seeds:
    tg, sr, Tgt = make_pair(n, seed=s)
    ok, T = ctx.registration(tg, sr)
    e = float(np.linalg.norm(T - Tgt))
    st = ctx.stats()
    print(f"seed {s:2d} ok {ok} err {e:.5f} planes {int(st['n_planes_tgt'])}+{int(st['n_planes_src'])} iterations {int(st.get('ransac_iterations', 0))}", flush=True)
    bad += (not ok) or e > 0.1
print("bad", bad)
