"""Quick timing of the full GPU registration on a synthetic pair (development helper)."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
tg, sr, Tgt = make_pair(n, seed=0)
ctx = plade_amd.Context(0, orient_normals=1)
ct, cs = ctx.upload(tg), ctx.upload(sr)
for it in range(4):
    t = time.time()
    ok, T = ctx.registration_dev(ct, cs)
    dt = time.time() - t
    print("iter", it, "ok", ok, "sec", round(dt, 4), "frob", float(np.linalg.norm(T - Tgt)))
st = ctx.stats()
for k, v in st.items():
    print(f"  {k:28s} {v:.6g}")
