"""RANSAC loop statistics of one registration of the bench pairs (which refit slot the chains stop at, etc.)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair
for seed in (0, 1):
    tg, sr, _ = make_pair(1000000, seed=seed)
    ctx = plade_amd.Context(0, orient_normals=1)
    ok, T = ctx.registration(tg, sr)
    st = ctx.stats()
    print(seed, ok, {k: v for k, v in st.items() if k.startswith("ransac_") and not k.startswith("ransac_t")})
    ctx.close()
