import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair
for seed in (0, 1):
    tg, sr, _ = make_pair(1000000, seed=seed)
    ctx = plade_amd.Context(0, orient_normals=1, dump=2)
    ct, cs = ctx.upload(tg), ctx.upload(sr)
    for it in range(3):
        ok, T = ctx.registration_dev(ct, cs)
    st = ctx.stats()
    print(seed, ok, {k: round(v * 1e6, 1) for k, v in st.items() if k.startswith(("k_overlap_s", "k_pen_walk_s"))}, st.get("n_candidates_verified"))
    ctx.close()
