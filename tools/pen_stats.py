import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair
for seed in (0, 1):
    tg, sr, _ = make_pair(1000000, seed=seed)
    ctx = plade_amd.Context(0, orient_normals=1, dump=1)
    ok, T = ctx.registration(tg, sr)
    st = ctx.stats()
    d = ctx.dump()
    print(seed, ok, {k: v for k, v in st.items() if k.startswith(("pen_", "n_"))}, "planes", len(d["tgt_planes"]) // 4, len(d["src_planes"]) // 4)
    ctx.close()
