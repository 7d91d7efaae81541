import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair
mode = sys.argv[3] if len(sys.argv) > 3 else "host"
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    tg, sr, Tgt = make_pair(1000000, seed=seed)
    ctx = plade_amd.Context(0, orient_normals=1)
    ct, cs = (ctx.upload(tg), ctx.upload(sr)) if mode == "dev" else (None, None)
    for rep in range(3):
        ok, T = ctx.registration_dev(ct, cs) if mode == "dev" else ctx.registration(tg, sr)
        st = ctx.stats()
        print(f"seed {seed} rep {rep} mode {mode} diverse {os.environ.get('PLADE_RANSAC_DIVERSE','unset')} ok {ok} err {np.linalg.norm(T-Tgt):.4f} planes {int(st['n_planes_tgt'])}+{int(st['n_planes_src'])} iters {int(st['ransac_iterations'])} rounds {int(st['ransac_rounds'])} batches {int(st['ransac_batches'])} verified {int(st['n_candidates_verified'])}", flush=True)
    ctx.close()
