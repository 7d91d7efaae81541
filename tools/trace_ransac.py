import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, plade_amd
from plade_amd.synth import make_pair
for seed in (0, 1):
    tg, sr, Tgt = make_pair(1000000, seed=seed)
    ctx = plade_amd.Context(0, orient_normals=1)
    ct, cs = ctx.upload(tg), ctx.upload(sr)
    ctx.registration_dev(ct, cs)
    ctx.set_params(dump=2)
    print("seed", seed, file=sys.stderr)
    ctx.registration_dev(ct, cs)
    ctx.close()
