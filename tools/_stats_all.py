import sys, json
sys.path.insert(0, "/root/repo")
import plade_amd
from plade_amd.synth import make_pair
ctx = plade_amd.Context(0, orient_normals=1)
for s in (0, 3):
    tg, sr, _ = make_pair(1000000, seed=s)
    r = ctx.registration_dev(ctx.upload(tg), ctx.upload(sr))
    print(json.dumps(ctx.stats(), indent=0))
