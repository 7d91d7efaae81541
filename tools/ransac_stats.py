"""Where the acceptance chains of the bench pairs stop: sums of the RANSAC counters (plade_stats_get) over the first N seeds
of the batch, every pair registered alone.
    python tools/ransac_stats.py [pairs] [points]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = plade_amd.Context(0, orient_normals=1)
tot = {}
for s in range(N):
    tg, sr, _ = make_pair(n, seed=s)
    r = ctx.registration_dev(ctx.upload(tg), ctx.upload(sr))
    st = ctx.stats()
    for k, v in st.items():
        if k.startswith("ransac"):
            tot[k] = tot.get(k, 0.0) + v
print(json.dumps({"pairs": N, "points": n, "sums": tot}, indent=1))
