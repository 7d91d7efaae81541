"""Per-iteration trace of the extraction (PLADE_TRACE_RANSAC, profiled mode: one iteration at a time) for a few bench pairs:
how many trailing iterations only draw hypotheses until the stopping rule holds?
    PLADE_TRACE_RANSAC=1 python tools/ransac_trace.py [pairs] 2> trace.txt
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLADE_TRACE_RANSAC"] = "1"
import plade_amd
from plade_amd.synth import make_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = plade_amd.Context(0, orient_normals=1, dump=2)
for s in range(N):
    tg, sr, _ = make_pair(1000000, seed=s)
    sys.stderr.write(f"[pair] seed {s}\n")
    ctx.registration_dev(ctx.upload(tg), ctx.upload(sr))
