"""One registration at a time on resident clouds, spinning waits: where a single pair's latency goes (t_* stage clocks of the library,
median over the scenes).    python tools/latency_alone.py [scenes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plade_amd
from plade_amd.synth import make_pair
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pairs = [make_pair(1000000, seed=s) for s in range(NP)]
for mode in (1, 0):
    c = plade_amd.Context(0, orient_normals=1, host_wait=0, closest_point_mode=mode)
    cl = [(c.upload(tg), c.upload(sr)) for tg, sr, _ in pairs]
    for ct, cs in cl: c.registration_dev(ct, cs)          # warm
    lat, acc = [], {}
    for rep in range(3):
        for ct, cs in cl:
            t0 = time.perf_counter(); c.registration_dev(ct, cs); lat.append((time.perf_counter() - t0) * 1e3)
            for k, v in c.stats().items():
                if k.startswith("t_") or k in ("ransac_iterations",): acc.setdefault(k, []).append(v)
    lat.sort()
    print(f"closest_point_mode {mode}: latency ms min {lat[0]:.2f} median {lat[len(lat)//2]:.2f} max {lat[-1]:.2f}")
    for k in sorted(acc): print(f"   {k:28s} median {np.median(acc[k])*(1e3 if k.startswith('t_') else 1):8.3f}{' ms' if k.startswith('t_') else ''}")
    c.close()
