"""Throughput of M registrations in flight on one GPU (one plade_ctx + host thread each)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
pairs = [make_pair(n, seed=s) for s in range(2)]
for M in (1, 2, 3, 4, 6):
    ctxs = [plade_amd.Context(0, orient_normals=1) for _ in range(M)]
    clouds = [[(c.upload(tg), c.upload(sr)) for (tg, sr, _) in pairs] for c in ctxs]
    K = 12
    errs = []

    def work(w):
        for i in range(K):
            ct, cs = clouds[w][i % 2]
            ok, T = ctxs[w].registration_dev(ct, cs)
            errs.append(float(np.linalg.norm(T - pairs[i % 2][2])) if ok else 1e9)

    for w in range(M):  # warm-up
        ctxs[w].registration_dev(*clouds[w][0]); ctxs[w].registration_dev(*clouds[w][1])
    ths = [threading.Thread(target=work, args=(w,)) for w in range(M)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"in flight {M}: {M * K / dt:7.1f} reg/s   ({dt / (M * K) * 1e3:.2f} ms per registration, max err {max(errs):.2e})", flush=True)
    for c in ctxs: c.close()
