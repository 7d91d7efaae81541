"""Steady-state throughput of M registrations in flight on one GPU (sleeping host waits, clouds resident in HBM) + the
plane-extraction statistics of one registration.  Knobs come from the environment (PLADE_* switches of the library).
    python tools/exp_throughput.py [steps] [inflight] [points]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
host = os.environ.get("EXP_HOST", "0") == "1"
pairs = [make_pair(n, seed=s) for s in range(2)]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(M)]
clouds = [[(c.upload(tg), c.upload(sr)) for (tg, sr, _) in pairs] for c in ctxs]
if host:
    for tg, sr, _ in pairs:
        ctxs[0].pin(tg); ctxs[0].pin(sr)
lock = threading.Lock()
nxt = [0]
done_t = []
res = {}


def work(w, total):
    while True:
        with lock:
            i = nxt[0]; nxt[0] += 1
        if i >= total:
            return
        if host:
            r = ctxs[w].registration(pairs[i % 2][0], pairs[i % 2][1])
        else:
            r = ctxs[w].registration_dev(*clouds[w][i % 2])
        t = time.perf_counter()
        with lock:
            done_t.append(t); res[i] = r


for w in range(M):
    for i in range(2):
        ctxs[w].registration_dev(*clouds[w][i])
W = 2 * M
total = W + K + M
cpu0 = time.process_time()
ths = [threading.Thread(target=work, args=(w, total)) for w in range(M)]
t_start = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
cpu1 = time.process_time()
done_t.sort()
t0, t1 = done_t[W - 1], done_t[W + K - 1]
ok = sum(bool(r[0]) for r in res.values())
same = all(np.array_equal(res[i][1], res[i % 2][1]) for i in res)
st = ctxs[0].stats()
keys = [k for k in st if k.startswith(("ransac_", "n_", "extract_"))]
print(json.dumps({"reg_per_s": K / (t1 - t0), "ms_per_step": (t1 - t0) / K * 1e3, "inflight": M, "steps": K, "host_clouds": host,
                  "bracketed_reg_per_s": total / (done_t[-1] - t_start), "ok": ok, "of": total, "identical": bool(same),
                  "busy_threads": (cpu1 - cpu0) / (done_t[-1] - t_start),
                  "env": {k: v for k, v in os.environ.items() if k.startswith(("PLADE_", "GPU_MAX"))},
                  "stats": {k: st[k] for k in keys}}))
