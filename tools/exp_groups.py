"""Steady-state throughput of G groups in flight on one GPU, S pairs per group (plade_registration_pairs[_dev]): the A/B of
round 4 (8 x 1 vs 4 x 2 vs 8 x 2 ...).  Every result is compared bit for bit with the same pair registered alone.
    python tools/exp_groups.py [steps] [groups in flight] [pairs per group] [host: 0 resident | 1 host clouds + prefetch] [points]
"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
host = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = int(sys.argv[5]) if len(sys.argv) > 5 else 1000000
NP = 2
pairs = [make_pair(n, seed=s) for s in range(NP)]
if os.environ.get("EXP_ORDER") == "scan":
    # the same points in the order a spinning scanner at the centre of the cloud would deliver them (256 elevation rings,
    # azimuth within a ring) instead of the generator's random order: what the index-driven gathers cost depends on it
    def scan_order(c):
        d = c[:, :3] - c[:, :3].mean(0)
        ring = np.floor((np.arctan2(d[:, 2], np.hypot(d[:, 0], d[:, 1])) / np.pi + 0.5) * 256).astype(np.int64)
        return np.ascontiguousarray(c[np.lexsort((np.arctan2(d[:, 1], d[:, 0]), ring))])
    pairs = [(scan_order(tg), scan_order(sr), T) for tg, sr, T in pairs]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(G)]
clouds = [[(c.upload(tg), c.upload(sr)) for (tg, sr, _) in pairs] for c in ctxs]
if host:
    for tg, sr, _ in pairs:
        ctxs[0].pin(tg); ctxs[0].pin(sr)
# reference results: every pair alone
ref = [ctxs[0].registration_dev(*clouds[0][i]) for i in range(NP)]
lock = threading.Lock()
done_t, res = [], {}


def group_items(j):      # group number j registers the steps j*S .. j*S + S - 1
    return [(j * S + q) for q in range(S)]


def work(w, n_groups):
    for j in range(w, n_groups, G):
        items = group_items(j)
        if host:
            cur = [(pairs[i % NP][0], pairs[i % NP][1]) for i in items]
            jn = j + G
            nxt = [(pairs[i % NP][0], pairs[i % NP][1]) for i in group_items(jn)] if jn < n_groups else None
            out = ctxs[w].registration_pairs(cur, nxt)
        else:
            out = ctxs[w].registration_pairs_dev([clouds[w][i % NP] for i in items])
        t = time.perf_counter()
        with lock:
            for i, r in zip(items, out):
                done_t.append(t); res[i] = r


# warm-up: every context runs every combination once
for w in range(G):
    for j in range(NP):
        ctxs[w].registration_pairs_dev([clouds[w][(j * S + q) % NP] for q in range(S)])
        if host:
            ctxs[w].registration_pairs([(pairs[(j * S + q) % NP][0], pairs[(j * S + q) % NP][1]) for q in range(S)])
lead_groups = 4 * G
n_groups = lead_groups + (K + S - 1) // S + G
cpu0 = time.process_time()
ths = [threading.Thread(target=work, args=(w, n_groups)) for w in range(G)]
t_start = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
cpu1 = time.process_time()
done_t.sort()
W = lead_groups * S
t0, t1 = done_t[W - 1], done_t[W + K - 1]
ok = sum(bool(r[0]) for r in res.values())
same = all(np.array_equal(res[i][1], ref[i % NP][1]) and res[i][0] == ref[i % NP][0] for i in res)
st = ctxs[0].stats()
keys = [k for k in st if k.startswith(("ransac_", "n_", "cpu_", "t_"))]
st1 = ctxs[0].stats(pair=1) if S > 1 else {}
total = n_groups * S
print(json.dumps({"reg_per_s": K / (t1 - t0), "ms_per_step": (t1 - t0) / K * 1e3, "groups_in_flight": G, "pairs_per_group": S, "steps": K,
                  "host_clouds": host, "bracketed_reg_per_s": total / (done_t[-1] - t_start), "ok": ok, "of": total,
                  "identical_to_single": bool(same), "busy_threads": (cpu1 - cpu0) / (done_t[-1] - t_start),
                  "cpu_ms_per_registration": (cpu1 - cpu0) / total * 1e3,
                  "env": {k: v for k, v in os.environ.items() if k.startswith(("PLADE_", "GPU_MAX", "EXP_"))},
                  "stats": {k: st[k] for k in keys}, "stats_pair1": {k: v for k, v in st1.items() if k.startswith(("cpu_", "t_"))}}))
