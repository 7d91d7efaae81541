"""Steady-state throughput of M registrations in flight on one GPU (sleeping host waits, clouds resident in HBM) + the
plane-extraction statistics of one registration.  Knobs come from the environment (PLADE_* switches of the library).
    python tools/exp_throughput.py [steps] [inflight] [points]"""
import json
import os
import sys
import threading
import time

import numpy as np
if os.environ.get("EXP_TORCH") == "after":      # the library (and with it the SYSTEM HIP runtime) first, torch second
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import plade_amd as _p
    _p.load_library()
    import torch
    torch.cuda.init()
    _x = torch.arange(8, device="cuda") * 2
    torch.cuda.synchronize()
    print("torch after the library: cuda ok", _x.sum().item(), torch.version.hip, flush=True)
    print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:2], flush=True)
elif float(os.environ.get("EXP_BG_COPY", "0")) > 0 or os.environ.get("EXP_TORCH"):
    import torch
    if os.environ.get("EXP_TORCH") != "import":
        torch.cuda.init()
        torch.cuda.synchronize()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
hostmode = int(os.environ.get("EXP_HOST", "0"))
host = hostmode > 0
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


tup = []


def work(w, total):
    if hostmode == 2:
        for i in range(w, total, M):
            j = i + M
            nx = pairs[j % 2] if j < total else (None, None)
            r = ctxs[w].registration_next(pairs[i % 2][0], pairs[i % 2][1], nx[0], nx[1])
            t = time.perf_counter()
            _st = ctxs[w].stats(); tu = (_st.get("t_upload_take", 0.0), _st.get("t_upload_submit", 0.0))
            with lock:
                done_t.append(t); res[i] = r; tup.append(tu)
        return
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
bg_stop = threading.Event()
bg_count = [0]
if float(os.environ.get("EXP_BG_COPY", "0")) > 0:
    import torch
    rate = float(os.environ["EXP_BG_COPY"])          # pairs of 24 MB copies per second
    src_t = [torch.empty(6 * n, dtype=torch.float32).pin_memory() for _ in range(2)]
    dst_t = [torch.empty(6 * n, dtype=torch.float32, device="cuda") for _ in range(2)]
    bg_stream = torch.cuda.Stream()

    def bg():
        nxt_t = time.perf_counter()
        while not bg_stop.is_set():
            with torch.cuda.stream(bg_stream):
                dst_t[0].copy_(src_t[0], non_blocking=True)
                dst_t[1].copy_(src_t[1], non_blocking=True)
            bg_stream.synchronize()
            bg_count[0] += 1
            nxt_t += 1.0 / rate
            d = nxt_t - time.perf_counter()
            if d > 0:
                time.sleep(d)
    bgt = threading.Thread(target=bg)
    bgt.start()
cpu0 = time.process_time()
ths = [threading.Thread(target=work, args=(w, total)) for w in range(M)]
t_start = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
cpu1 = time.process_time()
bg_stop.set()
done_t.sort()
t0, t1 = done_t[W - 1], done_t[W + K - 1]
ok = sum(bool(r[0]) for r in res.values())
same = all(np.array_equal(res[i][1], res[i % 2][1]) for i in res)
st = ctxs[0].stats()
keys = [k for k in st if k.startswith(("ransac_", "n_", "extract_"))]
print(json.dumps({"reg_per_s": K / (t1 - t0), "ms_per_step": (t1 - t0) / K * 1e3, "inflight": M, "steps": K, "host_clouds": hostmode, "t_upload_take_submit_ms_avg": [sum(t[0] for t in tup) / len(tup) * 1e3, sum(t[1] for t in tup) / len(tup) * 1e3] if tup else None,
                  "bracketed_reg_per_s": total / (done_t[-1] - t_start), "ok": ok, "of": total, "identical": bool(same),
                  "busy_threads": (cpu1 - cpu0) / (done_t[-1] - t_start),
                  "bg_copy_pairs": bg_count[0], "env": {k: v for k, v in os.environ.items() if k.startswith(("PLADE_", "GPU_MAX", "EXP_"))},
                  "stats": {k: st[k] for k in keys}}))
