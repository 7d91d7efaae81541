"""What do foreign kernels cost the registrations running beside them?  4 groups of 4 pairs in flight (tools/exp_groups.py's
pipeline) while a background context launches a diagnostic kernel as fast as it can:
    python tools/exp_interference.py <mode>     none | empty1 (1 workgroup) | empty4k (4096 workgroups that return at once) |
                                                 stream64 (64 MB read per launch, 2048 workgroups) | stream8 (8 MB, 256 workgroups)
Prints registrations/s and the background launches per second."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
K, G, S = 768, 4, 4
pairs = [make_pair(1000000, seed=s) for s in range(2)]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(G)]
clouds = [[(c.upload(tg), c.upload(sr)) for (tg, sr, _) in pairs] for c in ctxs]
bg = plade_amd.Context(0, host_wait=1)
spec = {"none": None, "empty1": (1, 0), "empty4k": (4096, 0), "stream64": (2048, 64), "stream8": (256, 8)}[mode]
stop = threading.Event()
bg_count = [0]


def background():
    while not stop.is_set():
        bg.diag_launches(200, spec[0], spec[1])
        bg_count[0] += 200


lock = threading.Lock()
done_t = []


def work(w, n_groups):
    for j in range(w, n_groups, G):
        ctxs[w].registration_pairs_dev([clouds[w][(j * S + q) % 2] for q in range(S)])
        t = time.perf_counter()
        with lock:
            done_t.extend([t] * S)


for w in range(G):
    ctxs[w].registration_pairs_dev([clouds[w][q % 2] for q in range(S)])
if spec:
    bg.diag_launches(10, spec[0], spec[1])
    bt = threading.Thread(target=background)
    bt.start()
lead, n_groups = 4 * G, 4 * G + K // S + G
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(w, n_groups)) for w in range(G)]
for t in ths: t.start()
for t in ths: t.join()
t1 = time.perf_counter()
stop.set()
if spec:
    bt.join()
done_t.sort()
W = lead * S
print(json.dumps({"mode": mode, "reg_per_s": K / (done_t[W + K - 1] - done_t[W - 1]), "background_launches_per_s": bg_count[0] / (t1 - t0)}))
