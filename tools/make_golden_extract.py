"""G2 (extract() level): the reference's auto-tuning loop extract() (code/PLADE/plade.cpp:602-635) driven over the REAL
Schnabel RANSAC (oracle/_ref, built from /root/reference by oracle/ref/Makefile) on the reference's sample clouds, for
several pinned time() seeds (libransac seeds its RNG from time(), so its plane count at a given min_support varies run
to run).  Records, per cloud and seed, the plane count of every detect call of the halving loop and the min_support the
loop ends at.  Run in the build container only:   python tools/make_golden_extract.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
R = Reference()
SEEDS = [1, 2, 3, 5, 8, 13, 21, 34]


def extract_trace(cloud, seed, min_num=10, max_num=40):
    """plade.cpp:602-635: detect at 10000; > 40 planes -> top 40, done; < 10 -> halve (<= 9 more calls, floor 200)."""
    ms, trials = 10000, 1
    trace = []
    pl = R.ransac_detect(cloud, ms, fake_time=seed)
    trace.append((ms, len(pl[0])))
    final_ms = ms
    ms //= 2
    while len(pl[0]) < min_num and trials < 10 and ms >= 200:
        pl = R.ransac_detect(cloud, ms, fake_time=seed)
        trace.append((ms, len(pl[0])))
        final_ms = ms
        ms //= 2
        trials += 1
    return trace, final_ms, min(len(pl[0]), max_num)


g8 = np.load(os.path.join(OUT, "g8_polyhedron.npz"))
g9 = np.load(os.path.join(OUT, "g9_room.npz"))
clouds = {"poly_t": g8["target"], "poly_s": g8["source"], "room_t": g9["target"], "room_s": g9["source"]}
out = {"seeds": np.array(SEEDS, np.int32)}
for name, c in clouds.items():
    finals, counts, traces = [], [], []
    for s in SEEDS:
        tr, fm, P = extract_trace(c, s)
        finals.append(fm); counts.append(P)
        row = np.full((10, 2), -1, np.int32)
        row[:len(tr)] = tr
        traces.append(row)
        print(name, s, tr, flush=True)
    out[name + "_final_min_support"] = np.array(finals, np.int32)
    out[name + "_final_planes"] = np.array(counts, np.int32)
    out[name + "_trace"] = np.array(traces, np.int32)
np.savez_compressed(os.path.join(OUT, "g2_extract.npz"), **out)
print("written", os.path.join(OUT, "g2_extract.npz"))
