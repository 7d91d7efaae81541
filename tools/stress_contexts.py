"""Stress: several contexts registering concurrently on one GPU must reproduce the single-context results bit
for bit.  Prints the number of mismatching registrations (0 expected)."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pairs = [make_pair(n, seed=s) for s in (3, 4)]
ref = plade_amd.Context(0, dump=1, orient_normals=1)
want, want_dump = [], []
for (tg, sr, _) in pairs:
    want.append(ref.registration(tg, sr))
    want_dump.append(ref.dump())
ref.close()
bad = []
lock = threading.Lock()


def work(w):
    c = plade_amd.Context(0, dump=1, orient_normals=1)
    for rep in range(reps):
        for i, (tg, sr, _) in enumerate(pairs):
            ok, T = c.registration(tg, sr)
            if ok != want[i][0] or not np.array_equal(T, want[i][1]):
                d = c.dump()
                first = [k for k in want_dump[i] if k in d and not (np.asarray(d[k]).shape == np.asarray(want_dump[i][k]).shape
                                                                     and np.array_equal(d[k], want_dump[i][k]))]
                with lock:
                    bad.append((w, rep, i, first[:6]))
    c.close()


ths = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
for t in ths: t.start()
for t in ths: t.join()
print(f"n={n} workers={workers} reps={reps}: {len(bad)} mismatching registrations of {workers * reps * len(pairs)}")
for b in bad[:6]:
    print("  ", b)
