"""What a fresh process pays before its first group returns: context creation, the first group (every work area is allocated
on the way), the second group (steady state).  python tools/time_startup.py [pairs per group]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import plade_amd
from plade_amd.synth import make_pair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = [make_pair(1000000, seed=s) for s in range(2)]
t1 = time.perf_counter()
c = plade_amd.Context(0, orient_normals=1, host_wait=1)
t2 = time.perf_counter()
grp = [(pairs[i % 2][0], pairs[i % 2][1]) for i in range(S)]
for tg, sr, _ in pairs:
    c.pin(tg); c.pin(sr)
t3 = time.perf_counter()
c.registration_pairs(grp)
t4 = time.perf_counter()
c.registration_pairs(grp)
t5 = time.perf_counter()
c.registration_pairs(grp)
t6 = time.perf_counter()
print(f"S={S}: context {1e3 * (t2 - t1):.0f} ms, pin 96 MB {1e3 * (t3 - t2):.0f} ms, first group {1e3 * (t4 - t3):.0f} ms, "
      f"second {1e3 * (t5 - t4):.0f} ms, third {1e3 * (t6 - t5):.0f} ms")
