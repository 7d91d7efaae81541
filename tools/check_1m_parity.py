"""Full-size parity: GPU registration of 1M-point pairs vs the oracle run on the planes the GPU extracted
(every dumped intermediate, bit-exact).  Takes ~10 s of CPU per pair."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from oracle.oracle import Oracle
from plade_amd.synth import make_pair

orc = Oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for seed in (0, 1):
    tg, sr, Tgt = make_pair(n, seed=seed)
    ctx = plade_amd.Context(0, dump=1, orient_normals=1)
    ok, T = ctx.registration(tg, sr)
    d = ctx.dump()
    tp = (d["tgt_planes"].reshape(-1, 4), d["tgt_plane_offsets"], d["tgt_plane_idx"])
    sp = (d["src_planes"].reshape(-1, 4), d["src_plane_offsets"], d["src_plane_idx"])
    t0 = time.perf_counter()
    ok_o, T_o, do = orc.registration(tg, sr, tp, sp, voxel_sort_mode=1)
    dt = time.perf_counter() - t0
    bad = [k for k in do if k in d and not (np.asarray(d[k]).shape == np.asarray(do[k]).shape and np.array_equal(d[k], do[k]))]
    print(f"seed {seed}: ok {ok}/{ok_o}  T equal {np.array_equal(T, T_o)}  |T-Tgt| {np.linalg.norm(T - Tgt):.2e}  "
          f"oracle |T-Tgt| {np.linalg.norm(T_o - Tgt):.2e}  compared {len([k for k in do if k in d])} arrays, mismatching {bad}  "
          f"(oracle {dt:.1f} s)  scores top3 {np.sort(d['scores'])[-3:]}", flush=True)
    ctx.close()
