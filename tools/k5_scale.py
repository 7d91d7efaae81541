"""K5 (descriptor radius match) at the size SURVEY.md 8d worries about: ~100 mutually non-parallel target planes
=> ~5 000 intersection lines => D_t ~ 2e7 ordered line-pair descriptors (the Manhattan scenes of the bench have
D_t ~ 1e4).  A synthetic "crystal" scene: P random planar patches in a 12 m cube; the source holds a subset of them,
moved by a random SE(3).  Planes are handed over (plade.h:74) so that only the registration stages run.

    python tools/k5_scale.py [P_target=100] [P_source=60] [points_per_patch=20000]

Prints the stage times and descriptor / match counts; run under `rocprofv3 --kernel-trace --stats` for the kernel table.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import random_se3

PT = int(sys.argv[1]) if len(sys.argv) > 1 else 100
PS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
rng = np.random.default_rng(7)


def patch(center, normal, size, n, seed):
    r = np.random.default_rng(seed)
    a = np.cross(normal, [0.3, 0.5, 0.8]); a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = (r.random((n, 2)) - 0.5) * size
    p = center + uv[:, :1] * a + uv[:, 1:] * b + normal * r.normal(0, 0.003, (n, 1))
    nn = normal + r.normal(0, 0.02, (n, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    return np.concatenate([p, nn], 1).astype(np.float32)


normals = rng.normal(size=(PT, 3)); normals /= np.linalg.norm(normals, axis=1, keepdims=True)
centers = (rng.random((PT, 3)) - 0.5) * 12.0
clouds, labels = [], []
for i in range(PT):
    clouds.append(patch(centers[i], normals[i], 2.5, NP, 100 + i))
    labels.append(np.full(NP, i, np.int32))
tg = np.concatenate(clouds); tl = np.concatenate(labels)
T = random_se3(3)
Ti = np.linalg.inv(T)
keep = np.sort(rng.permutation(PT)[:PS])
sel = np.isin(tl, keep)
src = tg[sel]
sr = np.concatenate([src[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3], src[:, 3:].astype(np.float64) @ Ti[:3, :3].T], 1).astype(np.float32)
sl = tl[sel]


def planes(cloud, lab):
    coef, off, idx = [], [0], []
    for f in np.unique(lab):
        ids = np.nonzero(lab == f)[0].astype(np.int32)
        p = cloud[ids, :3].astype(np.float64)
        c = p.mean(0)
        _, _, vt = np.linalg.svd(p - c, full_matrices=False)
        n = vt[2]
        if cloud[ids, 3:].astype(np.float64).mean(0) @ n < 0:
            n = -n
        coef.append([*n, -(n @ c)]); idx.append(ids); off.append(off[-1] + len(ids))
    return np.asarray(coef, np.float32), np.asarray(off, np.int32), np.concatenate(idx)


tp, sp = planes(tg, tl), planes(sr, sl)
print(f"scene: {len(tg)} + {len(sr)} points, {len(tp[0])} + {len(sp[0])} planes", flush=True)
ctx = plade_amd.Context(0, max_planes=max(PT, 40), max_candidates=200)
if os.environ.get("PLADE_MATCH_WINDOW"):
    print("PLADE_MATCH_WINDOW =", os.environ["PLADE_MATCH_WINDOW"])
for it in range(3):
    t0 = time.perf_counter()
    try:
        ok, Tr = ctx.registration_planes(tg, sr, tp, sp)
    except plade_amd.PladeError as e:
        print("error:", e)
        break
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(f"iter {it}: ok {ok} {dt * 1e3:.1f} ms  |T - T_gt|_F {np.linalg.norm(Tr - T):.2e}  D_t {int(st['n_descriptors_tgt'])} D_s {int(st['n_descriptors_src'])} "
          f"matches {int(st['n_matches'])}  t_match {st['t_match'] * 1e3:.2f} ms  t_transforms {st['t_transforms'] * 1e3:.2f} ms "
          f"t_cluster {st['t_cluster'] * 1e3:.2f} ms", flush=True)
for k, v in ctx.stats().items():
    if k.startswith(("t_", "n_")):
        print(f"  {k:28s} {v:.6g}")
