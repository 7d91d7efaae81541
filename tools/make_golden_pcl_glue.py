"""G10 / G11 (round 3): golden vectors for the two pieces of PCL glue that are radius-search compositions over FLANN --
ClusterTransformation (util.cpp:1245-1277 over conditional_euclidean_clustering.hpp:42-138) and the walks of
AreTwoPlanesPenetrable (util.cpp:1379-1442) -- produced by the FLANN compositions in oracle/ref/ref_shim.cpp (built from
/root/reference by oracle/ref/Makefile).  Run in the build container only:

    python tools/make_golden_pcl_glue.py

Inputs of G10 are the candidate transformations of real registrations (the reference's polyhedron sample pair and the
room scan, planes from libransac: g8 / g9 fixtures, run through the oracle) plus synthetic sets with exact duplicates and
distances straddling the tolerance; inputs of G11 are synthetic pairs of noisy plane patches."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
R, O = Reference(), Oracle()
rng = np.random.default_rng(20260929)
f32 = np.float32

# ---- G10 ------------------------------------------------------------------------------------------
cases = {}
for name, fix, pre in (("poly", "g8_polyhedron.npz", ""), ("room", "g9_room.npz", ""), ("roomb", "g9_room.npz", "b")):
    g = np.load(os.path.join(OUT, fix), allow_pickle=False)
    tp = (g[f"t{pre}_coef"], g[f"t{pre}_off"], g[f"t{pre}_idx"])
    sp = (g[f"s{pre}_coef"], g[f"s{pre}_off"], g[f"s{pre}_idx"])
    ok, T, d = O.registration(g["target"], g["source"], tp, sp, voxel_sort_mode=0)
    rt = d["initial_RT"].reshape(-1, 12)
    keep = slice(0, min(len(rt), 6000))
    t, eul = rt[keep, 9:12].copy(), O.euler_angles(rt[keep, :9])
    s = f32(d["average_spacing"][0])
    dist_th = f32(np.float64(f32(s * 5)) / 2)                 # plade.cpp:47, util.cpp:331
    g_angle = f32(np.float64(f32(5.0 / 180 * np.pi)) / 2)
    cases[name] = (t, eul, dist_th, g_angle)
# synthetic: tight clumps, exact duplicates (distance 0 ties), pairs right at the tolerance, angle gate on/off
m = 3000
centres = rng.uniform(-2, 2, (40, 3))
t = (centres[rng.integers(0, 40, m)] + rng.normal(0, 0.02, (m, 3))).astype(f32)
eul = (rng.integers(0, 3, (m, 1)) * 0.1 + rng.normal(0, 0.02, (m, 3))).astype(f32)
t[100:200] = t[0:100]; eul[100:150] = eul[0:50]              # exact duplicates, half of them with equal angles
tol = f32(0.05)
t[300:400] = t[200:300] + np.array([tol, 0, 0], f32) * rng.choice([0.999999, 1.0, 1.000001], (100, 1)).astype(f32)
cases["synth"] = (t, eul, tol, f32(0.0436 / 2 * 0 + 0.0015))
out = {}
for name, (t, eul, dist_th, g_angle) in cases.items():
    lab, n = R.cluster_transforms(t, eul, dist_th, g_angle)
    lab_o, n_o = O.cluster_transforms(t, eul, dist_th, g_angle)
    print(f"G10 {name}: {len(t)} candidates -> {n} clusters (oracle {n_o}), largest {np.bincount(lab).max()}, labels equal: {np.array_equal(lab, lab_o)}")
    out.update({f"{name}_t": t, f"{name}_euler": eul, f"{name}_dist": dist_th, f"{name}_angle": g_angle, f"{name}_cluster_of": lab,
                f"{name}_n": np.int32(n)})
out["names"] = np.array(";".join(cases.keys()))
np.savez_compressed(os.path.join(OUT, "g10_cluster.npz"), **out)

# ---- G11 ------------------------------------------------------------------------------------------
def patch(n, origin, eu, ev, lu, lv, noise, nrm):
    uv = rng.random((n, 2)) * [lu, lv]
    return (origin + uv[:, :1] * eu + uv[:, 1:] * ev + rng.normal(0, noise, (n, 1)) * nrm).astype(f32)

walks = []
for case in range(24):
    r = f32(rng.uniform(0.05, 0.3))                          # searchRadius = lengthThreshold (util.cpp:495)
    spacing = r / 5
    # plane A: z = 0 patch; plane B: tilted plane through the x axis, crossing A (penetrating) or ending at it (touching)
    na = int(rng.integers(200, 1500)); nb = int(rng.integers(2, 1500)) if case % 6 else int(rng.integers(0, 3))
    tilt = rng.uniform(0.3, 1.5)
    nB = np.array([0, -np.sin(tilt), np.cos(tilt)])
    ptsA = patch(na, np.array([-1.0, -1.0, 0.0]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), 2.0, 2.0, 0.004, np.array([0, 0, 1.0]))
    lo = -1.0 if case % 2 else 0.0                           # B on both sides of A, or on one side only
    ev = np.array([0, np.cos(tilt), np.sin(tilt)])
    ptsB = patch(nb, np.array([-1.0, 0, 0]) + lo * ev, np.array([1.0, 0, 0]), ev, 2.0, 1.0 - lo, 0.004, nB)
    if case % 5 == 0:                                        # a hole in the gate cloud: steps get skipped
        ptsB = ptsB[np.abs(ptsB[:, 0]) > 0.4]
    planeB = np.append(nB, 0.0).astype(f32)
    planeA = np.array([0, 0, 1.0, 0.0], f32)
    start = np.array([-0.9 + rng.uniform(0, 0.3), 0, 0], f32)
    direc = np.array([1.0, 0, 0], f32)
    length = f32(rng.uniform(0.3, 1.8))
    if case % 7 == 3:
        length = f32(r * 4)                                  # an exact multiple of the step (float accumulation of dist)
    min_d = f32(spacing)                                     # minDistance (util.cpp:496 passes the point spacing scale)
    for (A, B, plB) in ((ptsA, ptsB, planeB), (ptsB, ptsA, planeA)):
        pos, neg, sk = R.pen_walk(A, B, plB, start, direc, length, r, min_d)
        po, no, so = O.pen_walk(A, B, plB, start, direc, length, r, min_d)
        walks.append(dict(a=A, b=B, plane=plB, start=start, direc=direc, length=length, r=r, min_d=min_d,
                          res=np.array([pos, neg, sk], np.int32), same=(pos, neg, sk) == (po, no, so)))
print("G11:", len(walks), "walks, oracle equal on", sum(w["same"] for w in walks), "; results:",
      [tuple(w["res"]) for w in walks[:8]])
out = {"n": np.int32(len(walks))}
for i, w in enumerate(walks):
    for k in ("a", "b", "plane", "start", "direc", "length", "r", "min_d", "res"):
        out[f"{k}_{i}"] = w[k]
np.savez_compressed(os.path.join(OUT, "g11_penetration.npz"), **out)
print(sorted(f for f in os.listdir(OUT) if f.startswith(("g10", "g11"))), os.path.getsize(os.path.join(OUT, "g10_cluster.npz")),
      os.path.getsize(os.path.join(OUT, "g11_penetration.npz")))
