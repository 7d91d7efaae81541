import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, plade_amd
g = np.load("tests/golden/g8_polyhedron.npz")
ctx = plade_amd.Context(0)
for cloud, rc, ro, ri, nm in ((g["target"], g["t_coef"], g["t_off"], g["t_idx"], "tgt"), (g["source"], g["s_coef"], g["s_off"], g["s_idx"], "src")):
    coef, off, idx = ctx.extract_planes(cloud, 625)
    sets = [set(idx[off[p]:off[p + 1]].tolist()) for p in range(len(coef))]
    worst = []
    for p in range(len(rc)):
        ref = set(ri[ro[p]:ro[p + 1]].tolist())
        cos = coef[:, :3] @ rc[p, :3]
        js = [(len(sets[q] & ref) / len(sets[q] | ref), q) for q in range(len(coef))]
        j, q = max(js)
        worst.append((abs(cos[q]), abs(coef[q, 3] - rc[p, 3] * np.sign(cos[q])), j, len(ref), len(sets[q])))
    worst.sort()
    print(nm, len(rc), len(coef), "lowest |cos|:", [(round(float(a), 6), round(float(b), 5), round(j, 3), n, m) for a, b, j, n, m in worst[:4]])
