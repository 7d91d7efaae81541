import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, plade_amd
g = np.load("tests/golden/g8_polyhedron.npz")
ctx = plade_amd.Context(0)
ctx.extract_planes(g["target"], 625)
ctx.set_params(dump=2)
co, off, idx = ctx.extract_planes(g["target"], 625)
print("planes", len(co), sorted(np.diff(off).tolist()), file=sys.stderr)
