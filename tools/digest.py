"""Digest of the extracted planes + transforms of a few pairs (compare two builds of the library: LIBP=path)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plade_amd
if os.environ.get("LIBP"):
    plade_amd.load_library(os.environ["LIBP"])
from plade_amd.synth import make_pair
h = hashlib.sha256()
for n, seeds in ((1000000, (0, 1, 2)), (200000, (10, 11, 12, 13))):
    for seed in seeds:
        tg, sr, _ = make_pair(n, seed=seed)
        ctx = plade_amd.Context(0, orient_normals=1, dump=1)
        ok, T = ctx.registration(tg, sr)
        d = ctx.dump()
        for k in ("tgt_planes", "tgt_plane_offsets", "tgt_plane_idx", "src_planes", "src_plane_offsets", "src_plane_idx"):
            h.update(np.ascontiguousarray(d[k]).tobytes())
        h.update(np.ascontiguousarray(T).tobytes())
        ctx.close()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g8_polyhedron.npz"))
ctx = plade_amd.Context(0, orient_normals=1)
for cloud in (g["target"], g["source"]):
    coef, off, idx = ctx.extract_planes(cloud, 625)
    h.update(coef.tobytes()); h.update(off.tobytes()); h.update(idx.tobytes())
print("digest", h.hexdigest())
