"""Plane sets of the extraction with and without the pool top-up rule (development helper)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, plade_amd
    from plade_amd.synth import make_pair
    out = {}
    for seed in (0, 1, 2, 3):
        tg, sr, _ = make_pair(1000000, seed=seed)
        ctx = plade_amd.Context(0, orient_normals=1)
        for name, cl in (("t", tg), ("s", sr)):
            co, off, idx = ctx.extract_planes(cl, 10000)
            out[f"{seed}{name}"] = sorted((int(b - a) for a, b in zip(off[:-1], off[1:])), reverse=True)
        ctx.close()
    print(json.dumps(out))
else:
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, PLADE_RANSAC_TOPUP=mode)
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["0"]:
        a, b = res["0"][k], res["1"][k]
        print(k, "old", len(a), "new", len(b), "old-only tail", a[len(b):] if len(a) > len(b) else "", "new-only tail", b[len(a):] if len(b) > len(a) else "")
        print("   old", a)
        print("   new", b)
