"""Extraction iterations and registration latency for both hypothesis schedules over cloud sizes (development helper)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    import numpy as np, plade_amd
    from plade_amd.synth import make_pair
    out = {}
    for n in (100000, 200000, 500000, 1000000, 2000000, 4000000):
        ctx = plade_amd.Context(0, orient_normals=1)
        its, lat, errs = [], [], []
        for seed in (0, 1, 2):
            tg, sr, Tgt = make_pair(n, seed=seed)
            ct, cs = ctx.upload(tg), ctx.upload(sr)
            ctx.registration_dev(ct, cs)
            t = time.perf_counter(); ok, T = ctx.registration_dev(ct, cs); lat.append(time.perf_counter() - t)
            its.append(int(ctx.stats()["ransac_iterations"])); errs.append(float(np.linalg.norm(T - Tgt)) if ok else 9.9)
            ct.free(); cs.free()
        ctx.close()
        out[str(n)] = {"iterations": its, "ms": [round(1e3 * x, 2) for x in lat], "err": [round(e, 4) for e in errs]}
    print(json.dumps(out))
else:
    for mode in ("0", "1"):
        env = dict(os.environ, PLADE_RANSAC_TOPUP=mode)
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
        if r.returncode: print(r.stderr[-1500:])
        res = json.loads(r.stdout.strip().splitlines()[-1])
        for n, v in res.items(): print("topup", mode, "n", n, v)
