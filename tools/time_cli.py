"""CLI end-to-end on the GPU box ("report separately the CLI end-to-end (adds PLY parse)"):
single pair and batch mode of plade_amd/PLADE on binary 1M-point PLY pairs written to a temp dir."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plade_amd.plyio import write_ply
from plade_amd.synth import make_pair

CLI = os.path.join(ROOT, "plade_amd", "PLADE")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = tempfile.mkdtemp(prefix="plade_cli_")
files = []
for s in range(2):
    tg, sr, _ = make_pair(n, seed=s)
    pt, ps = os.path.join(d, f"t{s}.ply"), os.path.join(d, f"s{s}.ply")
    write_ply(pt, tg); write_ply(ps, sr)
    files.append((pt, ps))
print(f"PLY size {os.path.getsize(files[0][0]) / 1e6:.1f} MB per cloud", flush=True)
for rep in range(2):   # second run: page cache + GPU code objects warm
    t0 = time.perf_counter()
    r = subprocess.run([CLI, files[0][0], files[0][1], os.path.join(d, "one.txt")], capture_output=True, text=True)
    print(f"single pair, process start to exit: {time.perf_counter() - t0:.3f} s (rc {r.returncode})", flush=True)
lst = os.path.join(d, "pairs.txt")
with open(lst, "w") as f:
    for i in range(npairs):
        f.write(f"{files[i % 2][0]}\n{files[i % 2][1]}\n")
for infl in [int(v) for v in os.environ.get("CLI_INFLIGHT", "1,4,8").split(",")]:
    env = dict(os.environ, PLADE_INFLIGHT=str(infl), PLADE_GPUS="1")
    t0 = time.perf_counter()
    r = subprocess.run([CLI, lst, os.path.join(d, "batch.txt")], capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    print(f"batch of {npairs} pairs, PLADE_INFLIGHT={infl}: {dt:.3f} s = {npairs / dt:.1f} pairs/s end to end (rc {r.returncode})", flush=True)
