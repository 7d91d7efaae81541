"""CLI timing with PLADE_TRACE_CLI on the GPU box: single pair and a 64-pair list over 16 distinct 1M-point pairs."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plade_amd.plyio import write_ply
from bench import generate_pairs
CLI = os.path.join(ROOT, "plade_amd", "PLADE")


def main():
    d = tempfile.mkdtemp(prefix="plade_cli_", dir="/dev/shm")
    pairs = generate_pairs(1000000, list(range(16)), 16)
    names = []
    for k, (tg, sr, _) in enumerate(pairs):
        a, b = os.path.join(d, f"t{k}.ply"), os.path.join(d, f"s{k}.ply")
        write_ply(a, tg); write_ply(b, sr); names.append((a, b))
    env = dict(os.environ, PLADE_ORIENT_NORMALS="1")
    for rep in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([CLI, names[0][0], names[0][1], os.path.join(d, "one.txt")], capture_output=True, text=True, env=dict(env, PLADE_TRACE_CLI="1"))
        dt = time.perf_counter() - t0
        print(f"single pair: {dt:.3f} s rc {r.returncode}")
        if rep == 2: print(r.stderr[-1500:])
    for infl, grp in ((2, 4), (3, 4), (4, 4), (2, 8), (4, 2)):
        lst = os.path.join(d, "pairs64v.txt")
        with open(lst, "w") as f:
            for i in range(64):
                f.write(f"{names[i % 16][0]}\n{names[i % 16][1]}\n")
        best = 9
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([CLI, lst, os.path.join(d, "out.txt")], capture_output=True, text=True, env=dict(env, PLADE_INFLIGHT=str(infl), PLADE_GROUP=str(grp)))
            best = min(best, time.perf_counter() - t0)
        print(f"64 pairs, {infl} workers x groups of {grp}: {best:.3f} s")
    for n in (64, 512):
        lst = os.path.join(d, f"pairs{n}.txt")
        with open(lst, "w") as f:
            for i in range(n):
                f.write(f"{names[i % 16][0]}\n{names[i % 16][1]}\n")
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([CLI, lst, os.path.join(d, "out.txt")], capture_output=True, text=True, env=dict(env, PLADE_TRACE_CLI="1" if (rep == 1 and n == 64) else ""))
            dt = time.perf_counter() - t0
            ok = open(os.path.join(d, "out.txt")).read().count("transformation:")
            print(f"{n}-pair list: {dt:.3f} s = {n / dt:.1f} pairs/s, registered {ok}, rc {r.returncode}")
            if rep == 1 and n == 64:
                print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[plade") and ("main" in l or "batch" in l or "context" in l))[:3000])
    import shutil; shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
