"""Where the host CPU of the batch pipeline goes: G groups of S pairs in flight on host clouds (as bench.py times them), the cpu_* and
lock-step statistics of every pair summed, the process' threads by CPU time.    python tools/cpu_groups.py [groups per worker] [G] [S]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plade_amd
from plade_amd.synth import make_pair
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 12
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
NP = 16
pairs = [make_pair(1000000, seed=s) for s in range(NP)]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(G)]
for tg, sr, _ in pairs:
    ctxs[0].pin(tg); ctxs[0].pin(sr)
acc, cnt = {}, [0]
lock = threading.Lock()

def grp(j):
    return [(pairs[(j * S + q) % NP][0], pairs[(j * S + q) % NP][1]) for q in range(S)]

def work(w, n, collect):
    for j in range(w, n * G, G):
        ctxs[w].registration_pairs(grp(j), grp(j + G))
        if collect:
            with lock:
                for q in range(S):
                    for k, v in ctxs[w].stats(pair=q).items():
                        if k.startswith(("cpu_", "lockstep_", "t_", "ransac_iterations")): acc[k] = acc.get(k, 0.0) + v
                cnt[0] += S

def tasks():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, int(rest[11]), int(rest[12]))
        except Exception:
            pass
    return out

for phase, n in (("warm", 3), ("timed", NG)):
    ths = [threading.Thread(target=work, args=(w, n, phase == "timed")) for w in range(G)]
    tk0, c0, t0 = tasks(), time.process_time(), time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    tk1, c1, t1 = tasks(), time.process_time(), time.perf_counter()
regs = NG * G * S
tick = os.sysconf("SC_CLK_TCK")
print(f"rate {regs/(t1-t0):.1f} reg/s, process cpu {1e3*(c1-c0)/regs:.2f} ms/reg, busy threads {(c1-c0)/(t1-t0):.2f}")
by = {}
for tid, (comm, u, s_) in tk1.items():
    _, u0, s0 = tk0.get(tid, (comm, 0, 0))
    a = by.setdefault(comm, [0, 0, 0]); a[0] += u - u0; a[1] += s_ - s0; a[2] += 1
print("threads by name (user ms/reg, sys ms/reg, count; short-lived threads that ended before the sample are missing):")
for comm, (u, s_, n) in sorted(by.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    if u + s_: print(f"  {comm:20s} user {1e3*u/tick/regs:7.3f} sys {1e3*s_/tick/regs:7.3f}  x{n}")
print("stats per registration:")
for k in sorted(acc):
    print(f"  {k:32s} {acc[k]/cnt[0]*(1e3 if k.startswith(('cpu_','t_')) else 1):9.3f}{' ms' if k.startswith(('cpu_','t_')) else ''}")
