"""Where the host CPU time of a registration goes: 8 contexts in flight (sleeping waits), cpu_* stats averaged."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plade_amd
from plade_amd.synth import make_pair
M, K = 8, 100
pairs = [make_pair(1000000, seed=s) for s in range(2)]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=2) for _ in range(M)]
for tg, sr, _ in pairs:
    ctxs[0].pin(tg); ctxs[0].pin(sr)
acc = {}
lock = threading.Lock()
def work(w):
    for i in range(K):
        tg, sr, _ = pairs[i % 2]
        ctxs[w].registration_next(tg, sr, pairs[(i + 1) % 2][0], pairs[(i + 1) % 2][1])
        if i >= 8:
            st = ctxs[w].stats()
            with lock:
                for k, v in st.items():
                    if k.startswith(("cpu_", "t_")): acc[k] = acc.get(k, 0.0) + v
def tasks():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) , int(rest[12])))   # utime, stime in ticks
        except Exception:
            pass
    return out
ths = [threading.Thread(target=work, args=(w,)) for w in range(M)]
tk0 = tasks()
c0, t0 = time.process_time(), time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
c1, t1 = time.process_time(), time.perf_counter()
tk1 = tasks()
tick = os.sysconf("SC_CLK_TCK")
print("threads alive at the end (user ms, sys ms over the run):")
for tid, (comm, (u, s_)) in sorted(tk1.items()):
    u0, s0 = tk0.get(tid, (comm, (0, 0)))[1]
    if (u - u0) + (s_ - s0) > 0:
        print(f"  {tid} {comm:20s} user {1e3*(u-u0)/tick:8.0f} sys {1e3*(s_-s0)/tick:8.0f}")
n = M * (K - 8)
print(f"rate {M*K/(t1-t0):.1f} reg/s, process cpu {1e3*(c1-c0)/(M*K):.2f} ms/reg, busy threads {(c1-c0)/(t1-t0):.2f}")
for k in sorted(acc):
    print(f"  {k:28s} {1e3*acc[k]/n:8.3f} ms/reg")
