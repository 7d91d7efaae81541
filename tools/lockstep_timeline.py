"""Timeline of a group's tail in lock step (PLADE_TRACE_LOCKSTEP=1): G groups of S pairs in flight on host clouds, the [lockstep]
lines of the library's stderr aggregated by wait number within a call -> mean host gap / issue / GPU wait per wait.
    PLADE_TRACE_LOCKSTEP=1 python tools/lockstep_timeline.py [groups per worker] [G] [S] 2> trace.txt ; python tools/lockstep_timeline.py --parse trace.txt"""
import os, sys, re, collections
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = collections.defaultdict(list)
    per_call = collections.defaultdict(int)
    first = {}
    lines = [l for l in open(sys.argv[2]) if l.startswith("[lockstep]")]
    lines = lines[len(lines) // 2:]          # the second half of the run: first-use allocations and graph captures are behind it
    while lines and " wait 1 " not in lines[0]: lines.pop(0)
    for l in lines:
        m = re.match(r"\[lockstep\] (\S+) stage (\S*) wait (\d+) members (\d+) host_gap_us (\d+) queued (\d+) commands (\d+) issue_us (\d+) gpu_wait_us (\d+)", l)
        if not m: continue
        comb, stage, w, mem, gap, qd, cmd, iss, gw = m.groups()
        rows[(int(w), stage)].append((int(gap), int(qd), int(cmd), int(iss), int(gw), int(mem)))
    tot = [0, 0, 0]
    print("wait stage                 n   host_gap  queued commands issue_us gpu_wait_us members")
    for (w, stage), v in sorted(rows.items()):
        n = len(v)
        mean = [sum(x[i] for x in v) / n for i in range(6)]
        print(f"{w:4d} {stage:20s} {n:4d} {mean[0]:9.0f} {mean[1]:7.1f} {mean[2]:8.1f} {mean[3]:8.0f} {mean[4]:11.0f} {mean[5]:7.1f}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import threading
import plade_amd
from plade_amd.synth import make_pair
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
NP = 16
pairs = [make_pair(1000000, seed=s) for s in range(NP)]
ctxs = [plade_amd.Context(0, orient_normals=1, host_wait=1) for _ in range(G)]
for tg, sr, _ in pairs:
    ctxs[0].pin(tg); ctxs[0].pin(sr)
def grp(j):
    return [(pairs[(j * S + q) % NP][0], pairs[(j * S + q) % NP][1]) for q in range(S)]
def work(w, n):
    for j in range(w, n * G, G):
        ctxs[w].registration_pairs(grp(j), grp(j + G))
ths = [threading.Thread(target=work, args=(w, NG)) for w in range(G)]
for t in ths: t.start()
for t in ths: t.join()
