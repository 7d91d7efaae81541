"""Kernel-trace analysis of a bench run: how busy is the GPU, how many kernels overlap, what fills the time."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"], r.get("Queue_Id", "")))
ev.sort()
# the steady state: from the start of the kernel at the 40th percentile (by launch order) to that of the 70th
# (the trace also covers scene generation, warm-up and the CPU legs, where the GPU has nothing to do)
a, b = ev[int(0.40 * len(ev))][0], ev[int(0.70 * len(ev))][0]
mid = [(max(s, a), min(e, b), n, q) for s, e, n, q in ev if e > a and s < b]
pts = []
for s, e, n, q in mid:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0.0; conc_time = collections.Counter(); cur = 0; last = a
for t, d in pts:
    if cur > 0: busy += t - last
    conc_time[min(cur, 8)] += t - last
    cur += d; last = t
tot = sum(e - s for s, e, _, _ in mid)
print(f"window {(b - a) / 1e6:.1f} ms, kernels {len(mid)}, sum of durations {tot / 1e6:.1f} ms, GPU busy (>=1 kernel) {busy / (b - a) * 100:.1f} %, mean concurrency while busy {tot / max(busy, 1):.2f}")
print("time share by number of kernels running:", {k: round(v / (b - a), 3) for k, v in sorted(conc_time.items())})
queues = collections.Counter(q for _, _, _, q in mid)
print("queues:", dict(queues))
def nm(n):
    n = n.replace("(anonymous namespace)::", "").replace("plade::", ""); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:36]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in mid:
    agg[nm(n)][0] += 1; agg[nm(n)][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:38s} n {v[0]:6d} share {v[1] / tot * 100:5.1f} %  avg {v[1] / v[0] / 1e3:7.1f} us")
