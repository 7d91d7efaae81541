"""Concurrency analysis of a rocprofv3 kernel trace (``*_kernel_trace.csv``): how many kernels run at once, how busy every
hardware queue is, per-kernel durations, and the per-registration sums -- over the middle part of the run (steady state).
    python tools/trace_analyze.py <kernel_trace.csv> [out.json] [lo_frac hi_frac]
"""
import collections
import csv
import json
import sys

path = sys.argv[1]
out_path = sys.argv[2] if len(sys.argv) > 2 else None
lo_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.35
hi_f = float(sys.argv[4]) if len(sys.argv) > 4 else 0.9
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r.get("Stream_Id", "0"), r["Kernel_Name"]))
rows.sort()
t_min, t_max = rows[0][0], max(r[1] for r in rows)
lo, hi = t_min + lo_f * (t_max - t_min), t_min + hi_f * (t_max - t_min)
win = [r for r in rows if r[0] >= lo and r[1] <= hi]
wall = hi - lo


def short(name):
    for junk in ("void ", "plade::", "(anonymous namespace)::"):
        name = name.replace(junk, "")
    return name.split("(")[0][:44]


# concurrency histogram by a sweep over start / end events
ev = []
for s, e, q, st, n in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter()
cur, last = 0, lo
for t, d in ev:
    hist[cur] += t - last
    last = t
    cur += d
hist[cur] += hi - last
total_busy = sum(e - s for s, e, *_ in win)
# registrations in the window: one launch of the verification kernel per registration when the pairs run on their own; a MERGED
# launch (k_batch<k_overlap..., N = 8>) serves the pairs of a group -- PAIRS_PER_MERGED (environment, default 8) of them
import os
_ppm = int(os.environ.get("PAIRS_PER_MERGED", "8"))
regs = sum((_ppm if ("k_batch" in r[4] and "ELi8ENS_4Pack" in r[4]) else 1) for r in win if "k_overlap" in r[4])
per_q = collections.defaultdict(lambda: [0, 0])
for s, e, q, st, n in win:
    per_q[q][0] += e - s
    per_q[q][1] += 1
# gaps between consecutive kernels of one queue
gaps = collections.defaultdict(list)
last_end = {}
for s, e, q, st, n in win:
    if q in last_end:
        gaps[q].append(max(0, s - last_end[q]))
    last_end[q] = max(e, last_end.get(q, 0))
per_k = collections.defaultdict(lambda: [0, 0])
for s, e, q, st, n in win:
    k = short(n)
    per_k[k][0] += e - s
    per_k[k][1] += 1
streams = len({r[3] for r in win})
res = {
    "window_ms": wall / 1e6, "kernels": len(win), "registrations_started_in_window": regs, "streams_seen": streams,
    "registrations_per_s_in_window": regs / (wall / 1e9) if wall else None,
    "avg_concurrency": total_busy / wall,
    "gpu_ms_per_registration": total_busy / 1e6 / max(regs, 1),
    "kernels_per_registration": len(win) / max(regs, 1),
    "time_share_by_concurrency": {str(k): round(v / wall, 4) for k, v in sorted(hist.items())},
    "queues": {q: {"busy_share": round(v[0] / wall, 4), "kernels": v[1],
                   "median_gap_us": (sorted(gaps[q])[len(gaps[q]) // 2] / 1e3 if gaps[q] else None),
                   "mean_gap_us": (sum(gaps[q]) / len(gaps[q]) / 1e3 if gaps[q] else None)} for q, v in sorted(per_q.items())},
    "kernels_by_time": [{"kernel": k, "per_reg_us": round(v[0] / 1e3 / max(regs, 1), 1), "avg_us": round(v[0] / v[1] / 1e3, 2),
                         "calls_per_reg": round(v[1] / max(regs, 1), 2)} for k, v in sorted(per_k.items(), key=lambda kv: -kv[1][0])[:40]],
}
txt = json.dumps(res, indent=1)
if out_path:
    open(out_path, "w").write(txt)
print(json.dumps({k: res[k] for k in ("window_ms", "registrations_per_s_in_window", "avg_concurrency", "gpu_ms_per_registration",
                                      "kernels_per_registration", "time_share_by_concurrency", "queues")}))
for k in res["kernels_by_time"][:28]:
    print(f"  {k['kernel']:44s} {k['per_reg_us']:8.1f} us/reg  avg {k['avg_us']:7.2f} us  x{k['calls_per_reg']}")
