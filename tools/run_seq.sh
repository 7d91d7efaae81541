#!/bin/bash
# the commands of ONE registration in launch order (kernel trace + memory-copy trace of a single pair registered alone)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
D=$O/seq; rm -rf $D
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o t -- python $R/tools/exp_groups.py 4 1 1 1 > $D.log 2>&1
tail -1 $D.log | cut -c1-200
K=$(find $D -name "*kernel_trace.csv" | head -1); M=$(find $D -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" > $O/seq.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for junk in ("void ", "plade::", "(anonymous namespace)::"):
        n = n.replace(junk, "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", n.split("(")[0][:40], r.get("Stream_Id", ""), r.get("Queue_Id", "")))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), r.get("Stream_Id", ""), ""))
except Exception as e:
    print("no copy trace", e)
rows.sort()
# the last registration: from the last k_finish_uploads (or k_morton) on
last = max(i for i, r in enumerate(rows) if "k_finish_uploads" in r[3] or "k_morton_keys" in r[3])
while last > 0 and rows[last][0] - rows[last - 1][1] < 200000 and "k_finish" not in rows[last - 1][3]: last -= 1
t0 = rows[last][0]
prev = t0
n = 0
for s, e, kind, name, st, q in rows[last:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:7.1f} us  {kind} {name:42s} stream {st} q {q}")
    prev = max(prev, e); n += 1
print("commands", n, "span us", (prev - t0) / 1e3)
PY
tail -3 $O/seq.txt
rm -rf $D
