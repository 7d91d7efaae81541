#!/bin/bash
# kernel + memory-copy trace of the host-buffer leg of the bench (clouds in page-locked host memory)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace_host -o t -- python $R/bench.py --steps 64 --warmup 8 --host-steps 400 --profiled-steps 0 --no-cpu-baseline > $O/trace_host.log 2>&1
tail -1 $O/trace_host.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident', d['value'], 'host', d['host_buffers_rank0'])"
ls $O/trace_host
head -8 $O/trace_host/*memory_copy_stats.csv 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob("$O/trace_host/*memory_copy_trace.csv")
if f:
    rows=list(csv.DictReader(open(f[0])))
    big=[r for r in rows if int(r.get("Bytes",r.get("bytes","0")) or 0)>1000000] if rows and ("Bytes" in rows[0] or "bytes" in rows[0]) else []
    print(len(rows),"copies", list(rows[0].keys()) if rows else "")
    import statistics
    for r in rows[:0]: pass
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
    d.sort()
    print("copy durations us: median",statistics.median(d),"p90",d[int(0.9*len(d))],"max",d[-1], "n>300us", sum(1 for x in d if x>300), "sum ms of those", sum(x for x in d if x>300)/1e3)
PY
find $O/trace_host -name "*_trace.csv" -delete
