#!/bin/bash
# kernel trace of a few single registrations (one in flight): the ordered command list of one registration
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_one -o t -- python $R/tools/time_reg.py > $O/trace_one.log 2>&1
python - <<PY
import csv,glob,re,collections
f=glob.glob("$O/trace_one/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last registration = last quarter of rows roughly; find boundaries by k_aos? use k_morton occurrences (2 per registration)
names=[re.sub(r"\(.*","",r["Kernel_Name"]).replace("plade::","") for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith("k_morton")]
start=min(idx[-2:])-40 if len(idx)>=2 else 0
def short(n):
    if "rocprim" in n:
        m=re.search(r"wrapped_(\w+?)_config",n); k="histogram" if "histogram" in n else ("iteration" if "onesweep_iteration" in n else "")
        return "rp:"+(m.group(1) if m else n[:30])+":"+k
    return n[:40]
out=open("$O/trace_one_seq.txt","w")
prev=None
for r,n in list(zip(rows,names))[start:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    out.write(f"{short(n):42s} dur {(e-s)/1e3:8.1f} gap {gap:8.1f} grid {r.get('Grid_Size','')} q {r.get('Queue_Id','')}\n")
    prev=e
c=collections.Counter(short(n) for n in names[start:])
print(c.most_common(60))
PY
