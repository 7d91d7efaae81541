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
names=[re.sub(r"\(.*","",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("plade::","").replace("void ","")) for r in rows]
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
agg=collections.defaultdict(lambda:[0,0.0,0.0])
prev=None
for r,n in list(zip(rows,names))[start:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    a=agg[short(n)]; a[0]+=1; a[1]+=(e-s)/1e3; a[2]+=max(0.0,(s-prev)/1e3) if prev else 0.0
    prev=max(prev or 0,e)
tot=sum(a[1] for a in agg.values()); gaps=sum(a[2] for a in agg.values())
print(f"last registration: {sum(a[0] for a in agg.values())} kernels, {tot:.0f} us in kernels, {gaps:.0f} us of gaps before them")
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{k:42s} n {a[0]:4d} sum {a[1]:8.1f} us avg {a[1]/a[0]:7.1f} gap before avg {a[2]/a[0]:6.1f}")
PY
