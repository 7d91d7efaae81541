#!/bin/bash
# one group of eight 1M-point pairs at a time, nothing else on the GPU: the kernels' durations ALONE (rocprofv3 kernel statistics)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cat > /tmp/alone.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import numpy as np, plade_amd
from plade_amd.synth import make_pair
prs=[make_pair(1000000, seed=s)[:2] for s in range(8)]
c=plade_amd.Context(0, orient_normals=1, host_wait=1)
cl=[(c.upload(a), c.upload(b)) for a,b in prs]
for rep in range(6):
    r=c.registration_pairs_dev(cl)
print("registrations", 6*8)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_alone
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_alone -o a -- python /tmp/alone.py $R > $O/prof_alone.log 2>&1
find $O/prof_alone -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob,re
regs=48
f=glob.glob("$O/prof_alone/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("registrations", regs, "GPU ms/reg (alone)", round(tot/1e6/regs,3), "commands/reg", round(sum(int(r["Calls"]) for r in rows)/regs,1))
def short(n):
    m=re.search(r"k_batchITnDaXadL_ZNS_(?:12_GLOBAL__N_1)?\d+(k_\w+?)(?:I[A-Za-z0-9_]*?E)?E(?:RKNS|vRKNS)", n)
    if m: return "B:"+m.group(1)[:40]
    for j in ("void ","plade::","(anonymous namespace)::"): n=n.replace(j,"")
    return n.split("(")[0][:46]
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print(f'{short(r["Name"]):48s} calls/reg {int(r["Calls"])/regs:6.2f} avg_us {float(r["AverageNs"])/1e3:8.1f} us/reg {float(r["TotalDurationNs"])/1e3/regs:8.1f}')
PY
