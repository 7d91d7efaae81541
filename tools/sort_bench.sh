#!/bin/bash
# per-pass durations of the radix sort alone on the GPU (rocprofv3 kernel trace of tools/sort_bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/sort_own
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/sort_own -o t -- python $R/tools/sort_bench.py > $O/sort_own.log 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$O/sort_own/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[]
for r in rows:
    n=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "copyBuffer" in n:
        if seq: print(" ".join(seq)); seq=[]
        continue
    tag="H" if "histogram" in n else ("P" if "k_rs_pass" in n else ("F" if "fill" in n else "o"))
    seq.append(f"{tag}{d:.1f}")
if seq: print(" ".join(seq))
PY
rm -rf $O/sort_own
