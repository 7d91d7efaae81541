#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for mode in own rocprim; do
  if [ $mode = rocprim ]; then export PLADE_SORT_ROCPRIM=1; fi
  rm -rf $O/sort_$mode
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/sort_$mode -o t -- python $R/tools/sort_bench.py > $O/sort_$mode.log 2>&1
  python - <<PY
import csv,glob,re
rows=list(csv.DictReader(open(glob.glob("$O/sort_$mode/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
print("== $mode")
# group launches between copyBuffer boundaries: print sequences compactly
seq=[]
for r in rows:
    n=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "copyBuffer" in n:
        if seq: print(" ".join(seq)); seq=[]
        continue
    tag="H" if "histogram" in n else ("P" if ("k_rs_pass" in n or "onesweep_iteration" in n or "sort" in n) else ("F" if "fill" in n else "o"))
    seq.append(f"{tag}{d:.1f}")
if seq: print(" ".join(seq))
PY
done
