cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/k8q $O/k8q_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k8q -o k8 -- python $R/tools/k8_stress.py 10000000 10000 0 > $O/k8q.log 2>&1
find $O/k8q -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob("$O/k8q/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, "us")
PY
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/k8q_sq -o sq -- python $R/tools/k8_stress.py 10000000 10000 0 > $O/k8q_sq.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/k8q_sq/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)): acc[r["Kernel_Name"]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:4]:
    wc=c["SQ_WAVE_CYCLES"]; print(k[:80], "waves %.3g wavecyc %.3g wait_any %.2f wait_inst %.2f active %.2f valu/wave %.0f"%(c["SQ_WAVES"],wc,c["SQ_WAIT_ANY"]/wc,c["SQ_WAIT_INST_ANY"]/wc,c["SQ_ACTIVE_INST_ANY"]/wc,c["SQ_INSTS_VALU"]/c["SQ_WAVES"]))
PY
find $O/k8q_sq -name "*counter_collection.csv" -delete
