# K8 stress under rocprofv3: kernel statistics + the two HBM counter passes (separate --pmc runs, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python $R/tools/k8_stress.py 10000000 10000 64 > $O/k8_stress.json 2> $O/k8_stress.err; tail -c 1500 $O/k8_stress.json
rm -rf $O/prof_k8 $O/prof_k8_fetch $O/prof_k8_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k8 -o k8 -- python $R/tools/k8_stress.py 10000000 10000 0 > $O/prof_k8.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_k8_fetch -o pmc -- python $R/tools/k8_stress.py 10000000 10000 0 > $O/prof_k8_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_k8_write -o pmc -- python $R/tools/k8_stress.py 10000000 10000 0 > $O/prof_k8_write.log 2>&1
find $O/prof_k8 -name "*kernel_trace.csv" -delete
cp $(find $O/prof_k8 -name "*kernel_stats.csv" | head -1) $O/k8_kernel_stats.csv
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out"
def pmc(which):
    agg=collections.defaultdict(lambda:[0,0.0])
    f=glob.glob(f"{O}/prof_k8_{which}/**/*counter_collection.csv",recursive=True)
    for r in csv.DictReader(open(f[0])):
        a=agg[r["Kernel_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    return agg
fe,wr=pmc("fetch"),pmc("write")
with open(f"{O}/k8_pmc_hbm.csv","w",newline="") as f:
    w=csv.writer(f); w.writerow(["kernel","launches","hbm_read_bytes_total(FETCH_SIZE*1024*2)","hbm_write_bytes_total(WRITE_SIZE*1024)"])
    for k in sorted(fe,key=lambda k:-fe[k][1])[:12]:
        w.writerow([k,fe[k][0],f"{fe[k][1]*2048:.0f}",f"{wr.get(k,[0,0])[1]*1024:.0f}"])
print(open(f"{O}/k8_pmc_hbm.csv").read()[:1500])
PY
find $O/prof_k8_fetch $O/prof_k8_write -name "*counter_collection.csv" -delete
