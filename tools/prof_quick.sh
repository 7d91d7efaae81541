#!/bin/bash
# quick look: rocprofv3 kernel statistics of a short bench run, per registration (gpurun_out/prof_quick*)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BENCH_LEAD_ROUNDS=${BENCH_LEAD_ROUNDS:-1} BENCH_MIN_ROUNDS=${BENCH_MIN_ROUNDS:-6}
rm -rf $O/prof_quick
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_quick -o q -- python $R/bench.py --pairs 16 --steps 96 --warmup 8 --resident-steps 0 --closed-form-steps 0 --no-cpu-baseline --no-cli --no-default-mode --no-parity --profiled-steps 0 ${BENCH_EXTRA:-} > $O/prof_quick.log 2>&1
grep "registrations executed" $O/prof_quick.log
find $O/prof_quick -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob,re
regs=int(re.search(r"rank 0: (\d+)", open("$O/prof_quick.log").read()).group(1))
f=glob.glob("$O/prof_quick/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("registrations", regs, "GPU ms/reg", round(tot/1e6/regs,3), "commands/reg", round(sum(int(r["Calls"]) for r in rows)/regs,1))
import subprocess
def short(n):
    if n.startswith("_Z"):
        n=subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    for j in ("void ","plade::","(anonymous namespace)::"): n=n.replace(j,"")
    m=re.match(r"k_batch<&\(?([\w<>, ]+?)\(", n)
    if m:
        N=re.search(r"\), \d+, (\d+), Pack", n)
        return ("B%s:" % (N.group(1) if N else "?"))+m.group(1)[:44]
    return n.split("(")[0][:46]
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:60]:
    print(f'{short(r["Name"]):48s} calls/reg {int(r["Calls"])/regs:6.2f} avg_us {float(r["AverageNs"])/1e3:8.1f} us/reg {float(r["TotalDurationNs"])/1e3/regs:8.1f}')
PY
