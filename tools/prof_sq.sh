#!/bin/bash
# Where the wavefronts' time goes, kernel by kernel, under the load of the timed pipeline: one rocprofv3 --pmc pass with SQ counters
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, in quad-cycles) over the
# same command as tools/profile_bench.sh.  Output: gpurun_out/prof_sq_summary.csv (per kernel: launches, waves, wave cycles and
# the three shares), copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BENCH_LEAD_ROUNDS=1 BENCH_MIN_ROUNDS=${BENCH_MIN_ROUNDS:-6}
CMD="python $R/bench.py --pairs 16 --steps 96 --warmup 8 --resident-steps 0 --closed-form-steps 0 --no-cpu-baseline --no-cli --no-default-mode --no-parity --profiled-steps 0"
rm -rf $O/prof_sq
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/prof_sq -o sq -- $CMD > $O/prof_sq.log 2>&1
grep "registrations executed" $O/prof_sq.log
python3 - <<PY
import csv, glob, collections, re, subprocess
f = glob.glob("$O/prof_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (r["Dispatch_Id"], k)
    if d not in seen: seen.add(d); n[k] += 1
def short(nm):
    if nm.startswith("_Z"):
        nm = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip() or nm
    for j in ("void ", "plade::", "(anonymous namespace)::"): nm = nm.replace(j, "")
    m = re.match(r"k_batch<&\(?([\w<>, ]+?)\(", nm)
    if m:
        N = re.search(r"\), \d+, (\d+), Pack", nm)
        return ("B%s:" % (N.group(1) if N else "?")) + m.group(1)[:40]
    return nm.split("(")[0][:44]
rows = []
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    rows.append((wc, short(k), n[k], c.get("SQ_WAVES", 0), c.get("SQ_WAIT_ANY", 0), c.get("SQ_WAIT_INST_ANY", 0), c.get("SQ_ACTIVE_INST_ANY", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows) or 1.0
with open("$O/prof_sq_summary.csv", "w") as o:
    o.write("kernel,launches,waves,wave_quad_cycles,share_of_all_wave_cycles,wait_any,wait_inst_any,active_inst_any,valu_insts_per_wave,sq_busy_cycles,grbm_gui_active\n")
    for wc, nm, ln, wv, wa, wi, ai, vi, bz, ga in rows:
        o.write(f"{nm},{ln},{wv:.0f},{wc:.0f},{wc/tot:.4f},{wa/max(wc,1):.3f},{wi/max(wc,1):.3f},{ai/max(wc,1):.3f},{vi/max(wv,1):.1f},{bz:.0f},{ga:.0f}\n")
print(open("$O/prof_sq_summary.csv").read()[:6000])
PY
find $O/prof_sq -name "*counter_collection.csv" -size +20M -delete
