"""Copies the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/summarize_profiles.py r1

profiles/<round>_kernel_stats.csv : `rocprofv3 --kernel-trace --stats` of `BENCH_LEAD_ROUNDS=1 python bench.py --steps 96 --warmup 8 --resident-steps 0 --no-cpu-baseline --no-default-mode --profiled-steps 2` (tools/profile_bench.sh)
profiles/<round>_pmc_hbm.csv      : per-kernel average FETCH_SIZE / WRITE_SIZE (separate --pmc passes), with the gfx950
                                    FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied in the `hbm_read_bytes` column
"""
import collections
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
go = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
import glob
stats = glob.glob(os.path.join(go, "prof_bench", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(out, f"{rnd}_kernel_stats.csv"))


def pmc(which):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(go, f"prof_pmc_{which}", "**", "*counter_collection.csv"), recursive=True)
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


fe, wr = pmc("fetch"), pmc("write")
with open(os.path.join(out, f"{rnd}_pmc_hbm.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_avg_KB", "WRITE_SIZE_avg_KB", "hbm_read_bytes(FETCH_SIZE*1024*2)", "hbm_write_bytes(WRITE_SIZE*1024)"])
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n, v = fe[k]
        wv = wr.get(k, [1, 0.0])
        w.writerow([k, n, f"{v / n:.1f}", f"{wv[1] / max(wv[0], 1):.1f}", f"{v / n * 1024 * 2:.0f}", f"{wv[1] / max(wv[0], 1) * 1024:.0f}"])
src = os.path.join(go, f"bench_{rnd}.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(out, f"{rnd}_bench_n1.json"))
src = os.path.join(go, f"bench_{rnd}_driver.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(out, f"{rnd}_bench_n1_driver_command.json"))
print(sorted(os.listdir(out)))
