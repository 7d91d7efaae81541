"""Copies the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/summarize_profiles.py r1

profiles/<round>_kernel_stats.csv : `rocprofv3 --kernel-trace --stats` of `BENCH_LEAD_ROUNDS=1 python bench.py --steps 96 --warmup 8 --resident-steps 0 --no-cpu-baseline --no-default-mode --profiled-steps 2` (tools/profile_bench.sh)
profiles/<round>_pmc_hbm.csv      : per-kernel average FETCH_SIZE / WRITE_SIZE (separate --pmc passes), with the gfx950
                                    FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied in the `hbm_read_bytes` column
profiles/<round>_profile_meta.json: registrations the profiled command ran (its "[bench] registrations executed" line)
profiles/<round>_pmc_calibration.json : tools/pmc_calibrate.sh, when it has been run (gpurun_out/pmc_calibration.json)

FETCH_SIZE on gfx950 counts one 64-byte unit per memory-side read request.  Calibrated on kernels of known byte counts
(tools/pmc_calibrate.hip, r5): coalesced streams of 16 B AND of 4 B per lane move 2.000 x FETCH_SIZE x 1024 bytes (128-byte
requests tallied at 64: the guide's factor); index-driven gathers of 4-12 useful bytes per lane report 1.03-1.05 x the 64-byte
sectors they touch, i.e. FETCH_SIZE x 1024 is the sector traffic itself and the x2 column overstates them by up to 2x;
WRITE_SIZE x 1024 is exact for coalesced 16 B stores.  The csv therefore carries both bounds and a class per kernel:
  stream  (x2 is the figure)   scans, sorts, copies, voxel keys / centroids, view rebuild
  gather  (x1 .. x2)           index-driven reads: k_gather_cloud, k_vb_runs, k_r_select_cc, k_r_sample, k_sp_knn, k_overlap,
                               k_pen_walk, k_cluster_edges, k_cell_spans, k_rank_lists, k_match
  mixed   (between)            everything else
"""
GATHER = ("k_gather_cloud", "k_vb_runs", "k_voxel_runs", "k_r_select_cc", "k_r_sample", "k_sp_knn", "k_overlap", "k_pen_walk", "k_cluster_edges",
          "k_cell_spans", "k_rank_lists", "k_match", "k_occ_start", "k_gather_cells", "k_pen_cell_fill", "k_src_gather", "k_knn_grid")
STREAM = ("k_r_mark", "k_r_rescore", "k_r_score_sub", "k_rs_pass", "k_rs_histogram", "k_finish_uploads", "k_vb_keys", "k_voxel_keys", "k_vb_centroids",
          "k_voxel_centroids", "k_view_count", "k_view_compact", "k_morton", "k_tile_boxes", "k_ranges", "k_copy_out", "k_r_compact_raster", "k_scan_u32",
          "copyBuffer", "fillBuffer")


def klass(name):
    if any(k in name for k in GATHER):
        return "gather"
    if any(k in name for k in STREAM):
        return "stream"
    return "mixed"

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
    w.writerow(["kernel", "launches", "FETCH_SIZE_avg_KB", "WRITE_SIZE_avg_KB", "hbm_read_bytes(FETCH_SIZE*1024*2)", "hbm_write_bytes(WRITE_SIZE*1024)",
                "class", "hbm_read_bytes_low(FETCH_SIZE*1024*1)"])
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        n, v = fe[k]
        wv = wr.get(k, [1, 0.0])
        w.writerow([k, n, f"{v / n:.1f}", f"{wv[1] / max(wv[0], 1):.1f}", f"{v / n * 1024 * 2:.0f}", f"{wv[1] / max(wv[0], 1) * 1024:.0f}",
                    klass(k), f"{v / n * 1024:.0f}"])
import json
import re
for log in ("prof_bench.log", "prof_bench/bench.log"):
    lp = os.path.join(go, log)
    if os.path.exists(lp):
        m = re.search(r"registrations executed by rank 0: (\d+)", open(lp, errors="replace").read())
        if m:
            json.dump({"registrations": int(m.group(1)), "command": "tools/profile_bench.sh (kernel-trace pass)"},
                      open(os.path.join(out, f"{rnd}_profile_meta.json"), "w"))
            break
cal = os.path.join(go, "pmc_calibration.json")
if os.path.exists(cal):
    shutil.copy(cal, os.path.join(out, f"{rnd}_pmc_calibration.json"))
src = os.path.join(go, f"bench_{rnd}.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(out, f"{rnd}_bench_n1.json"))
src = os.path.join(go, f"bench_{rnd}_driver.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(out, f"{rnd}_bench_n1_driver_command.json"))
print(sorted(os.listdir(out)))
