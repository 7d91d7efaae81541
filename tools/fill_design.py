"""Fills the @TOKENS@ of DESIGN.md's measurement table from profiles/<round>_bench_n1.json (run after the final bench)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "DESIGN.md")
d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_n1.json")))
r = d["roofline"]
import csv
_rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"))))
_tot = sum(float(x["TotalDurationNs"]) for x in _rows)
_k1 = sum(float(x["TotalDurationNs"]) for x in _rows if any(k in x["Name"] for k in ("k_r_mark(", "k_r_rescore(", "k_r_score_sub("))) / _tot
_rp = r["rocprof"]
_idle = _rp["launches_per_registration"] - r["launches_per_step"]
tok = {
    "@K1SHARE@": f"{100 * _k1:.1f}", "@RPTOTAL@": f"{_rp['launches_per_registration'] * _rp['avg_launch_us']:.0f}",
    "@IDLEUS@": f"{(_rp['launches_per_registration'] * _rp['avg_launch_us'] - r['launches_per_step'] * r['avg_launch_us']) / max(_idle, 1e-9):.1f}",
    "@VALUE@": f"{d['value']:.0f}", "@MS@": f"{d['ms_per_step']:.2f}", "@BUSY@": f"{d['host_rank0']['busy_host_threads_avg']:.1f}",
    "@HOSTVALUE@": f"{d['host_buffers_rank0']['value']:.0f}", "@LAT@": f"{d['single_registration_latency_ms']:.1f}",
    "@RPUS@": f"{r['rocprof']['avg_launch_us']:.1f}", "@RPLAUNCH@": f"{r['rocprof']['launches_per_registration']:.1f}",
    "@RPFRAC@": f"{r['rocprof']['frac_at_rocprof_average']:.2f}", "@EVUS@": f"{r['avg_launch_us_hip_events']:.1f}",
    "@TRAFFIC@": f"{r['traffic_per_step'] / 1e9:.3f}", "@ALGO@": f"{r['algorithmic_bytes_per_step'] / 1e9:.3f}",
    "@KERNELS@": f"{r['rocprof']['kernels_per_registration']:.0f}", "@COPIES@": f"{r['rocprof']['copies_and_fills_per_registration']:.0f}",
    "@GPUMS@": f"{r['rocprof']['gpu_ms_per_registration']:.1f}",
    "@WORK@": f"{r['launches_per_step']:.0f}", "@ALGOL@": f"{r['algorithmic_bytes_per_launch'] / 1e6:.1f}",
    "@IDLE@": f"{r['rocprof']['launches_per_registration'] - r['launches_per_step']:.0f}",
    "@STEPB@": f"{r['step_algorithmic_bytes'] / 1e9:.2f}", "@STEPFRAC@": f"{100 * r['step_frac_of_hbm_peak']:.1f}",
    "@MARKUS@": f"{r['avg_launch_us']:.1f}", "@MARKTB@": f"{r['achieved'] / 1e3:.2f}", "@MARKFRAC@": f"{r['frac']:.2f}",
}
s = open(src).read()
for k, v in tok.items():
    s = s.replace(k, v)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print(tok)
