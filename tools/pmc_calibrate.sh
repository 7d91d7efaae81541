#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 on kernels of known byte counts (tools/pmc_calibrate.hip): factors for tools/summarize_profiles.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/cal_fetch $O/cal_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o c -- $R/tools/pmc_calibrate > $O/cal_expected.json 2> $O/cal_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o c -- $R/tools/pmc_calibrate > /dev/null 2> $O/cal_write.log
python3 - <<PY
import csv, glob, json, collections
exp = json.loads([l for l in open("$O/cal_expected.json") if l.startswith("{")][-1])
def pmc(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"$O/{d}/**/*counter_collection.csv", recursive=True)[0])):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe, wr = pmc("cal_fetch"), pmc("cal_write")
out = {"expected": exp, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "ratios": {}}
def name(k): return next((n for n in fe if k in n), None)
for k, e in (("k_cal_stream16", exp["k_cal_stream16_read_bytes"]), ("k_cal_stream4", exp["k_cal_stream4_read_bytes"])):
    out["ratios"][k + ": bytes read / (FETCH_SIZE x 1024)"] = e / (fe[name(k)] * 1024)
g = fe[name("k_cal_gather12")] * 1024
out["ratios"]["k_cal_gather12: (64 B sectors + index) / (FETCH_SIZE x 1024)"] = (exp["k_cal_gather12_sector64_bytes"] + exp["k_cal_gather12_index_bytes"]) / g
out["ratios"]["k_cal_gather12: (128 B lines + index) / (FETCH_SIZE x 1024)"] = (exp["k_cal_gather12_line128_bytes_upper"] + exp["k_cal_gather12_index_bytes"]) / g
g = fe[name("k_cal_gather4")] * 1024
out["ratios"]["k_cal_gather4: (64 B sectors + index) / (FETCH_SIZE x 1024)"] = (exp["k_cal_gather4_sector64_bytes"] + exp["k_cal_gather4_index_bytes"]) / g
out["ratios"]["k_cal_write16: bytes written / (WRITE_SIZE x 1024)"] = exp["k_cal_write16_written_bytes"] / (wr[next(n for n in wr if "k_cal_write16" in n)] * 1024)
json.dump(out, open("$O/pmc_calibration.json", "w"), indent=1)
print(json.dumps(out["ratios"], indent=1))
PY
