#!/bin/bash
# where the CLI's wall time goes at the HIP API level: 16 distinct 1M-point pairs as PLY files, a 64-line file_pairs.txt,
# `PLADE list out` under rocprofv3 --hip-trace --stats (API statistics only: no counters)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; D=/dev/shm/plade_cli_trace; mkdir -p $D $O
cd $R
python - <<PY
import sys; sys.path.insert(0, "$R")
from plade_amd.synth import make_pair
from plade_amd.plyio import write_ply
import multiprocessing as mp
def gen(s):
    tg, sr, _ = make_pair(1000000, seed=s); write_ply("$D/t%d.ply" % s, tg); write_ply("$D/s%d.ply" % s, sr)
if __name__ == "__main__":
    with mp.Pool(8) as p: p.map(gen, range(16))
    open("$D/list.txt", "w").write("".join("$D/t%d.ply\n$D/s%d.ply\n" % (i % 16, i % 16) for i in range(64)))
PY
export PLADE_ORIENT_NORMALS=1
export PLADE_CLI_FULL_EXIT=1
for v in pool nopool pool nopool; do
  if [ $v = nopool ]; then export PLADE_EXP_NO_POOL=1; else unset PLADE_EXP_NO_POOL; fi
  t0=$(date +%s%N); PLADE_DEBUG_ALLOC=1 $R/plade_amd/PLADE $D/list.txt $D/out.txt > /dev/null 2> $O/cli_run.err; t1=$(date +%s%N)
  echo "$v full-exit wall $(( (t1 - t0) / 1000000 )) ms; $(grep -c transformation: $D/out.txt) blocks; $(grep "device allocations" $O/cli_run.err | tail -1)"
  t0=$(date +%s%N); PLADE_CLI_FULL_EXIT= env -u PLADE_CLI_FULL_EXIT $R/plade_amd/PLADE $D/list.txt $D/out2.txt > /dev/null 2>&1; t1=$(date +%s%N)
  echo "$v normal wall $(( (t1 - t0) / 1000000 )) ms; same result file: $(cmp -s $D/out.txt $D/out2.txt && echo yes || echo NO)"
done; unset PLADE_EXP_NO_POOL
cd /tmp && export TMPDIR=/tmp; rm -rf $O/cli_hip
rocprofv3 --hip-trace --stats --output-format csv -d $O/cli_hip -o cli -- $R/plade_amd/PLADE $D/list.txt $D/out.txt > $O/cli_hip.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/cli_hip/**/*hip_api_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"]:36s} calls {int(r["Calls"]):7d} total ms {float(r["TotalDurationNs"])/1e6:9.1f} avg us {float(r["AverageNs"])/1e3:9.1f}')
PY
find $O/cli_hip -name "*trace.csv" -delete; rm -rf $D
