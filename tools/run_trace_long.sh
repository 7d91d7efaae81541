#!/bin/bash
# rocprofv3 kernel trace of the steady state of the batch pipeline (exp_groups.py, 4 groups of 8, host clouds) + concurrency analysis
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
D=$O/trace_long; rm -rf $D
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/exp_groups.py ${STEPS:-640} 4 8 1 > $D.log 2>&1
T=$(find $D -name "*kernel_trace.csv" | head -1)
tail -1 $D.log | cut -c1-300
python $R/tools/trace_analyze.py $T $O/trace_long.json 0.45 0.85 > $O/trace_long.txt 2>&1
cat $O/trace_long.txt
rm -rf $D
