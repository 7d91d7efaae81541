#!/bin/bash
# rocprofv3 kernel trace of tools/exp_groups.py for a few (groups, pairs per group) configurations + concurrency analysis
# usage: run_trace.sh "<G S> <G S> ..."   (outputs gpurun_out/trace_GxS.json / .txt)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  G=$1; S=$2; H=${3:-0}
  D=$O/trace_${G}x${S}
  rm -rf $D
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/exp_groups.py 96 $G $S $H > $D.log 2>&1
  T=$(find $D -name "*kernel_trace.csv" | head -1)
  echo "== $G x $S (host $H): $(tail -1 $D.log | cut -c1-200)"
  python $R/tools/trace_analyze.py $T $O/trace_${G}x${S}.json > $O/trace_${G}x${S}.txt 2>&1
  cat $O/trace_${G}x${S}.txt
  rm -rf $D
done
