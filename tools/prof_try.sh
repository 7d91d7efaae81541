#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 24 --warmup 8 --resident-steps 0 --no-cpu-baseline --no-default-mode --profiled-steps 2"
for mode in plain nograph plain inflight1; do
  rm -rf $O/try_$mode
  if [ $mode = nograph ]; then export PLADE_NO_GRAPH=1; else unset PLADE_NO_GRAPH; fi
  EXTRA=""; if [ $mode = inflight1 ]; then EXTRA="--inflight 2"; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/try_$mode -o t -- $CMD $EXTRA > $O/try_$mode.log 2>&1
  echo "$mode rc=$? $(grep -c SIGSEGV $O/try_$mode.log) $(tail -c 150 $O/try_$mode.log | tr '\n' ' ')"
  find $O/try_$mode -name "*_trace.csv" -delete
done
