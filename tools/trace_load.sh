#!/bin/bash
# kernel trace of a bench run under load (8 registrations in flight) + concurrency analysis; $1 = tag, $2... = extra env
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-load}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -o t -- python $R/bench.py --steps ${STEPS:-200} --warmup 20 --inflight ${INFLIGHT:-8} --host-steps 0 --profiled-steps 0 --no-cpu-baseline > $O/trace_$TAG.log 2>&1
python $R/tools/concurrency.py $O/trace_$TAG > $O/conc_$TAG.txt 2>&1
rm -rf $O/trace_$TAG
tail -1 $O/trace_$TAG.log | cut -c1-300
cat $O/conc_$TAG.txt
