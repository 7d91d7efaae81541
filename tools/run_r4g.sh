#!/bin/bash
# two processes on one GPU: is the limit in the process (runtime / host) or on the GPU?
O=gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('$1'))
print(round(d['reg_per_s'],1), d['identical_to_single'], 'busy', round(d['busy_threads'],2), 'cpu/reg', round(d['cpu_ms_per_registration'],2), d['ok'], d['of'])
"; }
for cfg in "4 4" "2 4" "3 4"; do
  set -- $cfg
  timeout 600 python tools/exp_groups.py 1536 $1 $2 0 > $O/r4g_a_$1x$2.json 2> $O/r4g_a.err &
  P1=$!
  timeout 600 python tools/exp_groups.py 1536 $1 $2 0 > $O/r4g_b_$1x$2.json 2> $O/r4g_b.err &
  P2=$!
  wait $P1 $P2
  echo "== two processes, each $1 x $2: A $(show $O/r4g_a_$1x$2.json)"
  echo "                               B $(show $O/r4g_b_$1x$2.json)"
done
