#!/bin/bash
O=gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('$1'))
print(round(d['reg_per_s'],1), d['identical_to_single'], 'busy', round(d['busy_threads'],2), 'cpu/reg', round(d['cpu_ms_per_registration'],2))
print('   pair0', {k: round(v*1e3,2) for k,v in d['stats'].items() if k.startswith('t_')})
print('   pair1', {k: round(v*1e3,2) for k,v in d['stats_pair1'].items() if k.startswith('t_')})
"; }
for cfg in "8 1" "4 2" "8 2" "12 2"; do
  set -- $cfg
  timeout 600 python tools/exp_groups.py 512 $1 $2 0 > $O/r4b_g$1x$2.json 2> $O/r4b_g$1x$2.err
  echo "== groups $1 x $2 resident"; show $O/r4b_g$1x$2.json
done
for q in 2 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/exp_groups.py 512 6 2 0 > $O/r4b_q$q.json 2> $O/r4b_q$q.err
  echo "== GPU_MAX_HW_QUEUES=$q groups 6 x 2"; show $O/r4b_q$q.json
done
