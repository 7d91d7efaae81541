#!/bin/bash
# hardware queues x groups in flight with the round's final kernels (r3's answer was 4 queues)
O=gpurun_out; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('$1'))
print(round(d['reg_per_s'],1), d['identical_to_single'], 'busy', round(d['busy_threads'],2), 'cpu/reg', round(d['cpu_ms_per_registration'],2), d['ok'], d['of'])
"; }
for cfg in "4 4 4" "8 4 4" "8 6 4" "8 8 4" "6 6 4" "4 6 4" "4 4 4" "8 8 2"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 600 python tools/exp_groups.py 1536 $2 $3 0 > $O/q$1_$2x$3.json 2> $O/q$1_$2x$3.err
  echo "== queues $1 groups $2 x $3 resident: $(show $O/q$1_$2x$3.json)"
done
