#!/bin/bash
# A/B of two builds of the library on one box: gpurun_ab/old.so vs gpurun_ab/new.so, alternating
O=gpurun_out
for rep in 1 2 3; do for v in old new; do
  cp gpurun_ab/$v.so plade_amd/libplade_hip.so
  timeout 600 python tools/exp_groups.py 1536 ${AB_G:-4} ${AB_S:-4} ${AB_H:-0} > $O/ab_$v.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/ab_$v.json'))
print('$v', round(d['reg_per_s'],1), d['identical_to_single'], round(d['cpu_ms_per_registration'],2), round(d['busy_threads'],2))"
done; done
