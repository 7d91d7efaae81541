#!/bin/bash
# the batch pipeline at smaller clouds: what part of the step is independent of the data?  (4 groups of 8, host clouds)
for n in 20000 100000 300000 1000000; do
timeout 300 python tools/exp_groups.py 1024 4 8 1 $n > gpurun_out/size.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/size.json'))
print('points $n:', round(d['reg_per_s'],1), 'registrations/s', round(1e3/d['reg_per_s'],3), 'ms', d['identical_to_single'], 'cpu ms', round(d['cpu_ms_per_registration'],2), 'threads', round(d['busy_threads'],2))"
done
