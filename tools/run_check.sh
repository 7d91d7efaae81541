#!/bin/bash
# quick regression + throughput check of a change inside the extraction loop
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_seams.py tests/test_gpu_groups.py tests/test_gpu_registration.py tests/test_gpu_ransac.py tests/test_gpu_golden.py tests/test_gpu_edge.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
python tools/digest.py 2>&1 | tail -1
show() { python -c "
import json,sys
d=json.load(open('$1'))
print(round(d['reg_per_s'],1), d['identical_to_single'], 'busy', round(d['busy_threads'],2), 'cpu/reg', round(d['cpu_ms_per_registration'],2), d['ok'], d['of'])
"; }
for cfg in "4 8 0" "4 8 1" "4 8 0" "4 8 1"; do
  set -- $cfg
  timeout 600 python tools/exp_groups.py 1536 $1 $2 $3 > $O/chk_$1x$2_h$3.json 2> $O/chk_$1x$2_h$3.err
  echo "== groups $1 x $2 host=$3: $(show $O/chk_$1x$2_h$3.json)"
done
