#!/bin/bash
# round 4, first GPU pass: the new group tests + the suites touched by the refactor, then the A/B of groups
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_groups.py tests/test_gpu_ransac.py tests/test_gpu_golden.py tests/test_gpu_seams.py tests/test_gpu_registration.py -x -q -m gpu > gpurun_out/r4a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4a_tests.log
tail -15 gpurun_out/r4a_tests.log
for cfg in "8 1" "4 2" "6 2" "8 2"; do
  set -- $cfg
  timeout 600 python tools/exp_groups.py 512 $1 $2 0 > gpurun_out/r4a_g$1x$2.json 2> gpurun_out/r4a_g$1x$2.err
  echo "groups $1 x $2 resident: $(python -c "import json;d=json.load(open('gpurun_out/r4a_g$1x$2.json'));print(round(d['reg_per_s'],1), d['identical_to_single'], round(d['busy_threads'],2), round(d['cpu_ms_per_registration'],2), d['ok'], d['of'])" 2>&1)"
done
for cfg in "8 1" "4 2" "8 2"; do
  set -- $cfg
  timeout 600 python tools/exp_groups.py 512 $1 $2 1 > gpurun_out/r4a_h$1x$2.json 2> gpurun_out/r4a_h$1x$2.err
  echo "groups $1 x $2 host: $(python -c "import json;d=json.load(open('gpurun_out/r4a_h$1x$2.json'));print(round(d['reg_per_s'],1), d['identical_to_single'], round(d['busy_threads'],2), round(d['cpu_ms_per_registration'],2), d['ok'], d['of'])" 2>&1)"
done
