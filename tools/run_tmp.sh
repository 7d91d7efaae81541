timeout 1500 python -m pytest tests/test_gpu_seams.py tests/test_gpu_ransac.py tests/test_gpu_groups.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
python tools/digest.py 2>&1 | tail -1
for h in 0 1; do timeout 600 python tools/exp_groups.py 1536 4 4 $h > gpurun_out/x$h.json 2>/dev/null; python -c "
import json
d=json.load(open('gpurun_out/x$h.json'))
print('host', $h, round(d['reg_per_s'],1), d['identical_to_single'], round(d['cpu_ms_per_registration'],2), round(d['busy_threads'],2))"; done
