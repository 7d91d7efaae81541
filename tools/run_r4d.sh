#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r4d_bench_driver.json 2> $O/r4d_bench_driver.err; echo "rc=$?"
tail -5 $O/r4d_bench_driver.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d_bench_driver.json'))
print({k:d[k] for k in ('value','ms_per_step','steps','requested_steps','registrations_ok','results_bit_identical_to_the_pair_alone_rank0','max_frobenius_vs_ground_truth_rank0','single_registration_latency_ms')})
print(d['host_rank0']); print(d['resident_rank0']['value'], d['host_buffers_rank0']['bracketed_value'], d['pipeline']['occupancy_value'])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','avg_launch_us','launches_per_step','algorithmic_bytes_per_launch','step_frac_of_hbm_peak','step_algorithmic_bytes')})
print(d['cpu_baseline']); print(d['config']['inflight_for_local_world_8'])
PY
timeout 600 python -m pytest tests/test_gpu_bench_world2.py -x -q -m gpu 2>&1 | tail -3
for g in 2 3; do
  timeout 600 python tools/exp_groups.py 512 $g 4 1 > $O/r4d_h${g}x4.json 2> $O/r4d_h${g}x4.err
  python -c "
import json;d=json.load(open('$O/r4d_h${g}x4.json'));print('groups $g x 4 host:', round(d['reg_per_s'],1), 'busy', round(d['busy_threads'],2), 'cpu/reg', round(d['cpu_ms_per_registration'],2))"
done
