#!/bin/bash
# the round-end checks as the driver runs them: GPU tests, smoke, the bench line
O=gpurun_out; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/full_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/full_bench.json 2> $O/full_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/full_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','steps','registrations_ok','results_bit_identical_to_the_pair_alone_rank0')})
print('resident', d['resident_rank0']['value'], 'host', d['host_rank0'])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','avg_launch_us','launches_per_step','step_frac_of_hbm_peak','traffic')})
print('cpu', d['cpu_baseline']['value'], 'batch', d['cpu_baseline_batch']['value'], 'cli', d['cli_end_to_end']['value'], d['cli_end_to_end']['seconds'])
PY
