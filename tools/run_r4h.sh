#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_cli.py tests/test_gpu_groups.py tests/test_gpu_properties.py -x -q -m gpu -s 2>&1 | grep -E "A6 at 1M|passed|failed|Error|error" | tail -12
timeout 900 python bench.py > $O/r4h_bench.json 2> $O/r4h_bench.err; echo "bench rc=$?"; tail -3 $O/r4h_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','steps','requested_steps','registrations_ok','results_bit_identical_to_the_pair_alone_rank0')})
print('cpu', d['cpu_baseline']['value'], 'cpu_batch', d['cpu_baseline_batch'], )
print('cli', d['cli_end_to_end'])
print('speedup', d.get('speedup_vs_cpu_baseline'))
PY
