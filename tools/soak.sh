#!/bin/bash
# soak: 40 000 registrations through the timed pipeline; HBM in use sampled while it runs (flat = no leak), all results compared
( for i in $(seq 1 12); do sleep 6; rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; done ) > gpurun_out/soak_mem.log &
python bench.py --pairs 16 --steps ${STEPS:-40000} --no-cpu-baseline --no-cli --no-default-mode --closed-form-steps 0 --resident-steps 0 --profiled-steps 0 --no-parity 2> gpurun_out/soak.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('soak steps', d['steps'], 'value', round(d['value'],1), 'identical', d['results_bit_identical_to_the_pair_alone_rank0'], 'ok', d['registrations_ok'], '/', d['registrations_timed'])
"
wait
cat gpurun_out/soak_mem.log
