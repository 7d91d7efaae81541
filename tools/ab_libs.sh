#!/bin/bash
# A/B of library builds on ONE box: gpurun_ab/<name>.so for every name in AB_LIBS, alternating, AB_REPS times;
# the bench's timed leg only (host clouds, AB_PAIRS distinct pairs, default mode unless BENCH_ARGS says otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp plade_amd/libplade_hip.so gpurun_ab/.current.so
for rep in $(seq 1 ${AB_REPS:-3}); do for v in ${AB_LIBS:-old new}; do
  cp gpurun_ab/$v.so plade_amd/libplade_hip.so
  python bench.py --pairs ${AB_PAIRS:-32} --no-cpu-baseline --no-cli --no-default-mode --closed-form-steps ${AB_CLOSED:-0} --resident-steps 0 --profiled-steps 0 --no-parity $BENCH_ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d.get('closed_form_mode_rank0'); print('$v value', round(d['value'],1), 'closed', c and round(c['value'],1), d['results_bit_identical_to_the_pair_alone_rank0'], 'cpu ms', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2), 'threads', round(d['host_rank0']['busy_host_threads_avg'],2), 'alone ms', d['value_by_seed']['pair_alone_latency_ms'])
    elif 'registrations executed' not in l: print(l.rstrip()[:300])
"
done; done
cp gpurun_ab/.current.so plade_amd/libplade_hip.so
