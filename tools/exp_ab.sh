#!/bin/bash
# A/B of environment switches on one box: every line of VARIANTS is "name ENV=... ENV=..." ; BENCH_ARGS extra bench flags
run() { name=$1; shift; env "$@" python bench.py --pairs 16 --no-cpu-baseline --no-cli --no-default-mode --closed-form-steps 0 --resident-steps 0 --profiled-steps 0 --no-parity $BENCH_ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name value', round(d['value'],1), d['results_bit_identical_to_the_pair_alone_rank0'], 'cpu ms', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2), 'threads', round(d['host_rank0']['busy_host_threads_avg'],2))
    elif 'registrations executed' not in l: print(l.rstrip()[:300])
"; }
for rep in 1 2; do
run base X=1
run ov_wgs256 PLADE_EXP_OV_WGS=256
run ov_wgs512 PLADE_EXP_OV_WGS=512
run poll50 PLADE_EXP_POLL=1
BENCH_ARGS="--group 4 --inflight 8" run g4x8 X=1
BENCH_ARGS="--group 4 --inflight 6" run g4x6 X=1
done
