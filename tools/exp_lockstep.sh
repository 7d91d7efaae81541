#!/bin/bash
# lock step: hardware queues x groups in flight (resident legs off, 16 distinct pairs)
for q in ${QUEUES:-4 8}; do for m in ${INFLIGHT:-4 6 8}; do
GPU_MAX_HW_QUEUES=$q python bench.py --pairs 16 --inflight $m --no-cpu-baseline --no-cli --no-default-mode --closed-form-steps 0 --resident-steps 0 --profiled-steps 0 --no-parity ${BENCH_EXTRA:-} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('queues $q inflight $m value', round(d['value'],1), d['results_bit_identical_to_the_pair_alone_rank0'], 'cpu ms', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2), 'threads', round(d['host_rank0']['busy_host_threads_avg'],2))
    elif 'registrations executed' not in l: print(l.rstrip()[:300])
"
done; done
