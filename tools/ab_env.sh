#!/bin/bash
# A/B of environment switches of ONE build on one box: AB_VARIANTS="name:ENV=1 name2:X=1,Y=2", alternating, AB_REPS times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { name=$1; shift; env "$@" python bench.py --pairs ${AB_PAIRS:-32} --no-cpu-baseline --no-cli --no-default-mode --closed-form-steps 0 --resident-steps 0 --profiled-steps 0 --no-parity $BENCH_ARGS 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name value', round(d['value'],1), d['results_bit_identical_to_the_pair_alone_rank0'], 'cpu ms', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2), 'threads', round(d['host_rank0']['busy_host_threads_avg'],2))
    elif 'registrations executed' not in l: print(l.rstrip()[:300])
"; }
for rep in $(seq 1 ${AB_REPS:-3}); do for v in ${AB_VARIANTS}; do
  name=${v%%:*}; envs=${v#*:}; run $name ${envs//,/ }
done; done
