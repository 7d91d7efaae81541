#!/bin/bash
run() {
python bench.py --steps 384 --warmup 5 --no-cpu-baseline --no-default-mode --resident-steps 0 --profiled-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'busy', round(d['host_rank0']['busy_host_threads_avg'],2), 'cpu ms/step', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2))"
}
run base_50_100
PLADE_POLL_NS=100000,200000 run p100_200
PLADE_POLL_NS=100000,300000 run p100_300
PLADE_POLL_NS=50000,100000 run base_again
PLADE_POLL_NS=150000,150000 run p150
