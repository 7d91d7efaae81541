#!/bin/bash
# host-wait polling experiment: throughput and busy host threads vs poll interval / query frequency
run() {
python bench.py --steps 384 --warmup 5 --no-cpu-baseline --no-default-mode --resident-steps 0 --profiled-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'busy', round(d['host_rank0']['busy_host_threads_avg'],2), 'cpu ms/step', round(d['host_rank0']['cpu_seconds_per_step']*1e3,2))"
}
run base
PLADE_POLL_QUERY_MASK=15 run qmask15
PLADE_POLL_QUERY_MASK=15 PLADE_POLL_NS=30000,80000 run q15_30_80
PLADE_POLL_QUERY_MASK=15 PLADE_POLL_NS=50000,100000 run q15_50_100
PLADE_POLL_QUERY_MASK=15 PLADE_POLL_NS=20000,150000 run q15_20_150
PLADE_POLL_NS=50000,100000 run q0_50_100
