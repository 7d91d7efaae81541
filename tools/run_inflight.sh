#!/bin/bash
for m in 6 8 10 12; do
python bench.py --steps 384 --warmup 5 --no-cpu-baseline --no-default-mode --resident-steps 0 --profiled-steps 1 --inflight $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inflight $m', round(d['value'],1), 'busy', round(d['host_rank0']['busy_host_threads_avg'],2))"
done
