cd $GRAFT_REPO_ROOT
for k in 20 512 20 512; do
python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-default-mode --resident-steps 128 --profiled-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench steps $k', round(d['value'],1), 'bracketed', round(d['host_buffers_rank0']['bracketed_value'],1), 'resident', round(d['resident_rank0']['value'],1), 'busy', round(d['host_rank0']['busy_host_threads_avg'],2))"
done
python -m pytest tests/test_gpu_bench_world2.py -q 2>&1 | tail -2
