timeout 900 python -m pytest tests/test_gpu_groups.py -x -q -m gpu 2>&1 | tail -3
PLADE_DEBUG_READS=1 timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_world2.py 2>&1 | tail -4
