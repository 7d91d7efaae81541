cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_ransac.py::test_batch_mode_prefetch_returns_the_bits_of_the_plain_call tests/test_gpu_bench_world2.py -m gpu -x -q > $O/t2.log 2>&1; echo "pytest rc $?" ; tail -15 $O/t2.log
python bench.py --steps 20 --warmup 5 > $O/b20.json 2> $O/b20.err; echo rc $?; tail -3 $O/b20.err
python - <<'PY'
import json
for f in ("b20",):
    d=json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["host_buffers_rank0"], d["resident_rank0"], d["default_mode_rank0"], d["host_rank0"], d["cpu_baseline"]["value"])
PY
python bench.py --steps 512 --warmup 16 --no-cpu-baseline --no-default-mode > $O/b512.json 2> $O/b512.err; echo rc $?
python - <<'PY'
import json
d=json.load(open("gpurun_out/b512.json"))
print("b512", d["value"], d["ms_per_step"], d["host_buffers_rank0"], d["resident_rank0"], d["host_rank0"])
PY
