cd $GRAFT_REPO_ROOT
O=gpurun_out
for h in 2 1 0; do
EXP_HOST=$h python tools/exp_throughput.py 512 8 > $O/exp_h.json 2> $O/exp_h.err; python -c "
import json; d=json.load(open('gpurun_out/exp_h.json')); d.pop('stats'); print(d)"
done
python -m pytest tests/test_gpu_ransac.py::test_batch_mode_prefetch_returns_the_bits_of_the_plain_call tests/test_gpu_edge.py -m gpu -x -q 2>&1 | tail -3
