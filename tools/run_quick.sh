# quick GPU check: the RANSAC / registration parity tests + steady-state throughput (resident and host clouds)
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_ransac.py tests/test_gpu_golden.py tests/test_gpu_registration.py tests/test_gpu_seams.py tests/test_gpu_faithful.py tests/test_gpu_edge.py "tests/test_gpu_properties.py::test_registration_full_size_every_intermediate_equals_oracle" -m gpu -x -q 2>&1 | tail -4
for h in 0 2; do
EXP_HOST=$h python tools/exp_throughput.py ${1:-512} ${2:-8} > $O/exp_q$h.json 2> $O/exp_q$h.err; python -c "
import json; d=json.load(open('gpurun_out/exp_q$h.json')); st=d.pop('stats'); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k not in ('env','bg_copy_pairs','t_upload_take_submit_ms_avg')}, 'iters', st.get('ransac_iterations'))"
done
