cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest "tests/test_gpu_seams.py::test_plane_component_bitmap_larger_than_the_lds_labelling" tests/test_gpu_ransac.py::test_tall_scene_takes_the_global_memory_labelling_path -m gpu -q 2>&1 | grep -E "^E|passed|failed|Error" | head -30
for h in 0; do
EXP_HOST=$h python tools/exp_throughput.py ${1:-512} ${2:-8} > $O/exp_q$h.json 2> $O/exp_q$h.err; python -c "
import json; d=json.load(open('gpurun_out/exp_q$h.json')); st=d.pop('stats'); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k not in ('env','bg_copy_pairs','t_upload_take_submit_ms_avg')}, 'iters', st.get('ransac_iterations'))"
done
