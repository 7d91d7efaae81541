cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_seams.py tests/test_gpu_golden.py tests/test_gpu_edge.py tests/test_gpu_faithful.py tests/test_gpu_cli.py tests/test_gpu_ransac.py "tests/test_gpu_properties.py::test_k1_full_size_lists_equal_numpy_restatement" "tests/test_gpu_properties.py::test_k1_full_size_subset_counts_equal_numpy_restatement" -m gpu -x -q > $O/t1.log 2>&1; echo "pytest rc $?" ; tail -5 $O/t1.log
python tools/exp_throughput.py 256 8 > $O/exp_base.json 2> $O/exp_base.err; cat $O/exp_base.json
PLADE_NO_SPECULATION=1 python tools/exp_throughput.py 256 8 > $O/exp_nospec.json 2>> $O/exp_base.err; cat $O/exp_nospec.json
EXP_HOST=1 python tools/exp_throughput.py 256 8 > $O/exp_host.json 2>> $O/exp_base.err; cat $O/exp_host.json
