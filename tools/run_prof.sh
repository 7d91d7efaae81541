cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ransac.py tests/test_gpu_golden.py tests/test_gpu_seams.py -m gpu -x -q 2>&1 | tail -4
bash tools/prof_exp.sh 96 8
