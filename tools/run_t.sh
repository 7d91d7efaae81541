cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py -k "config4" -q -s 2>&1 | grep -E "configs\[4\]|passed|failed|^E  " | cut -c1-300 | tail -14
