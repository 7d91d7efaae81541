cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "not config3 and not config4" 2>&1 | tail -5
