cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "auto" 2>&1 | tail -6
