cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_race.py -x -q -k "forked" 2>&1 | tail -5
