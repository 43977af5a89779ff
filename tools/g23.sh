cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_samplers.py -x -q -s -k "forked" 2>&1 | tail -12
