cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_sizes.py tests/test_gpu_race.py -x -q 2>&1 | tail -3
tools/ab_b1.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so -
AB_LINES=1 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so -
