cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nets.py tests/test_gpu_sizes.py tests/test_gpu_f16mx.py tests/test_gpu_fullsize.py tests/test_gpu_configs34.py -m gpu -x -q 2>&1 | tail -3
AB_LINES=1 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_nopool.so -
tools/ab_b1.sh drmnet_amd/csrc/_ab/libdrmnet_hip_nopool.so -
