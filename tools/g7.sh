cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py tests/test_gpu_nets.py tests/test_gpu_f16mx.py tests/test_gpu_ops.py tests/test_gpu_sizes.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -5
AB_LINES=4 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so - 2>&1 | grep -v strict
tools/ab_b1.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so - 2>&1
