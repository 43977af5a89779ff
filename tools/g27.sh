cd $GRAFT_REPO_ROOT
DRM_LIB_PATH=$GRAFT_REPO_ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_foldall.so timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_configs.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -3
AB_LINES=1 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so drmnet_amd/csrc/_ab/libdrmnet_hip_foldall.so -
