cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nets.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
AB_LINES=1 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so -
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 2>&1 | grep "128->32" | tail -1
