cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
python -m pytest tests/test_gpu_split.py tests/test_gpu_nets.py tests/test_gpu_f16mx.py -m gpu -x -q 2>&1 | tail -3
AB_LINES=5 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_nowide.so - 2>&1
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g2/shapes.log 2>&1
