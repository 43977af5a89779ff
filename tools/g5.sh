cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
python -m pytest tests/test_gpu_nets.py tests/test_gpu_sizes.py tests/test_gpu_fullsize.py tests/test_gpu_samplers.py tests/test_gpu_f16mx.py -m gpu -x -q 2>&1 | tail -5
AB_LINES=2 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_noends.so - 2>&1 | grep -v strict
tools/ab_b1.sh drmnet_amd/csrc/_ab/libdrmnet_hip_noends.so - 2>&1
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g5/shapes.log 2>&1
grep "6->128" gpurun_out/g5/shapes.log | tail -1
