cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
python -m pytest tests/test_gpu_split.py tests/test_gpu_nets.py -m gpu -x -q 2>&1 | tail -3
AB_LINES=4 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_w3only.so - 2>&1 | grep -v strict
tools/ab_b1.sh drmnet_amd/csrc/_ab/libdrmnet_hip_w3only.so - 2>&1
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g3/shapes.log 2>&1
