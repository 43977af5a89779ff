set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 8 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g1/bench.log 2>&1
python tools/bsum.py gpurun_out/g1/bench.log
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g1/shapes.log 2>&1
tools/stamp_probe.sh gpurun_out/g1/stamps.txt f16mx
python tools/stamp_read.py gpurun_out/g1/stamps.txt > gpurun_out/g1/stamps_read.txt
