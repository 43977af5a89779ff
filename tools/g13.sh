cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g13
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/g13/prof -o r05 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-strict-fp32 --no-secondary --no-live-traffic > $GRAFT_REPO_ROOT/gpurun_out/g13/bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/g13/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/g13/kernel_stats.csv
rm -rf gpurun_out/g13/prof
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 > gpurun_out/g13/shapes.log 2>&1
