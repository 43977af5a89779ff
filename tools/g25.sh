cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1 -o b1 -- python3 bench.py --batch 1 --height 128 --width 128 --steps 12 --warmup 3 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-profile --no-strict-fp32 > gpurun_out/b1/bench.log 2>&1
f=$(find gpurun_out/b1 -name "*kernel_trace.csv" | head -1)
python3 tools/step_trace.py $f g 70 > gpurun_out/b1_step_kernels_by_grid.txt
python3 tools/step_trace.py $f > gpurun_out/b1_step_kernels.txt
rm -rf gpurun_out/b1/*/
tail -2 gpurun_out/b1/bench.log | cut -c1-300
