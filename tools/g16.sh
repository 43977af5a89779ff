cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
tools/prof_round.sh r05 f16mx > gpurun_out/final/prof_round.log 2>&1
cp gpurun_out/prof_round/* gpurun_out/final/ 2>/dev/null
tools/pmc_sq.sh r05 f16mx > gpurun_out/final/pmc_sq.log 2>&1
cp gpurun_out/pmc_sq/*.json gpurun_out/final/ 2>/dev/null
DRM_PROF_DUMP=1 python bench.py --steps 4 --warmup 2 --precision f16mx --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-strict-fp32 2>&1 | grep "drm profile" > gpurun_out/final/r05_shapes.txt
python tools/race_screen.py 25 > gpurun_out/final/r05_race_screen.txt 2>&1
tail -2 gpurun_out/final/r05_race_screen.txt
