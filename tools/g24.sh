cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/gputests.txt
cat gpurun_out/gputests.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json
