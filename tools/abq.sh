#!/bin/bash
# quick same-box check of a kernel change: the conv / network parity tests, then tools/ab_bench.sh on the given libraries ("-" = product)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_sizes.py tests/test_gpu_race.py tests/test_gpu_f16mx.py -x -q 2>&1 | tail -3
AB_LINES=${AB_LINES:-2} tools/ab_bench.sh "$@"
