#!/bin/bash
# round 6: parity tests on a variant library, then tools/ab_bench.sh on the listed libraries ("-" = product); AB_TEST_LIB = the variant the tests run on
cd $GRAFT_REPO_ROOT
if [ -n "$AB_TEST_LIB" ]; then
  DRM_LIB_PATH="$GRAFT_REPO_ROOT/$AB_TEST_LIB" timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_fullsize.py tests/test_gpu_split.py -x -q 2>&1 | tail -3
fi
AB_LINES=${AB_LINES:-6} tools/ab_bench.sh "$@"
