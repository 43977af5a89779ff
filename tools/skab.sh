#!/bin/bash
# same-box comparison of the split-K hand-off forms: the product (sc1 accesses, no fences) against `tools/build_variant.sh skfenced "-DDRM_SK_FENCED=1"`
# (plain accesses around agent-scope release / acquire fences): the split / op parity tests, the race screen and the batch-1 step
#   usage: tools/skab.sh - drmnet_amd/csrc/_ab/libdrmnet_hip_skfenced.so
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$PWD/$lib"; fi
  echo "== $lib"
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_split.py tests/test_gpu_race.py -q 2>&1 | tail -n 2
  python bench.py --batch 1 --height 128 --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-check --no-secondary --no-strict-fp32 --no-profile 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1:', d['ms_per_step'], 'ms per step')"
done
