#!/bin/bash
# same-box A/B of two library builds on the batch-1 step (the reference's own use) and the headline: tools/ab_b1.sh <lib.so> [<lib.so> ...]   ("-" = the product library)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$ROOT/$lib"; fi
    python bench.py --batch 1 --height 128 --width 128 --steps 40 --warmup 5 --no-cpu-baseline --no-parity-check --no-secondary --no-strict-fp32 --no-profile 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('[$lib] B=1 @128x128:', d['ms_per_step'], 'ms per step')"
  done
done
