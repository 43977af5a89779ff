#!/bin/bash
# A/B of library builds inside ONE gpurun call (device-to-device spread is ~5 %): tools/ab_bench.sh <lib.so> [<lib.so> ...]
# "-" = the product library.  Two interleaved rounds; prints steps/s and the per-family ms per step.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$ROOT/$lib"; fi
    python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic > /tmp/ab.log 2>/dev/null
    echo "[$lib]"; python tools/bsum.py /tmp/ab.log | head -${AB_LINES:-3}
  done
done
