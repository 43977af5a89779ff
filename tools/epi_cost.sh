#!/bin/bash
# What the epilogue of the fused 3x3 conv costs: the diagnostic build (-DDRM_S2_STAMP) timed on two ResBlock shapes as shipped, without the
# output stores (DRM_S2_FLAGS=65536) and without the whole epilogue (DRM_S2_FLAGS=131072).  No timeline is recorded (no stamp file).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
bash tools/stamp_probe.sh > /dev/null 2>&1   # always rebuilt: a box may be reused with a stale /tmp
export DRM_LIB_PATH=/tmp/libdrmnet_hip_stamp.so
for rep in 1 2; do
  echo "as shipped:";        python3 tools/layer_probe.py f16x3 2>/dev/null | tail -2
  echo "no output stores:";  DRM_S2_FLAGS=65536 python3 tools/layer_probe.py f16x3 2>/dev/null | tail -2
  echo "no epilogue:";       DRM_S2_FLAGS=131072 python3 tools/layer_probe.py f16x3 2>/dev/null | tail -2
  echo "statistics fold without atomics (plain stores, wrong sums):"; DRM_S2_FLAGS=262144 python3 tools/layer_probe.py f16x3 2>/dev/null | tail -2
done
