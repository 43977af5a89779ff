#!/bin/bash
# Timing ablations of the conv epilogue on the diagnostic build (-DDRM_S2_STAMP; DRM_S2_FLAGS selects what is left out: the numbers are wrong,
# the instruction stream otherwise identical): 65536 no output stores, 4194304 no transposes + stores, 2097152 no residual loads, 1048576 no
# statistics arithmetic, 524288 no LDS statistics atomics, 262144 statistics fold by plain stores, 131072 no epilogue at all.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
export DRM_LIB_PATH="$ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_stamp.so"
for rep in 1 2; do
for f in 0 65536 4194304 2097152 1048576 524288 1572864 262144 131072; do
  echo "[flags $f]"; DRM_S2_FLAGS=$f python3 tools/layer_probe.py f16mx 2>&1 | grep resblock
done; done
