#!/bin/bash
# Builds the diagnostic library (every source with -DDRM_S2_STAMP, into its own object directory and .so -- the product library is
# untouched) and records the s_memtime timeline of one workgroup of the fused 3x3 conv on a few layer shapes.
#   usage (on the GPU box): tools/stamp_probe.sh [out file]      -> gpurun_out/stamps.txt ; read it with tools/stamp_read.py
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out /tmp/drm_stamp_obj
OUT="${1:-$ROOT/gpurun_out/stamps.txt}"
: > "$OUT"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -DDRM_S2_STAMP"
objs=""
for f in conv conv_split conv_split2 gn attn misc refmap transform engine samplers abi profiler; do
  hipcc $FLAGS -c drmnet_amd/csrc/$f.hip -o /tmp/drm_stamp_obj/$f.o &
  objs="$objs /tmp/drm_stamp_obj/$f.o"
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libdrmnet_hip_stamp.so $objs
DRM_LIB_PATH=/tmp/libdrmnet_hip_stamp.so DRM_S2_STAMP_FILE="$OUT" python3 tools/layer_probe.py f16x3
echo "wrote $OUT"
