#!/bin/bash
# Builds the diagnostic library (every source with -DDRM_S2_STAMP, into its own object directory and .so -- the product library is
# untouched) and records the s_memtime timeline of one workgroup of the fused 3x3 conv on a few layer shapes.
#   usage (on the GPU box): tools/stamp_probe.sh [out file]      -> gpurun_out/stamps.txt ; read it with tools/stamp_read.py
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
OUT="${1:-$ROOT/gpurun_out/stamps.txt}"
PREC="${2:-f16mx}"
: > "$OUT"
# (every source: -DDRM_S2_STAMP adds fields to ConvArgs; built here or beforehand in the build container -- the .so travels with gpurun)
[ -f drmnet_amd/csrc/_ab/libdrmnet_hip_stamp.so ] || python -m drmnet_amd.build --variant stamp --flags=-DDRM_S2_STAMP --srcs all
DRM_LIB_PATH="$ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_stamp.so" DRM_S2_STAMP_FILE="$OUT" python3 tools/layer_probe.py $PREC
echo "wrote $OUT"
