#!/bin/bash
# Builds an experimental variant of the library for a same-box A/B (tools/ab_bench.sh): conv_split2.hip (and any other source named in
# VARIANT_SRCS) recompiled with extra flags, linked with the product's other objects into drmnet_amd/csrc/_ab/libdrmnet_hip_<name>.so.
#   usage: tools/build_variant.sh <name> "<extra hipcc flags>"      (run the product build first: python -m drmnet_amd.build)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; EXTRA="$2"
SRCS="${VARIANT_SRCS:-conv_split2}"
OBJ="$ROOT/drmnet_amd/csrc/_obj"; OUT="$ROOT/drmnet_amd/csrc/_ab"; TMP="/tmp/drm_variant_$NAME"
mkdir -p "$OUT" "$TMP"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off $EXTRA"
objs=""
for f in conv conv_split conv_split2 gn attn attn_flash misc refmap transform engine samplers abi profiler; do
  if [[ " $SRCS " == *" $f "* ]]; then
    hipcc $FLAGS ${VARIANT_REMARKS:+-Rpass-analysis=kernel-resource-usage} -c "$ROOT/drmnet_amd/csrc/$f.hip" -o "$TMP/$f.o" 2> "$TMP/$f.log" &
    objs="$objs $TMP/$f.o"
  else
    objs="$objs $OBJ/$f.o"
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libdrmnet_hip_$NAME.so" $objs
echo "built $OUT/libdrmnet_hip_$NAME.so"
