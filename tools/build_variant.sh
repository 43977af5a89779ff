#!/bin/bash
# Builds an experimental variant of the library for a same-box A/B (tools/ab_bench.sh): conv_split2.hip (and any other source named in
# VARIANT_SRCS, comma-separated stems or "all") recompiled with extra flags, linked with the product's other objects into
# drmnet_amd/csrc/_ab/libdrmnet_hip_<name>.so.  VARIANT_UNITS (e.g. "92,93") restricts the conv_split2 kernel units that are recompiled.
#   usage: tools/build_variant.sh <name> "<extra hipcc flags>"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
python -m drmnet_amd.build --variant "$1" --flags="$2" --srcs "${VARIANT_SRCS:-conv_split2}" ${VARIANT_UNITS:+--units "$VARIANT_UNITS"}
