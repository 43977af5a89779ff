#!/bin/bash
# Builds and runs the Winograd F(2x2,3x3) staging-pipeline sizing probe on the GPU box: tools/winograd_probe.sh  (-> gpurun_out/winograd_probe.txt)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off "$ROOT/tools/winograd_probe.hip" -o /tmp/winograd_probe && /tmp/winograd_probe | tee "$ROOT/gpurun_out/winograd_probe.txt"
