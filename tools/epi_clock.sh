#!/bin/bash
# Clock and MFMA-busy of the fused 3x3 conv (diagnostic build) as shipped and with the epilogue removed (DRM_S2_FLAGS=131072): is the time the
# kernel saves by skipping a phase given back by the clock?   -> gpurun_out/pmc_sq/{epi_shipped,epi_none}_pmc_sq_conv.json
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
bash tools/stamp_probe.sh > /dev/null 2>&1   # always rebuilt: a box may be reused with a stale /tmp
export DRM_LIB_PATH=/tmp/libdrmnet_hip_stamp.so
mkdir -p gpurun_out/epi_clock
bash tools/pmc_sq.sh epi_shipped > /dev/null 2>&1; cp gpurun_out/pmc_sq/epi_shipped_pmc_sq_conv.json gpurun_out/epi_clock/
DRM_S2_FLAGS=131072 bash tools/pmc_sq.sh epi_none > /dev/null 2>&1; cp gpurun_out/pmc_sq/epi_none_pmc_sq_conv.json gpurun_out/epi_clock/
python3 - <<'PY'
import json
for tag in ("epi_shipped", "epi_none"):
    d = json.load(open(f"gpurun_out/epi_clock/{tag}_pmc_sq_conv.json"))
    for k, v in d["kernels"].items():
        print(tag, k.split(">")[-1][:28], {kk: vv for kk, vv in v.items() if kk != "raw"}, "MFMA busy cycles", v.get("raw", {}).get("SQ_VALU_MFMA_BUSY_CYCLES"))
PY
