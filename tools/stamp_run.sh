#!/bin/bash
# one stamped timeline of the top-level 3x3 conv from its third tile on (tools/stamp_probe.sh + the raw stamps of waves 0 and 4)
cd $GRAFT_REPO_ROOT
export DRM_S2_STAMP_TILE0=2
LAYER_SHAPES="32,128,128,128,256" tools/stamp_probe.sh gpurun_out/stamps_top.txt f16mx
