#!/bin/bash
# batch parts of the DRMNet step at 128 rows per GPU: 1 / 2 / 3 / 4 row ranges on forked streams (drm_drmnet_set_batch_parts), same box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for p in 1 2 3 4; do
  DRM_BATCH_PARTS=$p DRM_BATCH_PART_MIN=32 python bench.py --batch 128 --steps 4 --warmup 1 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-profile --no-strict-fp32 > /tmp/ab.log 2>/dev/null
  echo "[B=128 parts $p] $(grep -o '"value": [0-9.]*' /tmp/ab.log | head -1)"
done
done
