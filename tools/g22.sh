cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for p in 1 2 4; do
  DRM_BATCH_PARTS=$p python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-profile --no-strict-fp32 > /tmp/ab.log 2>/dev/null
  echo "[parts $p]"; python - <<'PY'
import json
for l in open('/tmp/ab.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('f16x3',{}).get('value') if isinstance(d.get('f16x3'),dict) else d.get('f16x3'))
PY
done
done
for p in 1 2; do
  DRM_BATCH_PARTS=$p python bench.py --batch 128 --steps 4 --warmup 1 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic --no-profile --no-strict-fp32 > /tmp/ab.log 2>/dev/null
  echo "[B=128 parts $p]"; grep -o '"value": [0-9.]*' /tmp/ab.log | head -1
done
