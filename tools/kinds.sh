#!/bin/bash
# prints per-kind totals (ms per bench step) for the given env settings: tools/kinds.sh "ENV=.." ...
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg DRM_PROF_DUMP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys,re,collections
t=collections.defaultdict(float)
val=None
for l in sys.stdin:
    m=re.search(r'kind (\d+) .*launches\s+([\d.]+) ms total',l)
    if m: t[int(m[1])]+=float(m[2])
    if l.startswith('{'):
        import json; val=json.loads(l)['ms_per_step']
print('[$cfg]', 'ms/step', val, ' '.join(f'k{k}={v/3:.2f}' for k,v in sorted(t.items())))
"
done
