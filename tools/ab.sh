#!/bin/bash
# A/B of kernel variants inside one gpurun call (device-to-device variance is ~5 %): tools/ab.sh "ENV1=.. ENV2=.." "ENV.." ...
# an argument "-" means no extra environment
for rep in 1 2; do
  for cfg in "$@"; do
    [ "$cfg" = "-" ] && cfg=""
    env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%s]' % '$cfg', d['value'], d['ms_per_step'])"
  done
done
