#!/bin/bash
# Same-box comparison of the arithmetic modes on the headline workload (two interleaved rounds): steps/s, conv3x3 TF, live parity figure.
for rep in 1 2; do
  for prec in f16x3 f16mx; do
    python bench.py --steps 8 --warmup 2 --precision $prec --no-cpu-baseline --no-secondary --no-strict-fp32 --no-live-traffic > /tmp/ab.log 2>/dev/null
    echo "[$prec]"; python tools/bsum.py /tmp/ab.log | head -8
  done
done
