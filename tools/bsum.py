#!/usr/bin/env python3
"""Prints the headline numbers of a bench.py log (last JSON line): value, ms/step, dominant-kernel figures, per-family ms per step."""
import json
import sys

line = [x for x in open(sys.argv[1]) if x.startswith('{"metric"')][-1]
d = json.loads(line)
r = d.get("roofline") or {}
print(f"{d['value']} {d['unit']}  {d['ms_per_step']} ms/step  conv3x3 {r.get('achieved')} TF  avg launch {r.get('avg_launch_ms')} ms  frac {r.get('frac')}")
for k, v in (d.get("kernel_breakdown") or {}).items():
    print(f"  {k:24s} {v['ms'] / d['steps']:8.3f} ms/step  {v['launches'] // d['steps']:4d} launches  {v['tflops']} TF")
for k in ("strict_fp32", "parity_check"):
    if k in d:
        print(" ", k, json.dumps(d[k])[:300])
