#!/usr/bin/env python3
"""Per-shape comparison of two DRM_PROF_DUMP logs (bench.py stderr): ms per launch of every 3x3 / 1x1 shape, and the per-step totals."""
import re, sys, collections
def parse(path):
    txt = open(path).read()
    last = re.split(r'\[drm profile\] \d+ shapes, [\d.]+ ms total\n', txt)[-1]
    d = collections.OrderedDict()
    for l in last.split('\n'):
        m = re.match(r'\[drm profile\] kind (\d) N(\d+)\s+(\d+)x(\d+)\s+(\d+)->(\d+)\s+:\s+(\d+) launches\s+([\d.]+) ms total', l)
        if m:
            d[(int(m.group(1)), m.group(3) + 'x' + m.group(4), m.group(5) + '->' + m.group(6))] = (int(m.group(7)), float(m.group(8)))
    return d
a, b = parse(sys.argv[1]), parse(sys.argv[2])
ta = tb = 0.0
for k in a:
    if k in b:
        na, xa = a[k]; nb, xb = b[k]
        ta += xa; tb += xb
        if abs(xa - xb) > 0.02 * max(xa, xb) and max(xa, xb) > 0.3:
            print(f"kind {k[0]} {k[1]:>8} {k[2]:>11}: {xa / na:.4f} -> {xb / nb:.4f} ms/launch  ({na} launches, total {xa:.3f} -> {xb:.3f})")
print(f"sum over common shapes: {ta:.3f} -> {tb:.3f} ms")
