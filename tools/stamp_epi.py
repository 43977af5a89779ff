#!/usr/bin/env python3
"""Epilogue part of the stamp timeline (ids 9 .. 10) of waves 0 and 4, cycles since the wave's epilogue start."""
import sys
cur = None
for line in open(sys.argv[1]):
    if line.startswith("launch"):
        cur = line.strip(); print(cur)
    elif line.startswith("wave"):
        w = int(line.split(":")[0].split()[1])
        if w not in (0, 4): continue
        toks = [(int(t.split(":")[0]), int(t.split(":")[1])) for t in line.split(":", 1)[1].split()]
        ids = [i for i, _ in toks]
        if 9 not in ids: continue
        k = ids.index(9); t9 = toks[k][1]
        first = toks[0][1]
        out = []
        for i, t in toks[k:]:
            out.append(f"{i}@{(t - t9) & 0xFFFFFFFF}")
            if i == 10: break
        print(f"  wave {w}: tile {((t9 - first) & 0xFFFFFFFF)} cyc before epilogue | " + " ".join(out))
