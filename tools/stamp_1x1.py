#!/usr/bin/env python3
"""Reads the 1x1 launches of a tools/stamp_probe.sh record: per wave group, cycles between consecutive stamps, and the timeline of waves 0 and 4.
Stamp ids of the 1x1 step: 2 step start, 41 next activations landed, 40 transform done / LDS writes start, 42 staged, 3 MFMAs issued, 7 weights landed, 8 barrier passed."""
import collections, sys
launches = []
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps_1x1.txt"):
    if line.startswith("launch"):
        launches.append({"hdr": line.strip(), "waves": []})
    elif line.startswith("wave"):
        toks = line.split(":", 1)[1].split()
        launches[-1]["waves"].append([(int(t.split(":")[0]), int(t.split(":")[1])) for t in toks])
seen = set()
for L in launches:
    if "taps 1" not in L["hdr"] or L["hdr"] in seen or not L["waves"][0]:
        continue
    seen.add(L["hdr"])
    print(L["hdr"])
    for grp, ws in (("waves0-3", L["waves"][:4]), ("waves4-7", L["waves"][4:])):
        acc = collections.OrderedDict(); tot = 0
        for w in ws:
            for (i0, t0), (i1, t1) in zip(w, w[1:]):
                d = (t1 - t0) & 0xFFFFFFFF; acc[(i0, i1)] = acc.get((i0, i1), 0) + d; tot += d
        print(" ", grp, tot / max(len(ws), 1))
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:12]:
            print("     ", k, round(v / len(ws)), round(100 * v / tot, 1))
    t00 = L["waves"][0][0][1]
    for w in (0, 4):
        print("   wave", w, " ".join(f"{i}@{(t - t00) & 0xFFFFFFFF}" for i, t in L["waves"][w][:50]))
