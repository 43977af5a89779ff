#!/usr/bin/env python3
"""Reads the timeline written by tools/stamp_probe.sh: per launch, where one workgroup's waves spend their cycles.
Stamp ids (epilogue blocks: 20+4b start, 21+4b values ready, 22+4b stores issued, 23+4b statistics issued -- tools/stamp_epi.py): 1 tile start, 2 step start, 3 step's MFMAs/requests issued, 4 chunk-end barrier 1, 5 activation loads landed, 6 staged,
7 weights landed, 8 barrier passed, 9 epilogue start, 10 tile end."""
import collections
import sys

NAMES = {(2, 3): "mfma+issue", (3, 7): "wait weights", (7, 8): "barrier", (3, 4): "barrier(chunk end 1)", (4, 5): "wait act loads", (5, 6): "stage A",
         (6, 7): "wait weights(after stage)", (8, 2): "loop", (8, 9): "loop", (9, 10): "epilogue+fold", (10, 1): "tile turn", (1, 2): "tile head"}


def main(path):
    launches = []
    for line in open(path):
        if line.startswith("launch"):
            launches.append({"hdr": line.strip(), "waves": []})
        elif line.startswith("wave"):
            toks = line.split(":", 1)[1].split()
            launches[-1]["waves"].append([(int(t.split(":")[0]), int(t.split(":")[1])) for t in toks])
    seen = set()
    for L in launches:
        if L["hdr"] in seen or not L["waves"] or not L["waves"][0]:
            continue
        seen.add(L["hdr"])
        print("=" * 120)
        print(L["hdr"])
        for grp, ws in (("waves 0-3", L["waves"][:4]), ("waves 4-7", L["waves"][4:])):
            acc = collections.OrderedDict()
            total = 0
            for w in ws:
                for (i0, t0), (i1, t1) in zip(w, w[1:]):
                    d = (t1 - t0) & 0xFFFFFFFF
                    acc[(i0, i1)] = acc.get((i0, i1), 0) + d
                    total += d
            print(f"  {grp}: {total / max(len(ws), 1):.0f} cycles recorded per wave")
            for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
                print(f"      {NAMES.get(k, str(k)):28s} {v / len(ws):9.0f}  {100.0 * v / total:5.1f} %")
        # wall-clock picture of the first recorded tile: when does each wave pass each stamp, relative to wave 0's first stamp
        t00 = L["waves"][0][0][1]
        print("  timeline (cycles since wave 0's first stamp), waves 0 and 4, first 40 stamps:")
        for w in (0, 4):
            print(f"    wave {w}: " + " ".join(f"{i}@{(t - t00) & 0xFFFFFFFF}" for i, t in L["waves"][w][:40]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.txt")
