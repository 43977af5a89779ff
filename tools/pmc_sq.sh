#!/bin/bash
# SQ / GRBM counters of the shipped fused 3x3 conv on two ResBlock shapes (NHWC inside, realistic epilogues): MFMA-busy fraction,
# wave wait fractions, LDS bank conflicts and the clock the chip holds (GRBM_GUI_ACTIVE / 8 / duration).  Counters are collected
# in their own passes with --kernel-trace only (no --stats / sys-trace), as the GPU pool requires.
#   usage (on the GPU box): tools/pmc_sq.sh <tag>   -> gpurun_out/pmc_sq/<tag>_pmc_sq_conv.json   (copy it into profiles/)
tag=${1:-r02}
PREC=${2:-f16x3}   # f16x3 / f16mx / f16 / fp32 (every mode runs conv_split2_kernel<9,...> on these shapes)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_sq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/p1" -- python3 "$ROOT/tools/layer_probe.py" $PREC > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/p2" -- python3 "$ROOT/tools/layer_probe.py" $PREC > "$OUT/p2.log" 2>&1
cd "$ROOT"
python3 - "$tag" "$PREC" <<'PY'
import collections, glob, hashlib, json, sqlite3, sys
tag = sys.argv[1]
prec = sys.argv[2]
KERN = 'conv_split2_kernel<9'
OUT = 'gpurun_out/pmc_sq'
def rows(sub):
    dbs = sorted(glob.glob(f'{OUT}/{sub}/**/*_results.db', recursive=True))
    if not dbs:
        return []
    con = sqlite3.connect(dbs[-1])
    return con.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection").fetchall()
res = {"command": "tools/pmc_sq.sh (rocprofv3 --pmc <SQ/GRBM counters> --kernel-trace -- python3 tools/layer_probe.py " + prec + "; two passes)",
       "conv_split2_sha16": hashlib.sha256(open('drmnet_amd/csrc/conv_split2.hip', 'rb').read()).hexdigest()[:16],
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; "
                "clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
       "kernels": {}}
for sub in ("p1", "p2"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for disp, name, grid, cname, val, dur in rows(sub):
        if KERN not in name:
            continue
        per[disp][cname] += val
        meta[disp] = (name.split('(')[0], grid, dur)
    groups = collections.defaultdict(list)
    order = sorted(per)  # layer_probe.py runs shape 1 (4 ResBlock calls = 8 conv launches), then shape 2: split by dispatch order
    shapes = ["128->128 @128x256 B=32", "256->256 @64x128 B=32"]
    for pos, disp in enumerate(order):
        name, grid, dur = meta[disp]
        groups[(name + " " + shapes[min(pos * len(shapes) // max(len(order), 1), len(shapes) - 1)], grid)].append((per[disp], dur))
    for (name, grid), lst in groups.items():
        lst = lst[len(lst) // 3:]  # drop warm-up launches
        n = len(lst)
        avg = collections.defaultdict(float)
        for ctr, dur in lst:
            for k, v in ctr.items():
                avg[k] += v / n
            avg["duration_ns"] += dur / n
        key = f"{name} grid_threads={grid}"
        e = res["kernels"].setdefault(key, {"launches_averaged": n})
        e[f"pass_{sub}_duration_us"] = round(avg["duration_ns"] / 1e3, 1)
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
        if gui and avg["duration_ns"]:
            e[f"pass_{sub}_clock_GHz"] = round(gui / 8 / avg["duration_ns"], 3)
        if sub == "p1" and gui:
            wc = avg.get("SQ_WAVE_CYCLES", 0.0)
            e["mfma_busy_frac"] = round(avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * gui / 8), 3)
            if wc:
                e["wave_wait_any_frac"] = round(avg.get("SQ_WAIT_ANY", 0.0) / wc, 3)
                e["wave_wait_inst_frac"] = round(avg.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
                e["wave_active_inst_frac"] = round(avg.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
            li = avg.get("SQ_LDS_IDX_ACTIVE", 0.0)
            if li:
                e["lds_bank_conflict_frac_of_lds_cycles"] = round(avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / li, 4)
        for k, v in avg.items():
            if k not in ("duration_ns",):
                e.setdefault("raw", {})[k] = round(v, 1)
json.dump(res, open(f'{OUT}/{tag}_pmc_sq_conv{"" if prec == "f16x3" else "_" + prec}.json', 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf "$OUT/p1" "$OUT/p2"
