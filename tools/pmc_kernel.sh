#!/bin/bash
# SQ / GRBM counters of ONE kernel (name substring) under a python command: MFMA-busy fraction, wave wait / issue-stall / active fractions, instruction
# mix, LDS conflicts and the clock the chip holds.  Counters are collected in their own passes with --kernel-trace only, as the GPU pool requires.
#   usage (GPU box): tools/pmc_kernel.sh <out.json> <kernel substring> <python script> [args...]
#   e.g.             tools/pmc_kernel.sh gpurun_out/r04_pmc_attn_flash.json attn_flash_kernel tools/attn_bench.py 32 32 64 f16x3
out=$1; kern=$2; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TMPD=/tmp/pmc_kernel; rm -rf $TMPD; mkdir -p $TMPD
SCRIPT="$ROOT/$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace -d $TMPD/p1 -- python3 "$SCRIPT" "$@" > $TMPD/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace -d $TMPD/p2 -- python3 "$SCRIPT" "$@" > $TMPD/p2.log 2>&1
cd "$ROOT"
python3 - "$out" "$kern" "$SCRIPT $*" <<'PY'
import collections, glob, json, sqlite3, sys
out, kern, cmd = sys.argv[1:4]
res = {"command": f"tools/pmc_kernel.sh: rocprofv3 --pmc <SQ / GRBM counters> --kernel-trace -- python3 {cmd} (two passes)", "kernel_filter": kern,
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration",
       "kernels": {}}
for sub in ("p1", "p2"):
    dbs = sorted(glob.glob(f'/tmp/pmc_kernel/{sub}/**/*_results.db', recursive=True))
    if not dbs:
        continue
    con = sqlite3.connect(dbs[-1])
    per, meta = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    for disp, name, cname, val, dur in con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
        if kern in name:
            per[disp][cname] += val
            meta[disp] = (name.split('(')[0], dur)
    order = sorted(per)[len(per) // 3:]  # drop warm-up launches
    if not order:
        continue
    name = meta[order[0]][0]
    avg = collections.defaultdict(float)
    for d in order:
        for k, v in per[d].items():
            avg[k] += v / len(order)
        avg["duration_ns"] += meta[d][1] / len(order)
    e = res["kernels"].setdefault(name, {})
    e[f"{sub}_launches_averaged"] = len(order)
    e[f"{sub}_duration_us"] = round(avg["duration_ns"] / 1e3, 1)
    gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
    if gui:
        e[f"{sub}_clock_ghz"] = round(gui / 8 / avg["duration_ns"], 3)
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    if sub == "p1" and wc and gui:
        e["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui / 8), 3)
        e["wave_wait_any"] = round(avg["SQ_WAIT_ANY"] / wc, 3)
        e["wave_wait_inst_any"] = round(avg["SQ_WAIT_INST_ANY"] / wc, 3)
        e["wave_active_inst"] = round(avg["SQ_ACTIVE_INST_ANY"] / wc, 3)
        e["lds_conflict_frac"] = round(avg["SQ_LDS_BANK_CONFLICT"] / max(avg["SQ_LDS_IDX_ACTIVE"], 1), 3)
    for k, v in avg.items():
        if k != "duration_ns":
            e.setdefault("raw", {})[k] = round(v, 1)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res["kernels"], indent=1)[:2500])
PY
