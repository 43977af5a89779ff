#!/bin/bash
# ObsNet forward (B = 32 @3x128x256: the T = 2048 attention path) under rocprofv3: kernel stats + the two HBM-traffic PMC passes.
#   usage (GPU box): tools/prof_obsnet.sh <tag> [precision]   -> gpurun_out/prof_obsnet/<tag>_obsnet_<precision>_kernel_stats.csv,
#                    <tag>_obsnet_<precision>_pmc_hbm_traffic.json, <tag>_obsnet_<precision>_bench.json   (precision defaults to f16mx: what bench.py's default, `auto`, resolves to on the synthetic weights)
tag=${1:-r04}
prec=${2:-f16mx}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_obsnet"
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--workload obsnet --precision $prec --no-cpu-baseline --no-parity-check --no-strict-fp32 --no-secondary"
cd "$ROOT"
python3 bench.py $ARGS --steps 10 --warmup 3 > $OUT/${tag}_obsnet_${prec}_bench.log 2>&1
grep "^{\"metric\"" $OUT/${tag}_obsnet_${prec}_bench.log | tail -1 > $OUT/${tag}_obsnet_${prec}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $ROOT/bench.py $ARGS --steps 5 --warmup 2 --no-profile > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-profile > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-profile > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 - "$tag" "$prec" <<'PY'
import sqlite3, glob, json, sys, csv, collections, os
tag, prec = sys.argv[1], sys.argv[2]
OUT = 'gpurun_out/prof_obsnet'
def db(sub):
    return sqlite3.connect(max(glob.glob(f'{OUT}/{sub}/**/*_results.db', recursive=True), key=os.path.getmtime))
rows = db('stats').execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
with open(f'{OUT}/{tag}_obsnet_{prec}_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
    for r in rows: w.writerow(r)
def counter(sub, name):
    con = db(sub)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kcol = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
    acc = collections.defaultdict(lambda: [0.0, set()])
    for k, disp, v in con.execute(f"select {kcol}, dispatch_id, value from counters_collection where counter_name = ?", (name,)):
        acc[k][0] += v; acc[k][1].add(disp)
    return {k: (s, len(d)) for k, (s, d) in acc.items()}
fe, wr = counter('pmc_fetch', 'FETCH_SIZE'), counter('pmc_write', 'WRITE_SIZE')
import hashlib
sha = lambda f: hashlib.sha256(open(f'drmnet_amd/csrc/{f}', 'rb').read()).hexdigest()[:16]
res = {"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --workload obsnet --precision {prec} --steps 2 --warmup 1 --no-profile (two separate passes; tools/prof_obsnet.sh)",
       "precision": prec, "conv_split2_sha16": sha("conv_split2.hip"), "attn_sha16": sha("attn.hip"),
       "units": "FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md HBM section): corrected = 2 x FETCH_SIZE",
       "workload": f"ObsNet U-Net forward, B = 32 @3x128x256, {prec}", "kernels": {}}
for k, (s, n) in sorted(fe.items(), key=lambda kv: -kv[1][0])[:16]:
    w_s, w_n = wr.get(k, (0.0, 1))
    res["kernels"][k.split('(')[0]] = {"launches": n, "fetch_kb_per_launch_raw": round(s / n, 1), "write_kb_per_launch": round(w_s / max(w_n, 1), 1),
                                       "hbm_bytes_per_launch_corrected": int((2 * s / n + w_s / max(w_n, 1)) * 1024),
                                       "hbm_bytes_all_launches_corrected": int((2 * s + w_s) * 1024)}
json.dump(res, open(f'{OUT}/{tag}_obsnet_{prec}_pmc_hbm_traffic.json', 'w'), indent=1)
print(open(f'{OUT}/{tag}_obsnet_{prec}_bench.json').read()[:300])
for r in rows[:14]: print(f"{r[2]/1e3/7:9.3f} ms/step {r[1]/7:6.1f} calls/step {r[3]:9.1f} us avg  {r[0][:100]}")
PY
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
