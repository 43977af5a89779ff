#!/bin/bash
# Regenerates the round's judged profile artifacts on the GPU box (run through gpurun, then copy gpurun_out/prof_round/*
# into profiles/):  kernel stats of the default bench command, the bench line measured without the profiler, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE) for the HBM traffic of the dominant kernels.
#   usage: tools/prof_round.sh <tag> [precision]       e.g. tools/prof_round.sh r03 f16mx   (precision defaults to bench.py's default)
tag=${1:-r03}
prec=${2:-f16mx}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_round"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --precision $prec --no-live-traffic > $OUT/${tag}_bench_${prec}.log 2>&1   # (the PMC passes below are this script's own)
grep "^{\"metric\"" $OUT/${tag}_bench_${prec}.log | tail -1 > $OUT/${tag}_bench_${prec}.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $ROOT/bench.py --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-parity-check --no-strict-fp32 --no-secondary --no-live-traffic > $OUT/${tag}_bench_${prec}_under_rocprof.log 2>&1
grep "^{\"metric\"" $OUT/${tag}_bench_${prec}_under_rocprof.log | tail -1 > $OUT/${tag}_bench_${prec}_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-check --no-strict-fp32 --no-secondary > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- python3 $ROOT/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-check --no-strict-fp32 --no-secondary > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 - "$tag" "$prec" <<'PY'
import sqlite3, glob, json, sys, csv, collections
tag, prec = sys.argv[1], sys.argv[2]
OUT = 'gpurun_out/prof_round'
def db(sub):
    return sqlite3.connect(sorted(glob.glob(f'{OUT}/{sub}/**/*_results.db', recursive=True))[-1])
con = db('stats')
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
with open(f'{OUT}/{tag}_bench_{prec}_kernel_stats.csv', 'w', newline='') as f:
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
res = {"conv_split2_sha16": hashlib.sha256(open('drmnet_amd/csrc/conv_split2.hip', 'rb').read()).hexdigest()[:16],
       "precision": prec,
       "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --precision {prec} --steps 2 --warmup 1 --no-cpu-baseline --no-profile (two separate passes; tools/prof_round.sh)",
       "units": "FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md HBM section): corrected = 2 x FETCH_SIZE",
       "kernels": {}}
for k, (s, n) in sorted(fe.items(), key=lambda kv: -kv[1][0])[:12]:
    w_s, w_n = wr.get(k, (0.0, 1))
    name = k.split('(')[0]
    res["kernels"][name] = {"launches": n, "fetch_kb_per_launch_raw": round(s / n, 1), "write_kb_per_launch": round(w_s / max(w_n, 1), 1),
                            "hbm_bytes_per_launch_corrected": int((2 * s / n + w_s / max(w_n, 1)) * 1024)}
# the algorithmic bytes of the SAME instantiation's launches (bench.py's per-variant profiler totals of this run): the traffic ratio of the
# dominant kernel compares like with like
try:
    bl = json.load(open(f'{OUT}/{tag}_bench_{prec}.json'))
    dv = (bl.get("roofline") or {}).get("dominant_variant")
    if dv and dv["kernel"] in res["kernels"]:
        k = res["kernels"][dv["kernel"]]
        k["algorithmic_bytes_per_launch"] = dv["algorithmic_bytes_per_launch"]
        k["traffic_ratio"] = round(k["hbm_bytes_per_launch_corrected"] / dv["algorithmic_bytes_per_launch"], 3)
        res["dominant_variant"] = dv["kernel"]
except (OSError, ValueError, KeyError) as e:
    res["dominant_variant_note"] = f"bench line not readable: {e}"
json.dump(res, open(f'{OUT}/{tag}_pmc_hbm_traffic_{prec}.json', 'w'), indent=1)
print(open(f'{OUT}/{tag}_bench_{prec}.json').read()[:400])
for r in rows[:8]: print(r)
print(json.dumps(list(res["kernels"].items())[:2], indent=0)[:600])
PY
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
