#!/bin/bash
# GPU busy fraction of a bench workload: sum of kernel durations (rocprofv3 kernel trace) / wall time of the timed steps
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/idle_probe"; rm -rf "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-profile --no-parity-check > "$OUT.log" 2>&1
cd "$ROOT"
python3 - <<'PY'
import sqlite3, glob, json
db = sorted(glob.glob('gpurun_out/idle_probe/**/*_results.db', recursive=True))[-1]
con = sqlite3.connect(db)
tot, calls = con.execute("select sum(total_duration), sum(total_calls) from top_kernels").fetchone()
line = [l for l in open('gpurun_out/idle_probe.log') if l.startswith('{"metric"')][-1]
d = json.loads(line)
steps = d["steps"] + d["warmup"]
print(f"kernels: {calls} launches, {tot/1e3:.1f} ms total (incl. weight packing); bench: {d['ms_per_step']:.3f} ms/step x {steps} steps = {d['ms_per_step']*steps:.1f} ms")
rows = con.execute("select name, total_calls, total_duration from top_kernels where name not like '%pack%' and name not like '%absmax%' and name not like '%split_scale%' and name not like '%copyBuffer%' order by total_duration desc").fetchall()
busy = sum(r[2] for r in rows) / 1e3
print(f"busy (without load-time kernels): {busy:.1f} ms -> {busy / (d['ms_per_step']*steps):.2f} of the stepped wall time")
PY
