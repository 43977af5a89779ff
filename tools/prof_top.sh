#!/bin/bash
# rocprofv3 kernel stats of a short bench run; prints the top kernels (name, calls, total ms, avg us)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_top"
rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_top" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --no-strict-fp32 --no-secondary --no-live-traffic "$@" > "$ROOT/gpurun_out/prof_top.log" 2>&1
cd "$ROOT"
python3 - <<'PY'
import sqlite3, glob
db = sorted(glob.glob('gpurun_out/prof_top/**/*_results.db', recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average from top_kernels order by total_duration desc limit 40").fetchall()
for n, c, t, a in rows:
    print(f"{t/1e3/7:9.3f} ms/step {c/7:7.1f} calls/step {a:9.1f} us avg  {n[:110]}")  # 1 warm-up + 3 timed + 3 breakdown-pass steps
PY
tail -1 gpurun_out/prof_top.log | cut -c1-160
