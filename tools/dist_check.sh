#!/bin/bash
# The N > 1 code path of bench.py with one rank over RCCL (the driver launches it the same way with N ranks): exactly one JSON line on stdout,
# the traffic import of the default mode, the barrier-safe full-chain pass of every rank.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
DRM_BENCH_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null > /tmp/dist_check.out
python3 - <<'PY'
import json
L = [l for l in open('/tmp/dist_check.out') if l.strip()]
print(len(L), "line(s) on stdout")
d = json.loads(L[-1])
print("value", d["value"], "dtype", d["dtype"][:40], "traffic", d["roofline"]["traffic"], (d["roofline"].get("traffic_source") or d["roofline"].get("traffic_note") or "")[:90])
print("full_chain_all_gpus", d.get("full_chain_all_gpus"))
PY
