#!/bin/bash
# GPU box: bench.py (direct loop only matters) under one environment setting per line of stdin ("NAME=VALUE ..." or empty for the defaults)
cd "$GRAFT_REPO_ROOT"
K=${K:-200}
while read -r line; do
  for rep in 1 2; do
    env $line python3 bench.py --steps $K --no-cpu-baseline --no-extras > /tmp/es.json 2>/dev/null
    python3 - "$line" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/es.json")); print(f"{sys.argv[1]!r:60s} {d['value']:10.0f} {d['ms_per_step']:.4f}  tf {d.get('templates_first_evaluations_per_s', 0):.0f}", flush=True)
except Exception as exc:
    print(sys.argv[1], "ERR", exc, flush=True)
PY
  done
done
