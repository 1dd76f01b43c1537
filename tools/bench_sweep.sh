#!/bin/bash
# GPU box: bench.py over a grid of (steps, EFTB_BENCH_DEPTH, EFTB_BENCH_COALESCE, EFTB_SUB_INFLIGHT); one line per run.
# usage: tools/bench_sweep.sh OUTDIR "K..." "DEPTH..." "COALESCE..." "INFLIGHT..."
cd "$GRAFT_REPO_ROOT"
OUT=$1; mkdir -p $OUT
for k in $2; do for d in $3; do for c in $4; do for i in $5; do
  f=$OUT/k${k}_d${d}_c${c}_i${i}.json
  EFTB_BENCH_DEPTH=$d EFTB_BENCH_COALESCE=$c EFTB_SUB_INFLIGHT=$i python3 bench.py --steps $k --no-cpu-baseline --no-extras > $f 2> $f.err || echo "FAILED $f"
  python3 - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    h = d.get("host_us_per_step", {})
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), "launches", h.get("launches"), "issue", round(h.get("issue_us_per_step", 0), 1), "wait", round(h.get("wait_us_per_step", 0), 1), flush=True)
except Exception as exc:
    print(sys.argv[1], "ERR", exc, flush=True)
PY
done; done; done; done
