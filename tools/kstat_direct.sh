#!/bin/bash
# GPU box: per-kernel times (rocprofv3 --stats) of tools/direct_probe.py: min = nothing beside it
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/kd
rm -rf $OUT && mkdir -p $OUT
HP_K=20 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/direct_probe.py > $OUT/out.txt 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kd/t_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:24]: print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f}")
PY
