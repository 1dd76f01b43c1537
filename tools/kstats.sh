#!/bin/bash
# rocprofv3 kernel-trace summary of a short bench run -> $1/stats_kernel_stats.csv (GPU box; the program itself follows `--`)
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
f=$(ls "$OUT"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {float(r["Percentage"]):5.1f} %')
PY
