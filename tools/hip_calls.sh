#!/bin/bash
# GPU box: HIP API calls of the staged direct-P_l loop (rocprofv3 --hip-trace --stats of tools/host_split_probe.py): calls per step and mean duration
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/hc
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --hip-trace --stats --output-format csv -d $OUT -o t -- python3 tools/host_split_probe.py > $OUT/out.txt 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
cat $OUT/out.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/hc/**/*hip_api_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
steps = 60 * 6 + 2 * 6
for r in rows[:14]:
    print(f"{r['Name'][:34]:34s} calls {int(r['Calls']):7d} ({int(r['Calls']) / steps:5.1f} per step) avg {float(r['AverageNs']) / 1e3:7.2f} us  total/step {float(r['TotalDurationNs']) / steps / 1e3:7.1f} us")
PY
