#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/gtrace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o gt -- python3 tools/gather_probe.py > $OUT/out.txt 2> $OUT/err.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/gtrace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('eftb::','').replace('void ','')[:28], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith('resum_mfma')]
a, b = idx[60], idx[62]   # inside the first gather=True block (43 + 3 + ...)
t0 = rows[a][0]
for r in rows[a - 3:b + 1]:
    print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3]:>4s}  {r[2]}")
PY
