#!/bin/bash
# kernel timeline of the pipelined sampler loop (GPU box): where are the gaps?
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/ptrace
rm -rf $OUT && mkdir -p $OUT
PINNED=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o pt -- python3 tools/pipeline_timing_probe.py > $OUT/out.txt 2> $OUT/err.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ptrace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('eftb::','').replace('void ','')[:22], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in csv.DictReader(open(f))]
rows.sort()
# find the resum_mfma launches; print the window between the 20th and the 23rd
idx = [i for i, r in enumerate(rows) if r[2].startswith('resum_mfma')]
a, b = idx[20], idx[23]
t0 = rows[a][0]
for r in rows[a - 2:b + 1]:
    print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3]:>4s}  {r[2]}")
PY
