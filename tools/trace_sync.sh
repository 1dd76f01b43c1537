#!/bin/bash
# device timeline of one dependent-sampler step (GPU box): rocprofv3 --kernel-trace --memory-copy-trace of tools/sync_probe.py
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/trs}
rm -rf $OUT && mkdir -p $OUT
python3 tools/sync_probe.py > $OUT/plain.txt 2>&1
HP_N=12 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tr -- python3 tools/sync_probe.py > $OUT/traced.txt 2> $OUT/err.log || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void ', '').replace('eftb::', '')[:28]))
for f in glob.glob(out + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'dma', r.get('Direction', 'copy')))
rows.sort()
pr = [i for i, r in enumerate(rows) if r[3].startswith('prep_rows')]
a, b = pr[-4], pr[-3]
while a > 0 and rows[a - 1][3].startswith('stage_copy') : a -= 1
t0 = rows[a][0]
with open(out + '/step.txt', 'w') as fh:
    for r in rows[a:b + 1]:
        fh.write(f"{r[2]:>4s} {r[3]:30s} {(r[0] - t0) / 1e3:8.1f} -> {(r[1] - t0) / 1e3:8.1f}  ({(r[1] - r[0]) / 1e3:6.1f} us)\n")
PY
for f in $OUT/*.csv $OUT/*/*.csv; do rm -f $f; done
cat $OUT/plain.txt $OUT/traced.txt $OUT/step.txt
