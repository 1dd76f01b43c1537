#!/bin/bash
# per-kernel times of one bench run (GPU box): rocprofv3 --kernel-trace --stats, printed as a table
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/kstats
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kstats/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:72]:72s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:10.1f} us {r['Percentage']:>6s}%")
PY
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/kstats/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'synth' in n or 'expand' in n:
        agg[(n.split('(')[0][-24:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items(): print('  per-grid', k, len(v), round(sum(v)/len(v),1), 'us')
PY
