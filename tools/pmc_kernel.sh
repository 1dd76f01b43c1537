#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> COUNTER...   (GPU box) -> average counter values of the matching kernel
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
pat=$1; shift
OUT=gpurun_out/pmck
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/err.log
python3 - "$pat" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
f=glob.glob('gpurun_out/pmck/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(pat, {c: round(sum(v)/len(v)) for c,v in agg.items()}, 'dispatches', max(len(v) for v in agg.values()) if agg else 0)
PY
