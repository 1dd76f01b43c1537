#!/bin/bash
# kernel timeline of the pipelined bench loop (GPU box): rocprofv3 --kernel-trace of a short bench.py, then one steady-state step per queue
# usage: tools/trace_step.sh OUTDIR [resum index]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$1; IDX=${2:-9}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 bench.py --steps ${STEPS:-16} --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.log || exit 1
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/timeline.py $f $IDX > $OUT/step.txt
python3 - $f $OUT/compact.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as fh:
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eftb::", "")[:28]
        fh.write(f'{r["Queue_Id"]},{n},{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - t0) / 1e3:.1f}\n')
PY
rm -f $f
cat $OUT/step.txt
