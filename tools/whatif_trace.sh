#!/bin/bash
# GPU box: kernel timeline of one steady-state step of the resident pipelined loop (tools/resum_ab.py) under EFTB_WHATIF=$1
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/wt$1
rm -rf $OUT && mkdir -p $OUT
EFTB_WHATIF=$1 HP_K=30 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 tools/resum_ab.py > $OUT/out.txt 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 tools/timeline.py $(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1) ${2:-100} > $OUT/timeline.txt
cat $OUT/out.txt; cat $OUT/timeline.txt
