#!/bin/bash
# GPU box: (1) per-kernel times with every overlap off, (2) one steady-state step of the pipelined bench loop from the kernel trace
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/qt
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 tools/kstats_serial.sh $OUT/serial > $OUT/serial.txt 2>&1 || { cat $OUT/serial.txt; tail -5 $OUT/serial/stats.err; exit 1; }
unset EFTB_AP_OVERLAP EFTB_PREP_OVERLAP
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pipe -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/pipe.json 2> $OUT/pipe.err || { tail -5 $OUT/pipe.err; exit 1; }
python3 tools/timeline.py $(ls $OUT/pipe/*kernel_trace.csv $OUT/pipe/*/*kernel_trace.csv 2>/dev/null | head -1) 60 > $OUT/timeline.txt
cat $OUT/serial.txt
cat $OUT/timeline.txt
