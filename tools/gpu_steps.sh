#!/bin/bash
# Runs GPU steps one after another on the GPU box; a step that hits its time limit (or is killed) ends the call -- no further GPU
# step is started after a hang.  Usage: tools/gpu_steps.sh OUTDIR 'name|seconds|command' ...
OUT=$1; shift
mkdir -p "$OUT"
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "== $name (limit ${secs}s)"
    timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.out" 2> "$OUT/$name.err"
    rc=$?
    echo "== $name rc=$rc"
    tail -n 6 "$OUT/$name.out"
    if [ $rc -ne 0 ]; then tail -n 12 "$OUT/$name.err"; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name hit its limit: stopping here"; exit $rc; fi
done
exit 0
