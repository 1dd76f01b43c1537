#!/bin/bash
# same-box A/B of one environment setting on the pipelined loop (GPU box): alternates unset / set; usage: tools/env_ab.sh NAME=VALUE [rounds]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/envab.txt; : > $out
for i in $(seq 1 ${2:-4}); do
  for setting in "X=0" "$1"; do
    echo -n "$setting : " >> $out
    env $setting HP_K=40 python3 tools/staged_vs_resident.py 2>/dev/null | grep -E "fetch depth 2|resident" | awk '{printf "%s  ", $(NF-1)}' >> $out
    echo >> $out
  done
done
cat $out
