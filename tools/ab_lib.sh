#!/bin/bash
# same-box A/B of two builds of the library on the pipelined loop (GPU box): alternates EFTB_LIB between the in-tree build and $1
# usage: tools/ab_lib.sh other.so [rounds]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab.txt; : > $out
for i in $(seq 1 ${2:-4}); do
  for lib in "" "$1"; do
    echo -n "${lib:-in-tree} : " >> $out
    EFTB_LIB=$lib HP_K=40 python3 tools/staged_vs_resident.py 2>/dev/null | grep -E "fetch depth 2|resident" | awk '{printf "%s %s  ", $(NF-4), $(NF-1)}' >> $out
    echo >> $out
  done
done
cat $out
