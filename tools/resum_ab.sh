#!/bin/bash
# same-box A/B of builds of the library (GPU box): tools/resum_ab.sh rounds lib ... ; "-" = the in-tree build
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/resum_ab.txt; : > $out
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    lib=$spec
    [[ $lib == "-" ]] && lib=""
    EFTB_LIB=$lib timeout -k 10 120 python3 tools/resum_ab.py 2>>gpurun_out/resum_ab.err | tail -1 >> $out || { echo "failed: $spec" >> $out; cat $out; exit 1; }
  done
done
cat $out
