#!/bin/bash
# GPU box: resum_plk_kernel wave shapes (EFTB_RPLK = KPL SH as two digits): alone time at B = 128 / 384 and the bench at 200 steps
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for shape in 42 44 81 82 84 24; do
  for b in 128 384; do
    OUT=gpurun_out/rplk; rm -rf $OUT; mkdir -p $OUT
    EFTB_RPLK=$shape HP_B=$b HP_ONLY_DIRECT=1 HP_K=10 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/direct_probe.py > $OUT/out.txt 2> $OUT/err.log
    python3 - $OUT/t_kernel_stats.csv $shape $b <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "resum_plk" in r["Name"]: print("shape", sys.argv[2], "B", sys.argv[3], "resum_plk min", round(float(r["MinNs"]) / 1e3, 1), "avg", round(float(r["AverageNs"]) / 1e3, 1), flush=True)
PY
  done
  printf "EFTB_RPLK=$shape\n" | K=200 bash tools/env_sweep.sh
done
