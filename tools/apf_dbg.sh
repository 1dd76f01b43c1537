#!/bin/bash
# GPU box: min duration of ap_plk_fused_kernel with parts of it switched off (EFTB_APF_DBG bits: 1 no interval walk, 2 no prefix sweeps, 4 no pieces)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for d in 0 1 2 4 7; do
  OUT=gpurun_out/apf$d; rm -rf $OUT; mkdir -p $OUT
  EFTB_APF_DBG=$d HP_ONLY_DIRECT=1 HP_K=10 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/direct_probe.py > $OUT/out.txt 2> $OUT/err.log
  python3 - $OUT/t_kernel_stats.csv $d <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ap_plk_fused" in r["Name"]: print("dbg", sys.argv[2], "ap_plk_fused min", float(r["MinNs"]) / 1e3, "avg", float(r["AverageNs"]) / 1e3, flush=True)
PY
done
