#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in 41 42 21 22 12 24; do
  echo "== EFTB_RSD_CFG=$cfg"
  EFTB_RSD_CFG=$cfg tools/kstat_direct.sh 2>&1 | grep -E "resum_plk"
  grep -E "direct=True|set 0" gpurun_out/kd/out.txt | head -2
done
