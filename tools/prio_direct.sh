#!/bin/bash
# GPU box: stream priorities (1 low, 0 normal, -1 high) on direct-P_l steps: resident and staged (views) loops
cd "$GRAFT_REPO_ROOT"
for cfg in "1 -1 0 1" "1 -1 0 0" "1 -1 0 -1" "0 -1 0 1" "1 -1 -1 1" "1 0 0 0" "0 0 0 0" "1 -1 0 1"; do
  set -- $cfg
  echo -n "main=$1 pre=$2 side=$3 back=$4 : "
  EFTB_MAIN_PRIO=$1 EFTB_PRE_PRIO=$2 EFTB_SIDE_PRIO=$3 EFTB_BACK_PRIO=$4 HP_K=60 timeout -k 10 120 python3 tools/staged_vs_resident.py 2>/dev/null | grep -E "resident|depth 3|views" | awk '{printf "%s ", $(NF-5)}'
  echo
done
