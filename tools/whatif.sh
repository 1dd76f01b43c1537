#!/bin/bash
# GPU box: step time with kernels of the look-ahead chain skipped (EFTB_WHATIF bits; timing only)
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for w in 0 1 2 3; do echo -n "whatif=$w: "; EFTB_WHATIF=$w timeout -k 10 120 python3 tools/resum_ab.py 2>/dev/null | tail -1; done; done
