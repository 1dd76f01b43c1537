#!/bin/bash
# GPU box: step time under EFTB_WHATIF settings (timing only): tools/whatif.sh rounds w1 w2 ...
cd "$GRAFT_REPO_ROOT"
r=$1; shift
for i in $(seq 1 $r); do for w in "$@"; do echo -n "whatif=$w: "; EFTB_WHATIF=$w timeout -k 10 120 python3 tools/resum_ab.py 2>/dev/null | tail -1; done; done
