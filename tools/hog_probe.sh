#!/bin/bash
# 32 CPU hogs for ~40 s; bench with and without graph replay meanwhile (does the step time depend on how busy the host is?)
cd "$GRAFT_REPO_ROOT"
pids=""
for i in $(seq 1 32); do timeout 50 python3 -c "
while True: pass" & pids="$pids $!"; done
sleep 2
python bench.py --no-cpu-baseline --steps 20 2>/dev/null | cut -c1-190
EFTB_NO_GRAPH=1 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | cut -c1-190
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
echo done
