#!/bin/bash
# stream-priority A/B at the driver's K = 20 (GPU box): value per setting -> gpurun_out/prio.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prio.txt; : > $out
run() { echo -n "$* : " >> $out; env "$@" python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(j['value']), round(j['ms_per_step'],4), j.get('sync_step_ms'))" >> $out; }
run X=0
run EFTB_BACK_PRIO=-1
run EFTB_BACK_PRIO=0
run EFTB_BACK_PRIO=-1 EFTB_MAIN_PRIO=0
run EFTB_MAIN_PRIO=0
run EFTB_MAIN_PRIO=-1 EFTB_PRE_PRIO=0 EFTB_BACK_PRIO=-1
run EFTB_BENCH_DEPTH=1
run EFTB_BENCH_DEPTH=3
run X=0
cat $out
