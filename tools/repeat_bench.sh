#!/bin/bash
# run-to-run spread of the default bench line (GPU box): N runs, value / ms_per_step and the per-step fetch-complete intervals -> gpurun_out/repeat.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/repeat.txt; : > $out
for i in $(seq 1 ${1:-8}); do
  EFTB_BENCH_STEP_TIMES=1 python3 bench.py --no-cpu-baseline --no-extras 2> gpurun_out/rb.err | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(j['value']), round(j['ms_per_step'],4), end=' ')" >> $out
  python3 - >> $out <<'PY'
l=[x for x in open('gpurun_out/rb.err') if x.startswith('[bench] fetch-complete')][0]
t=[float(x) for x in l.split('):')[1].split()]
print(' '.join(f"{b-a:.2f}" for a,b in zip([0]+t[:-1],t)))
PY
done
cat $out
