#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> one bench line per value (GPU box)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in r['stage_ms'].items()})" || exit 1
done
