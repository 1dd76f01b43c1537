#!/bin/bash
# throughput against the batch size per GPU (GPU box): one bench line per batch
for b in "$@"; do
  timeout -k 10 300 python bench.py --steps 10 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=$b', round(d['value']), 'evals/s', round(d['ms_per_step'],3), 'ms/step', {k:round(x,3) for k,x in r['stage_ms'].items()})" || exit 1
done
