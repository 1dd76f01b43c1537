set -e
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -W ignore 2>&1 | tail -3
for rs in 1 2 3; do
EFTB_AP_ROWSPLIT=$rs timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rs=$rs', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in r['stage_ms'].items()})"
done
