set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/approf && mkdir -p gpurun_out/approf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/approf -o ap -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/approf/bench.json 2> gpurun_out/approf/err.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/approf/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
