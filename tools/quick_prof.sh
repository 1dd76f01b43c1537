#!/bin/bash
# One GPU call, scratch numbers while tuning: per-kernel times with every overlap off (rocprofv3 --kernel-trace --stats), then three --pmc passes
# (HBM fetch, HBM write, wave cycles) of a 2-step bench -> gpurun_out/qp/summary.txt.  Not the judged evidence (tools/profile_round.sh is).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/qp
rm -rf $OUT && mkdir -p $OUT
tools/kstats_serial.sh $OUT/serial > $OUT/serial.txt 2>&1 || exit 1
unset EFTB_AP_OVERLAP EFTB_PREP_OVERLAP
pmc() {
    local name=$1; shift
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/$name.err || exit 1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/qp/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    g = collections.defaultdict(collections.Counter)
    for r in rows: g[r['Kernel_Name']][int(r['Grid_Size'])] += 1
    gmax = {k: max(c.items(), key=lambda kv: kv[1] * kv[0])[0] for k, c in g.items()}
    for r in rows:
        if int(r['Grid_Size']) != gmax[r['Kernel_Name']]: continue
        n = r['Kernel_Name'].replace('void ', '').split('(')[0]
        per[n][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': per[n]['dur_us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':44s} {'us(pmc)':>8s} {'rdMB':>7s} {'wrMB':>7s} {'waveMcyc':>9s} {'wait_any':>8s} {'wait_inst':>9s}")
for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])) / max(1, len(kv[1].get('SQ_WAVE_CYCLES', [0])))):
    if not n.startswith('eftb::'): continue
    a = lambda c: sum(d[c]) / len(d[c]) if d.get(c) else float('nan')
    wc = a('SQ_WAVE_CYCLES')
    print(f"{n[6:50]:44s} {a('dur_us'):8.1f} {2 * a('FETCH_SIZE') / 1024:7.1f} {a('WRITE_SIZE') / 1024:7.1f} {wc / 1e6:9.1f} {a('SQ_WAIT_ANY') / wc:8.2f} {a('SQ_WAIT_INST_ANY') / wc:9.2f}")
PY
cat $OUT/serial.txt
cat $OUT/summary.txt
