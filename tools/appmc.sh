set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/appmc && mkdir -p gpurun_out/appmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/appmc -o p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/appmc/b1.json 2> gpurun_out/appmc/e1.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/appmc -o p2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/appmc/b2.json 2> gpurun_out/appmc/e2.log
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/appmc/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,d in agg.items():
        if any(x in kn for x in ('ap_apply','resum_kernel','spline','pair_gemm4')):
            print(kn, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
