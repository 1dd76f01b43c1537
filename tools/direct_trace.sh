#!/bin/bash
# GPU box: kernel stats and one steady-state step of the direct-P_l pipelined loop (tools/direct_probe.py)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/dt${HP_B:-128}
rm -rf $OUT && mkdir -p $OUT
HP_ONLY_DIRECT=1 HP_K=30 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/direct_probe.py > $OUT/out.txt 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
cat $OUT/out.txt
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rs = [r for r in rows if "resum_plk_kernel" in r["Kernel_Name"]]
print("direct resum launches:", len(rs))
i0 = len(rs) - 20
t0, t1 = rs[i0]["s"], rs[i0 + 1]["s"]
print("period us", (t1 - t0) / 1e3, [round((rs[i + 1]["s"] - rs[i]["s"]) / 1e3) for i in range(i0 - 8, i0 + 8)])
for r in rows:
    if r["e"] >= t0 - 60000 and r["s"] <= t1 + 200000:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eftb::", "")[:30]
        print(f"q{r['Queue_Id']:>3} {name:30s} {(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f} us)")
PY
