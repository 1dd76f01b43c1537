#!/bin/bash
# GPU box: kernel timeline (rocprofv3 --kernel-trace) of bench.py's timed direct-P_l loop in the coalescing configuration given by the environment
# (EFTB_BENCH_DEPTH / EFTB_BENCH_COALESCE / EFTB_SUB_INFLIGHT); prints ~1.5 ms of the steady state, every queue.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/tb}
rm -rf $OUT && mkdir -p $OUT
EFTB_BENCH_DIRECT_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps ${K:-80} --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" $OUT/bench.json <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[2]))
print("bench under the tracer:", round(d["value"]), d["ms_per_step"], d.get("host_us_per_step"))
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rs = [r for r in rows if "resum_plk_kernel" in r["Kernel_Name"]]
print("direct resum launches:", len(rs))
i0 = len(rs) * 2 // 3
t0 = rs[i0]["s"]
print("periods us", [round((rs[i + 1]["s"] - rs[i]["s"]) / 1e3) for i in range(i0 - 6, min(i0 + 10, len(rs) - 1))])
for r in rows:
    if r["e"] >= t0 - 100000 and r["s"] <= t0 + 1400000:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eftb::", "")[:26]
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        print(f"q{r['Queue_Id']:>3} {name:26s} {(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f} us) grid {g}")
PY
