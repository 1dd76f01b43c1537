#!/bin/bash
# GPU box: kernel timeline of the whole K = 20 timed region of bench.py (direct-P_l, coalescing defaults): segments of the resum_plk timeline
# separated by idle gaps, then every kernel of the 20-step segment
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=${1:-gpurun_out/tk20}
rm -rf $OUT && mkdir -p $OUT
EFTB_BENCH_PREWARM_MS=5 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 20 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
f=$(ls $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" $OUT/bench.json <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[2]))
print("bench under the tracer:", round(d["value"]), d["ms_per_step"], d.get("host_us_per_step"))
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
gk = "Grid_Size" if "Grid_Size" in rows[0] else "Grid_Size_X"
rs = [r for r in rows if "resum_plk_kernel" in r["Kernel_Name"]]
segs, cur = [], [rs[0]]
for a, b in zip(rs, rs[1:]):
    if b["s"] - a["e"] > 400000:
        segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
def steps(seg): return sum(int(r[gk]) // 294912 for r in seg)
print("segments (launches, steps):", [(len(s), steps(s)) for s in segs])
seg = [s for s in segs if steps(s) == 20][0]
t0 = min(r["s"] for r in rows if r["s"] >= seg[0]["s"] - 600000 and "stage_gather" in r["Kernel_Name"])
t1 = seg[-1]["e"] + 400000
for r in rows:
    if r["s"] >= t0 - 1000 and r["s"] <= t1:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eftb::", "")[:24]
        print(f"q{r['Queue_Id']:>3} {name:24s} {(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f}) g {int(r[gk]) }")
PY
