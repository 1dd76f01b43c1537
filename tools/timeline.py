#!/usr/bin/env python3
"""One steady-state step of an overlapped run from a rocprofv3 kernel trace (stats_kernel_trace.csv): per queue, every kernel's
start / end relative to the start of a resummation kernel in the middle of the run.  Usage: tools/timeline.py TRACE.csv [index [kernel]]
(kernel: the name fragment that marks a step, default resum_mfma = templates-first steps; resum_plk = direct-P_l steps)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mfma_peak" not in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
mark = sys.argv[3] if len(sys.argv) > 3 else "resum_mfma"
rs = [r for r in rows if mark in r["Kernel_Name"]]
print("resummation kernels:", len(rs), "periods (us):", [round((rs[i + 1]["s"] - rs[i]["s"]) / 1e3) for i in range(min(16, len(rs) - 1))])
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
t0, t1 = rs[i0]["s"], rs[i0 + 1]["s"]
for r in rows:
    if r["e"] >= t0 - 30000 and r["s"] <= t1 + 10000:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eftb::", "")[:30]
        print(f"q{r['Queue_Id']:>3} {name:30s} {(r['s'] - t0) / 1e3:8.1f} -> {(r['e'] - t0) / 1e3:8.1f}  ({(r['e'] - r['s']) / 1e3:6.1f} us)")
