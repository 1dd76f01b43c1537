#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by tools/profile_round.sh on the GPU box) into profiles/rNN_*.

    python tools/profile_collect.py r01

Writes: rNN_bench.json, rNN_bench_under_rocprof.json, rNN_kernel_stats.csv (rocprofv3 --stats, verbatim),
rNN_pmc_per_kernel.json (average counter values per kernel) and rNN_pmc_dominant.json (the dominant kernel: HBM bytes
per launch corrected as MI355X_MICROARCH.md prescribes, MFMA busy fraction, L2 hit rate, rocprof vs HIP-event time).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pat):
    hits = glob.glob(os.path.join(src, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pat} under {src}")
    return hits[0]


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


for n in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, n), os.path.join(dst, f"{tag}_{n}"))
shutil.copy(find("stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
import subprocess
# a step in the middle of the TIMED region of the traced run: its warm-up ran `warmup_steps_run` steps before it (bench.py's own count)
import json as _json
_traced = _json.loads([ln for ln in open(os.path.join(src, "bench_under_rocprof.json")) if ln.startswith("{")][0])
_idx = int(_traced.get("warmup_steps_run", _traced.get("warmup", 3))) + _traced.get("steps", 20) // 2
_mark = "resum_plk" if "roofline_templates_first" in _traced.get("roofline", {}) else "resum_mfma"  # the timed loop runs direct-P_l steps when it reports both
# (coalesced launches: fewer resummation launches than steps -- take the index as the same FRACTION of the launches the trace holds)
import csv as _csv
_nres = sum(1 for r in _csv.DictReader(open(find("stats_kernel_trace.csv"))) if _mark in r["Kernel_Name"])
_nsteps_total = int(_traced.get("warmup_steps_run", 3)) + 3 * _traced.get("steps", 20) + 40   # warm-up + timed + templates-first + checks (rough: only the fraction matters)
_idx = max(2, min(_nres - 3, int(_nres * (int(_traced.get("warmup_steps_run", 3)) + _traced.get("steps", 20) // 2) / max(_nsteps_total, 1))))
tl = subprocess.run([sys.executable, os.path.join(root, "tools", "timeline.py"), find("stats_kernel_trace.csv"), str(_idx), _mark], capture_output=True, text=True).stdout
open(os.path.join(dst, f"{tag}_overlap_timeline.txt"), "w").write(
    "One steady-state step of the pipelined timed loop (rocprofv3 --kernel-trace of bench.py; q = HSA queue: main / side / look-ahead / back / copy; under the tracer the host, not the GPU, sets the pace of direct-P_l steps).\n" + tl)

per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("pmc_fetch_counter_collection.csv", "pmc_write_counter_collection.csv", "pmc_sq_counter_collection.csv",
          "pmc_l2hit_counter_collection.csv", "pmc_l2miss_counter_collection.csv"):
    rows_f = list(csv.DictReader(open(find(f))))
    # the command also launches every kernel at batch 1-2 (drop-in latency probe, parity check) and at batch 256 (side key): keep the launches
    # of the bench batch, i.e. the grid with the largest launches x size per kernel
    gcount = collections.defaultdict(collections.Counter)
    for r in rows_f:
        gcount[r["Kernel_Name"]][int(r["Grid_Size"])] += 1
    gmax = {k: max(c.items(), key=lambda kv: kv[1] * kv[0])[0] for k, c in gcount.items()}  # launches x grid: batch 1 is frequent but tiny, batch 256 large but rare
    for r in rows_f:
        if int(r["Grid_Size"]) != gmax[r["Kernel_Name"]]:
            continue
        per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            per[short(r["Kernel_Name"])]["duration_ns_under_pmc"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items() if k.startswith("eftb::")}
for k, d in per.items():
    if k in avg:
        avg[k]["launches_counted"] = max(len(v) for v in d.values())
json.dump(avg, open(os.path.join(dst, f"{tag}_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)

# FETCH_SIZE calibration (tools/fetch_calib.py: 1 GiB streamed once per launch at 8 and 16 bytes per lane)
calib = {}
try:
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(find("calib_counter_collection.csv"))):
        if "stream_read_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in rows.items():
        calib[k] = {"bytes_streamed": 1 << 30, "FETCH_SIZE_KB": v, "bytes_over_counter": [(1 << 30) / (x * 1024.0) for x in v]}
    json.dump(calib, open(os.path.join(dst, f"{tag}_fetch_calibration.json"), "w"), indent=1, sort_keys=True)
    print("calibration", json.dumps(calib))
except SystemExit:
    print("no calibration pass found")

bench = json.load(open(os.path.join(src, "bench.json")))
stats = {short(r["Name"]): r for r in csv.DictReader(open(find("stats_kernel_stats.csv")))}
B, NK = bench["config"]["batch_per_gpu"], 512
tr = list(csv.DictReader(open(find("stats_kernel_trace.csv"))))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in tr]
gy = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) for r in tr}

ALGO = {
    "eftb::resum_mfma_kernel": f"H table 3*80*{NK}*8 = {3 * 80 * NK * 8 / 1e6:.1f} MB (L2-resident, re-read per cosmology) + per-s records {B * 80 * 48 * 8 / 1e6:.1f} MB + per-s operand "
                               f"{B * 80 * 512 * 8 / 1e6:.1f} MB + template read-modify-write 2*{B}*63*{NK}*8 = {2 * B * 63 * NK * 8 / 1e6:.0f} MB",
    "eftb::resum_plk_kernel": f"H table {3 * 80 * NK * 8 / 1e6:.1f} MB (L2-resident) + coefficient table {B * 80 * 160 * 8 / 1e6:.1f} MB (scalar loads) + P_l read-modify-write "
                              f"2*{B}*3*{NK}*8 = {2 * B * 3 * NK * 8 / 1e6:.1f} MB",
    "eftb::ap_plk_kernel": f"B-spline coefficients of the contracted rows {B * 3 * NK * 8 / 1e6:.1f} MB + per-interval matrices {NK * 16 * 8 / 1e6:.2f} MB (L2) + stochastic rows and P_l "
                           f"{B * 3 * 4 * NK * 8 / 1e6:.1f} MB (P_l to mapped host memory)",
    "eftb::synth_kernel": f"synthesis bases syn_k / syn_s / lin_k / lin_s ({(528 * NK + 528 * 80 + 288 * NK + 288 * 80) * 8 / 1e6:.1f} MB, re-read per row tile through the L2) + row operands "
                          f"{B * (8 * 528 + 32 * 528 + 10 * 288 + 6 * 288) * 8 / 1e6:.1f} MB + outputs {B * (8 * NK + 32 * 80 + 10 * NK + 6 * 80) * 8 / 1e6:.1f} MB",
}


def kernel_block(prefix, hip_ms=None, hip_alone_ms=None):
    """PMC / rocprof summary of the bench-batch launches of the kernel whose (short) name starts with `prefix`."""
    dom = next(k for k in avg if k.startswith(prefix))
    c = avg[dom]
    o = {
        "kernel": f"{dom}, batch {B}, Nk {NK}",
        "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
        "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
        "note": "(2*FETCH_SIZE+WRITE_SIZE)*1024 from separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of wide streams as 64 B) -- "
                f"calibrated for 8-byte-per-lane reads as well: {tag}_fetch_calibration.json, 1 GiB streamed at 8 and at 16 bytes per lane both read bytes / (FETCH_SIZE*1024) = 2.000. "
                "Algorithmic bytes per launch: " + next((v for k, v in ALGO.items() if dom.startswith(k)), "n/a") + ".",
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "kernel_cycles_est": c["GRBM_GUI_ACTIVE"] / 8.0,
        "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
        "valu_insts_per_simd_cycle": c["SQ_INSTS_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
        "effective_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8.0 / c["duration_ns_under_pmc"],
        "duration_us_under_pmc": c["duration_ns_under_pmc"] / 1e3,
        "L2_hit": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "rocprof_avg_us": float(stats[dom]["AverageNs"]) / 1e3,
    }
    if hip_ms is not None:
        o["bench_hip_event_us"] = hip_ms * 1e3
    if hip_alone_ms is not None:
        o["bench_hip_event_alone_us"] = hip_alone_ms * 1e3
    # the rocprof average mixes the pipelined launches of the timed region (other kernels beside them) with stand-alone launches (per-stage timings, PMC-free
    # synchronous calls): split the kernel trace by whether a kernel of another queue ran during the launch; bench-batch launches only (largest launches x grid)
    gcnt = collections.Counter(gy[(s_, e_)] for s_, e_, q_, n_ in ev if n_ == dom)
    gbig = max(gcnt.items(), key=lambda kv: kv[0] * kv[1])[0]
    beside, alone = [], []
    for s0, e0, q0, n0 in ev:
        if n0 != dom or gy[(s0, e0)] != gbig:
            continue
        shared = sum(max(0, min(e0, e1) - max(s0, s1)) for s1, e1, q1, n1 in ev if q1 != q0 and e1 > s0 and s1 < e0)
        (beside if shared > 0.2 * (e0 - s0) else alone).append((e0 - s0) / 1e3)
    o["rocprof_avg_us_pipelined_launches"] = sum(beside) / len(beside) if beside else None
    o["rocprof_avg_us_standalone_launches"] = sum(alone) / len(alone) if alone else None
    o["rocprof_launches_pipelined_standalone"] = [len(beside), len(alone)]
    o["rocprof_split_note"] = ("kernel trace of the same command; 'pipelined' = a kernel of another queue ran during more than 20 % of the launch (compare with "
                               "bench_hip_event_us), 'standalone' = the rest")
    return o


roof = bench["roofline"]
top = "eftb::" + roof["kernel"].split(" (")[0].split("<")[0]
out = kernel_block(top, roof["ms_per_launch"], roof.get("ms_per_launch_alone"))
out["others"] = []
for o in roof.get("roofline_others", []):
    try:
        out["others"].append(kernel_block("eftb::" + o["kernel"].split(" (")[0].split("<")[0], o["ms_per_launch"]))
    except StopIteration:
        pass
tf = roof.get("roofline_templates_first")
if tf:  # the dominant kernel of the templates-first step (the step `value` was measured on up to round 2), timed alone
    out["templates_first"] = kernel_block("eftb::resum_mfma_kernel", None, tf.get("ms_per_launch_alone"))
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_dominant.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print("evaluations/s", bench["value"], "ms/step", bench["ms_per_step"], "stage_ms", (tf or roof).get("stage_ms"))
