#!/usr/bin/env python3
"""Per-kernel resource usage and inner-loop instruction counts from the compiled gfx950 assembly.

`bench.py` prices the dominant kernel's EXECUTED FP64 work from these numbers (MFMA, v_fma_f64, v_mul/add_f64 per trip of the
s loop) instead of a hand-typed constant; `__graft_entry__.build()` regenerates the file next to the sources
(eftpipe_amd/csrc/isa_counts.json) every time the library is built, so the counts always describe the code that runs.

    python tools/isa_counts.py [--out eftpipe_amd/csrc/isa_counts.json] [--asm /tmp/x.s]

Method: `hipcc -S --cuda-device-only` of eftbird.hip; for every kernel the text between its label and `s_endpgm`/`.Lfunc_end`;
loops are the blocks LLVM annotates "Loop Header" -- the body runs from the header label to the last branch back to it.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "eftpipe_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FLOPS_PER_LANE = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_ldexp_f64": 0, "v_rcp_f64": 1, "v_sqrt_f64": 1,
                  "v_div_fmas_f64": 2, "v_div_fixup_f64": 1, "v_div_scale_f64": 1, "v_min_f64": 1, "v_max_f64": 1}
MFMA_FLOPS = {"v_mfma_f64_16x16x4_f64": 2 * 16 * 16 * 4, "v_mfma_f64_16x16x4f64": 2 * 16 * 16 * 4, "v_mfma_f64_4x4x4_4b_f64": 2 * 4 * 4 * 4 * 4,
              "v_mfma_f64_4x4x4f64": 2 * 4 * 4 * 4 * 4}


def source_hash():
    """first 16 hex digits of sha256(eftbird.hip + eftb_kernels.hpp): the same value csrc/Makefile compiles into the library"""
    import hashlib

    h = hashlib.sha256()
    for f in ("eftbird.hip", "eftb_kernels.hpp"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def emit_asm(path):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function", "-Wno-unused-command-line-argument",
           "-S", "--cuda-device-only", "-o", path, os.path.join(CSRC, "eftbird.hip")]
    subprocess.run(cmd, check=True, capture_output=True)


def demangle(name):
    m = re.match(r"_ZN4eftb(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    start = m.end()
    base = name[start:start + n]
    rest = name[start + n:]
    t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
    if t:
        base += "<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
    return base


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_") and op.endswith("_f64") or op in FLOPS_PER_LANE:
        return "valu_f64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def count_block(lines):
    c = {"mfma": 0, "valu_f64": 0, "valu_other": 0, "vmem": 0, "lds": 0, "smem": 0, "salu": 0, "wait": 0, "other": 0}
    ops = {}
    flops_valu_lane = 0
    flops_mfma = 0
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.endswith("_e32") or op.endswith("_e64"):
            op = op[:-4]
        c[classify(op)] += 1
        if op.startswith("v_") and (op.endswith("_f64") or op.startswith("v_mfma")):
            ops[op] = ops.get(op, 0) + 1
        if op in MFMA_FLOPS:
            flops_mfma += MFMA_FLOPS[op]
        elif op in FLOPS_PER_LANE:
            flops_valu_lane += FLOPS_PER_LANE[op]
    c["f64_ops"] = ops
    c["flops_per_wave_trip"] = flops_mfma + 64 * flops_valu_lane
    c["mfma_flops_per_wave_trip"] = flops_mfma
    return c


def analyse(asm_text):
    out = {}
    lines = asm_text.splitlines()
    # kernel bodies
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) if (m := re.match(r"^(_ZN4eftb\w+):", ln))]
    for idx, (i0, name) in enumerate(starts):
        i1 = next((j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        body = lines[i0:i1]
        info = {}
        # resource comments emitted after the body
        tail = lines[i1:i1 + 80]
        for ln in tail:
            for key, pat in (("vgprs", r"; NumVgprs: (\d+)"), ("agprs", r"; NumAgprs: (\d+)"), ("sgprs", r"; NumSgprs: (\d+)"),
                             ("scratch_bytes", r"; ScratchSize: (\d+)"), ("lds_bytes", r"; LDSByteSize: (\d+)"), ("occupancy", r"; Occupancy: (\d+)"),
                             ("total_vgprs", r"; TotalNumVgprs: (\d+)")):
                m = re.search(pat, ln)
                if m and key not in info:
                    info[key] = int(m.group(1))
        loops = []
        for j, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)", ln)
            if not m:
                continue
            label, depth = m.group(1), int(m.group(2))
            last = None
            for q in range(j + 1, len(body)):
                s = body[q].strip()
                if s.startswith(("s_cbranch", "s_branch")) and s.split()[-1] == label:
                    last = q
            if last is None:
                continue
            blk = count_block(body[j + 1:last + 1])
            blk["label"], blk["depth"] = label, depth
            loops.append(blk)
        info["loops"] = loops
        info["whole"] = {k: v for k, v in count_block(body).items() if k != "f64_ops"}
        out[demangle(name)] = info
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(CSRC, "isa_counts.json"))
    ap.add_argument("--asm", default=None, help="existing assembly file (default: compile eftbird.hip)")
    ap.add_argument("--show", default=None, help="print the entry of this kernel")
    a = ap.parse_args()
    if a.asm:
        text = open(a.asm).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            p = os.path.join(tmp, "eftbird_gfx950.s")
            emit_asm(p)
            text = open(p).read()
    res = analyse(text)
    res["_source_hash"] = source_hash()
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    if a.show:
        print(json.dumps(res.get(a.show), indent=1))
    else:
        for k, v in sorted(res.items()):
            if k.startswith("_"):
                continue
            big = max(v["loops"], key=lambda b: b["mfma"] + b["valu_f64"], default=None)
            print(f"{k:44s} vgpr {v.get('vgprs', '?'):>4} agpr {v.get('agprs', '?'):>3} scratch {v.get('scratch_bytes', '?'):>4} lds {v.get('lds_bytes', '?'):>6}"
                  + (f"  hot loop: {big['mfma']} mfma, {big['valu_f64']} f64 valu, {big['valu_other']} other valu, {big['vmem']} vmem, {big['smem']} smem" if big else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
