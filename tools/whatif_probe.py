#!/usr/bin/env python3
"""What does each kernel of a direct-P_l step cost the pipelined, coalescing loop?  (GPU box; needs `make -C eftpipe_amd/csrc whatif`.)
Runs the staged loop of bench.py (depth 12, coalesce 4, 128 per step) once per entry of SKIPS in a fresh child process with
EFTB_LIB=libeftbird_whatif.so and EFTB_WHATIF_SKIP set: the named kernels are not launched (the results of those runs are garbage; only the time
per step is read).  The difference to the full run is what the kernel costs the step -- against its stand-alone duration (tools/kstat_direct.sh)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {1: "prep_rows_qf", 2: "gemm_direct (first stage)", 4: "antidiag", 8: "build_rows_plk", 16: "synth", 32: "back_prep_plk", 64: "resum_plk", 128: "spline",
         256: "ap_plk_fused", 512: "the copy-out of P_l (DMA; EFTB_PLK_DMA=0: copy16_kernel)", 1024: "stage_gather", 2048: "the PCIe leg of copy16_kernel (device destination)"}


def child():
    import numpy as np

    sys.path.insert(0, ROOT)
    from eftpipe_amd import synth
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    Z, B, K, DEPTH = 0.7, 128, int(os.environ.get("WI_K", 200)), 12
    cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
    eng = Engine(cfg, max_batch=B, coalesce=4)
    eng.set_latency_mode(False)
    eng.set_plk_direct(True)
    sets = []
    for i in range(8):
        d = synth.draw_batch(B, z=Z, seed=100 + i)
        d["bias"] = np.stack([bias_row(float(f), [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49], None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
        sets.append(d)
    mask = eng.full_mask(reduce=True)

    def loop(n):
        for i in range(n):
            d = sets[i % 8]
            eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=DEPTH if i >= DEPTH else -1, shape=(B, 3, 512))
        for back in range(min(DEPTH, n) - 1, -1, -1):
            eng.fetch_previous("PLK", (B, 3, 512), back=back, copy=False)
        eng.sync()

    loop(30)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        loop(K)
        best = min(best, (time.perf_counter() - t0) / K)
        if os.environ.get("WI_VERBOSE"):
            print(f"pass: {(time.perf_counter() - t0) / K * 1e6:.1f} us per step", file=sys.stderr)
    print(json.dumps({"ms_per_step": best * 1e3}))
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    skips = [int(x) for x in os.environ["WI_SKIPS"].split(",")] if os.environ.get("WI_SKIPS") else [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 2048, 1 | 2 | 4 | 8, 16 | 32, 128 | 256, 0]
    base = None
    for sk in skips:
        env = dict(os.environ, EFTB_LIB=os.path.join(ROOT, "eftpipe_amd", "libeftbird_whatif.so"), EFTB_WHATIF_SKIP=str(sk))
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        try:
            ms = json.loads(res.stdout.strip().splitlines()[-1])["ms_per_step"]
        except Exception:
            print("skip", sk, "FAILED", res.stderr[-500:], flush=True)
            continue
        if sk == 0 and base is None:
            base = ms
        what = " + ".join(n for b, n in NAMES.items() if sk & b) or "(nothing skipped)"
        print(f"skip {sk:5d}  {ms * 1e3:7.1f} us per step   {(base - ms) * 1e3:+7.1f} us   without {what}", flush=True)
