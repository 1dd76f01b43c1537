#!/usr/bin/env python3
"""The dependent-sampler loop of bench.py on its own (GPU box): stage -> run -> fetch the step just launched, nothing queued behind it.
Prints ms per step and the host time of each call; run under rocprofv3 --kernel-trace for the device timeline (tools/trace_sync.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, N = 0.7, int(os.environ.get("HP_B", 128)), int(os.environ.get("HP_N", 30))
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
sets = []
for i in range(6):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
out = np.zeros((B, 3, 512))
for rep in range(2):
    t = {"stage": 0.0, "run": 0.0, "fetch": 0.0}
    t0 = time.perf_counter()
    for i in range(N):
        d = sets[i % 6]
        a = time.perf_counter()
        eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
        b = time.perf_counter()
        eng.run_staged(mask, B)
        c = time.perf_counter()
        if os.environ.get("HP_COPY"):
            eng.fetch_previous("PLK", (B, 3, 512), out=out, back=0)
        else:
            view = eng.fetch_previous("PLK", (B, 3, 512), back=0, copy=False)
        e = time.perf_counter()
        t["stage"] += b - a
        t["run"] += c - b
        t["fetch"] += e - c
    tot = time.perf_counter() - t0
print(f"sync step B={B}: {tot / N * 1e3:.3f} ms per step = {B * N / tot:.0f} evaluations/s; host: stage_inputs {t['stage'] / N * 1e6:.0f} us, "
      f"run_staged {t['run'] / N * 1e6:.0f} us, fetch(0) {t['fetch'] / N * 1e6:.0f} us (wait + copy)")
eng.close()
