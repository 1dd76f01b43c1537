#!/usr/bin/env python3
"""How long does the GPU stay in its sustained state?  (GPU box)  The pipelined, coalescing direct-P_l loop of bench.py: a continuous run of 600 steps,
an idle gap of G ms, then 20 timed steps -- per-step time against G.  (bench.py's warmup_note / bare_warmup_* keys: right after idle the same loop
runs 10-25 % slower.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

B, DEPTH = 128, 12
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, 0.7)), H_AP=float(synth.hubble(synth.OM_AP, 0.7)))
eng = Engine(cfg, max_batch=B, coalesce=4)
eng.set_latency_mode(False)
eng.set_plk_direct(True)
sets = []
for i in range(16):
    d = synth.draw_batch(B, z=0.7, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), bench.BS, None, bench.ES, kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
out = eng.pinned_empty((16, B, 3, 512))


def loop(n):
    for i in range(n):
        d = sets[i % 16]
        eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=DEPTH if i >= DEPTH else -1, shape=(B, 3, 512), out=out[i % 16])
    eng.flush()
    for back in range(min(DEPTH, n) - 1, -1, -1):
        eng.fetch_previous("PLK", (B, 3, 512), back=back, copy=False)
    eng.sync()


loop(50)
for gap_ms in (0, 0.2, 1, 3, 10, 30, 100, 300, 1000, 0):
    res = []
    for rep in range(3):
        loop(600)
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        t0 = time.perf_counter()
        loop(20)
        res.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f"idle {gap_ms:6.1f} ms -> 20 steps at " + " ".join(f"{x:6.1f}" for x in res) + " us per step", flush=True)
eng.close()
