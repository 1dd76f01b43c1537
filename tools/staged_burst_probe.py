#!/usr/bin/env python3
"""Staged direct-P_l steps in bursts of G (stage + run G steps back to back, then fetch all of them): the GPU-side rate of the staged path with
the host out of the loop, against the resident loop.  (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, 128
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
eng.set_latency_mode(False)
eng.set_plk_direct(True)
sets = []
for i in range(8):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
for G in (7, 7, 4):
    ts, th = [], []
    for rep in range(6):
        eng.sync()
        t0 = time.perf_counter()
        for i in range(G):
            d = sets[i]
            eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
            eng.run_staged(mask, B)
        t1 = time.perf_counter()
        eng.fetch_previous("PLK", (B, 3, 512), back=0, copy=False)
        t2 = time.perf_counter()
        ts.append((t2 - t0) / G * 1e3)
        th.append((t1 - t0) / G * 1e3)
    print(f"burst of {G}: {min(ts):.4f} ms per step until the last P_l is on the host (host enqueue {min(th):.4f} ms per step)")
