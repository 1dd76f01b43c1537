#!/usr/bin/env python3
"""Where does the host spend a step of the staged direct-P_l loop?  (GPU box)  stage_inputs | run_staged | fetch_previous, mean us per step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, K = 0.7, 128, 60
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
eng.set_latency_mode(False)
eng.set_plk_direct(os.environ.get("HP_DIRECT", "1") == "1")
sets = []
for i in range(8):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
out = np.zeros((K, B, 3, 512))
for depth in (2, 3):
    for rep in range(3):
        ts = np.zeros(3)
        eng.sync()
        t00 = time.perf_counter()
        for i in range(K):
            d = sets[i % 8]
            t0 = time.perf_counter()
            eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
            t1 = time.perf_counter()
            eng.run_staged(mask, B)
            t2 = time.perf_counter()
            if i >= depth:
                eng.fetch_previous("PLK", (B, 3, 512), out=out[i - depth], back=depth)
            t3 = time.perf_counter()
            ts += (t1 - t0, t2 - t1, t3 - t2)
        for back in range(depth - 1, -1, -1):
            eng.fetch_previous("PLK", (B, 3, 512), out=out[K - 1 - back], back=back)
        eng.sync()
        tot = time.perf_counter() - t00
    print(f"depth {depth}: {tot / K * 1e3:.4f} ms/step; host per step: stage {ts[0] / K * 1e6:.0f} us, run {ts[1] / K * 1e6:.0f} us, fetch (wait + copy) {ts[2] / K * 1e6:.0f} us", flush=True)
