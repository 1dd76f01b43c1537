#!/usr/bin/env python3
"""Where does the staged (H2D + D2H) loop lose against the resident loop?  (GPU box)  K steps each, synchronised before and after:
resident eftb_run | staged without fetching | staged + fetch at depth 1 / 2 / 3 | staged + views (no host copy)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, K = 0.7, 128, int(os.environ.get("HP_K", 20))
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
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
out.fill(0.0)


def resident():
    d = sets[0]
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    for _ in range(K):
        eng.run(mask, B, sync=False)
    eng.sync()


def staged(depth, view=False):
    for i in range(K):
        d = sets[i % 8]
        eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
        eng.run_staged(mask, B)
        if depth is not None and i >= depth:
            if view:
                eng.fetch_previous("PLK", (B, 3, 512), back=depth, copy=False)
            else:
                eng.fetch_previous("PLK", (B, 3, 512), out=out[i - depth], back=depth)
    if depth is not None:
        for back in range(min(depth, K) - 1, -1, -1):
            eng.fetch_previous("PLK", (B, 3, 512), out=out[K - 1 - back], back=back)
    eng.sync()


cases = [("resident eftb_run", resident), ("staged, no fetch", lambda: staged(None)), ("staged, fetch depth 1", lambda: staged(1)),
         ("staged, fetch depth 2", lambda: staged(2)), ("staged, fetch depth 3", lambda: staged(3)), ("staged, views depth 2", lambda: staged(2, True))]
for rep in range(2):
    for name, fn in cases:
        fn()
        eng.sync()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if rep:
            print(f"{name:26s} {dt / K * 1e3:.4f} ms per step  {B * K / dt:9.0f} evaluations/s")
eng.close()
