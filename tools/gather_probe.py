#!/usr/bin/env python3
"""Per-step cost of the asynchronous P_l gather (single rank, self exchange through RCCL) on top of the step itself.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine, comm_unique_id
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, 128
eng = Engine(EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z))), max_batch=B)
d = synth.draw_batch(B, z=Z)
bias = np.stack([bias_row(float(f), [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49], None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
mask = eng.full_mask(reduce=True)
eng.comm_init(1, 0, comm_unique_id())
for gather in (False, True, False, True):
    for _ in range(3):
        eng.run(mask, B, sync=False)
        if gather:
            eng.gather_plk(B, root=0)
    eng.sync()
    n = 40
    th = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run(mask, B, sync=False)
        if gather:
            ta = time.perf_counter()
            eng.gather_plk(B, root=0)
            th += time.perf_counter() - ta
    t1 = time.perf_counter()
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"gather={gather}: {dt * 1e3:.3f} ms/step -> {B / dt:.0f} evaluations/s   (host: enqueue loop {(t1 - t0) / n * 1e3:.3f} ms/step, of which gather calls {th / n * 1e3:.3f})", flush=True)
eng.close()
