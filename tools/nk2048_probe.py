#!/usr/bin/env python3
"""BASELINE cfg 5 shape on one GPU: Nk = 2048, Nl = 3, IR-resum + AP (+ bias contraction), inputs resident.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z = 0.7
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(2048), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
for B in [int(x) for x in sys.argv[1:]] or [32, 128]:
    eng = Engine(cfg, max_batch=B)
    d = synth.draw_batch(B, z=Z)
    bias = np.stack([bias_row(float(f), [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49], None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
    m = eng.full_mask(reduce=True)
    for _ in range(3):
        eng.run(m, B, sync=False)
    eng.sync()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run(m, B, sync=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"Nk=2048 B={B}: {dt * 1e3:.3f} ms/step -> {B / dt:.0f} evaluations/s", flush=True)
    eng.close()
