#!/usr/bin/env python3
"""Several engines created one after another in one process (a Cobaya run holds one engine per tracer): does a later engine run as fast as
the first?  usage: engine_seq_probe.py [n_engines] ; KEEP=1 keeps the earlier engines alive."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig
Z, B = 0.7, 128
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
d = synth.draw_batch(B, z=Z)
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
keep, alive = bool(os.environ.get("KEEP")), []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng = Engine(cfg, max_batch=B)
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
    m = eng.full_mask(reduce=True)
    for _ in range(3):
        eng.run(m, B, sync=False)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.run(m, B, sync=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / 30
    print(f"engine {i}: {dt * 1e3:.3f} ms per step -> {B / dt:.0f} evaluations/s", flush=True)
    if keep:
        alive.append(eng)
    else:
        eng.close()
