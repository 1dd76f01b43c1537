#!/usr/bin/env python3
"""Diagnosis of the two-pass with_NNLO path (EFTB_NNLO_INLINE=0): per-run wall time, synchronous and queued."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import _lib as L, synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig
Z, B = 0.7, 128
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
d = synth.draw_batch(B, z=Z)
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, with_NNLO=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
eng.put("BIASN", np.full((B, 3), 0.1))
m = eng.full_mask(reduce=True)
for _ in range(3):
    eng.run(m, B, sync=True)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); eng.run(m, B, sync=True); ts.append(time.perf_counter() - t0)
print("sync runs (ms):", [round(t * 1e3, 3) for t in ts], flush=True)
for n in (1, 2, 4, 8, 30):
    eng.sync(); t0 = time.perf_counter()
    for _ in range(n):
        eng.run(m, B, sync=False)
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print(f"queued {n}: enqueue {(t1 - t0) / n * 1e3:.3f} ms per run, total {(t2 - t0) / n * 1e3:.3f} ms per run", flush=True)
for name, mm in (("front", L.S_PREP | L.S_LOOPS | L.S_CF), ("regroup", L.S_REGROUP), ("resum", L.S_RESUM), ("ap", L.S_AP), ("reduce", L.S_REDUCE)):
    print(name, round(eng.run_timed(mm, B, 5), 4), flush=True)
eng.close()
