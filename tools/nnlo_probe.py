#!/usr/bin/env python3
"""Cost of with_NNLO (the k^4 P11 counter-terms PctNNLOl): steps of 128 resident cosmologies with and without it (GPU box).
The NNLO block rides in the resummation kernel as three more accumulators per lane (resum_mfma_kernel<2, true>); AP and reduce still see
a second template block."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import _lib as L, synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, int(os.environ.get("NNLO_B", 128))
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
d = synth.draw_batch(B, z=Z)
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
res = {}
for nnlo in ((True, False) if os.environ.get("NNLO_FIRST") else (False, True)):
    cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, with_NNLO=nnlo,
                       DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
    eng = Engine(cfg, max_batch=B)
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
    if nnlo:
        eng.put("BIASN", np.full((B, 3), 0.1))
    m = eng.full_mask(reduce=True)
    for _ in range(3):
        eng.run(m, B, sync=False)
    eng.sync()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run(m, B, sync=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    st = {k: round(eng.run_timed(mm, B, 3), 4) for k, mm in (("regroup", L.S_REGROUP), ("resum", L.S_RESUM), ("ap", L.S_AP), ("reduce", L.S_REDUCE))}
    res[nnlo] = dt
    print(f"with_NNLO={nnlo}: {dt * 1e3:.3f} ms per step of {B} -> {B / dt:.0f} evaluations/s; stages alone (ms) {st}", flush=True)
    eng.close()
print(f"NNLO overhead: {100 * (res[True] / res[False] - 1):.1f} % per step")
