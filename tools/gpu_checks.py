"""One-off GPU measurements quoted in DESIGN.md: single-evaluation latency, PCIe-inclusive batch rate."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z = 0.7
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
out = {}
t0 = time.perf_counter()
eng = Engine(cfg, max_batch=128)
out["engine_init_s"] = time.perf_counter() - t0
d = synth.draw_batch(128, z=Z)
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
for B in (1, 8, 128):
    sl = slice(0, B)
    eng.eval_batch(d["Pin"][sl], d["f"][sl], d["DA"][sl], d["H"][sl], bias=bias[sl])
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.eval_batch(d["Pin"][sl], d["f"][sl], d["DA"][sl], d["H"][sl], bias=bias[sl])
    dt = (time.perf_counter() - t0) / n
    out[f"host_in_out_B{B}"] = dict(ms_per_call=dt * 1e3, evals_per_s=B / dt)
    eng.load_inputs(d["Pin"][sl], d["f"][sl], d["DA"][sl], d["H"][sl], bias[sl])
    m = eng.full_mask(reduce=True)
    eng.run(m, B)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run(m, B, sync=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    out[f"device_resident_B{B}"] = dict(ms_per_call=dt * 1e3, evals_per_s=B / dt)
print(json.dumps(out, indent=1))
