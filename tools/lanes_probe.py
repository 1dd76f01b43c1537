"""Experiment: do two half-batch engines on independent streams overlap MFMA-bound and VALU-bound stages?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z = 0.7
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
def setup(B, n):
    engs = []
    for i in range(n):
        e = Engine(cfg, max_batch=B)
        d = synth.draw_batch(B, z=Z, seed=100 + i)
        bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
        e.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias)
        engs.append(e)
    return engs
def bench(engs, B, steps=20):
    m = engs[0].full_mask(reduce=True)
    for e in engs: e.run(m, B, sync=False)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e in engs: e.run(m, B, sync=False)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    return len(engs) * B * steps / dt
for total, n in ((128, 1), (128, 2), (128, 4), (256, 1), (256, 2), (256, 4), (512, 4)):
    engs = setup(total // n, n)
    print(f"total batch {total} as {n} lane(s): {bench(engs, total // n):.0f} evals/s")
    for e in engs: e.close()
