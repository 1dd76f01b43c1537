#!/usr/bin/env python3
"""One line per run for the library in EFTB_LIB (GPU box; templates-first runs): the resummation kernel alone (back-to-back launches, HIP events), the
resident pipelined loop and the staged loop at fetch depth 2 (ms per step of 128 cosmologies), and a checksum of P_l of the last
staged step (two builds that should agree agree here to ~1e-12)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import _lib as L
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, K = 0.7, 128, int(os.environ.get("HP_K", 40))
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
eng.set_latency_mode(False)
sets = []
for i in range(8):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
out = np.zeros((K, B, 3, 512))


def resident():
    d = sets[0]
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    for _ in range(K):
        eng.run(mask, B, sync=False)
    eng.sync()


def staged(depth=2):
    for i in range(K):
        d = sets[i % 8]
        eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
        eng.run_staged(mask, B)
        if i >= depth:
            eng.fetch_previous("PLK", (B, 3, 512), out=out[i - depth], back=depth)
    for back in range(min(depth, K) - 1, -1, -1):
        eng.fetch_previous("PLK", (B, 3, 512), out=out[K - 1 - back], back=back)
    eng.sync()


def timeit(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        eng.sync()
        t0 = time.perf_counter()
        fn()
        best = min(best, (time.perf_counter() - t0) * 1e3 / K)
    return best


resident()
resident()
alone = min(eng.run_timed(L.K_RESUM, B, 20) for _ in range(3))
r = timeit(resident)
s = timeit(staged)
print(f"{os.environ.get('EFTB_LIB', 'in-tree'):28s} resum alone {alone * 1e3:6.1f} us   resident {r:.4f}   staged {s:.4f} ms/step"
      f"   checksum {np.abs(out[K - 1]).sum():.12e}")
