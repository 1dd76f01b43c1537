#!/usr/bin/env python3
"""Do two pipelined engines in flight fill the GPU better than one?  (GPU box)  Resident inputs, B = 128 each, steps alternate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, K = 0.7, 40
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]


def make(B):
    cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
    e = Engine(cfg, max_batch=B)
    e.set_latency_mode(False)
    d = synth.draw_batch(B, z=Z, seed=100)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    e.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    return e


def run(engs, B, tag):
    m = engs[0].full_mask(reduce=True)
    best = 1e9
    for rep in range(4):
        for e in engs:
            e.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            for e in engs:
                e.run(m, B, sync=False)
        for e in engs:
            e.sync()
        best = min(best, time.perf_counter() - t0)
    n = K * len(engs) * B
    print(f"{tag}: {best / K * 1e3:.4f} ms per round, {n / best / 1e3:.1f} k evaluations/s", flush=True)


for B in (128, 64):
    a = make(B)
    run([a], B, f"one engine  B={B}")
    b = make(B)
    run([a, b], B, f"two engines B={B} each")
    run([a], B, f"one engine  B={B} (second alive)")
    c = make(B)
    run([a, b, c], B, f"three engines B={B} each")
    a.close(); b.close(); c.close()
