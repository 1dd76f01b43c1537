#!/usr/bin/env python3
"""Does splitting the batch over several engines (= HIP streams) of one GPU help?  The latency-bound kernels of one lane
(loop path, AP) can overlap the FP64-pipe-bound resum kernel of another.  GPU box.  usage: lanes_probe.py [lanes ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, int(os.environ.get('LANES_TOTAL', 128))
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
d = synth.draw_batch(B, z=Z)
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
for lanes in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    b = B // lanes
    engs = [Engine(cfg, max_batch=b) for _ in range(lanes)]
    for i, e in enumerate(engs):
        sl = slice(i * b, (i + 1) * b)
        e.load_inputs(d["Pin"][sl], d["f"][sl], d["DA"][sl], d["H"][sl], bias[sl])
    m = engs[0].full_mask(reduce=True)
    for _ in range(3):
        for e in engs:
            e.run(m, b, sync=False)
    for e in engs:
        e.sync()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        for e in engs:
            e.run(m, b, sync=False)
    for e in engs:
        e.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"lanes={lanes} batch/lane={b}: {dt * 1e3:.3f} ms per {B} cosmologies -> {B / dt:.0f} evaluations/s", flush=True)
    for e in engs:
        e.close()
