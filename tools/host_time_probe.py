#!/usr/bin/env python3
"""Host time per call of the pipelined sampler loop (GPU box): is the loop bound by the host enqueueing, or by the GPU?
stage_inputs and run_staged never wait for the GPU, so their wall time is pure host cost; fetch_previous(back=3) = wait + copy."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, N = 0.7, int(os.environ.get("HP_B", 128)), 60
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
sets = []
for i in range(6):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
comm = bool(os.environ.get("HP_COMM"))
if comm:  # one-rank RCCL self exchange: the per-step host cost of the multi-GPU loop on the root
    from eftpipe_amd.engine import comm_unique_id
    eng.comm_init(1, 0, comm_unique_id())
mask = eng.full_mask(reduce=True)
out = np.empty((B, 3, 512))
t = {"stage": 0.0, "run": 0.0, "gather": 0.0, "fetch": 0.0}
for rep in range(2):
    for k in t:
        t[k] = 0.0
    t0 = time.perf_counter()
    for i in range(N):
        d = sets[i % 6]
        a = time.perf_counter()
        eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
        b = time.perf_counter()
        eng.run_staged(mask, B)
        c = time.perf_counter()
        if comm:
            eng.gather_plk(B, root=0)
        g_ = time.perf_counter()
        if i > 2:
            if comm:
                out[:] = eng.fetch_gathered(B, back=2, copy=False)[0]
            else:
                eng.fetch_previous("PLK", (B, 3, 512), out=out, back=3)
        e = time.perf_counter()
        t["stage"] += b - a
        t["run"] += c - b
        t["gather"] += g_ - c
        t["fetch"] += e - g_
    eng.sync()
    tot = time.perf_counter() - t0
print(f"B={B}: {tot / N * 1e3:.3f} ms per step; host per step: stage_inputs {t['stage'] / N * 1e6:.0f} us, run_staged {t['run'] / N * 1e6:.0f} us, "
      f"gather_plk {t['gather'] / N * 1e6:.0f} us, fetch {t['fetch'] / N * 1e6:.0f} us (wait + 1.5 MB copy)")
if comm:
    eng.close()
    sys.exit(0)
# the copy alone: fetch of a step that finished long ago
eng.sync()
a = time.perf_counter()
for _ in range(20):
    eng.fetch_previous("PLK", (B, 3, 512), out=out, back=1)
print(f"fetch of a finished step (copy only): {(time.perf_counter() - a) / 20 * 1e6:.0f} us")
eng.close()
