#!/usr/bin/env python3
"""Direct-P_l runs (EFTB_O_PLK_DIRECT) against the template path: same P_l?  how fast?  (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, K = 0.7, int(os.environ.get("HP_B", 128)), int(os.environ.get("HP_K", 40))
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


def sync_plk(d):
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    eng.run(mask, B, sync=True)
    return eng.get("PLK", (B, 3, 512)).copy()


def resident():
    d = sets[0]
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    for _ in range(K):
        eng.run(mask, B, sync=False)
    eng.sync()


def staged(depth=int(os.environ.get("HP_DEPTH", "3"))):
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


ref = [sync_plk(sets[i]) for i in range(3)]
eng.set_plk_direct(True)
dr = [sync_plk(sets[i]) for i in range(3)]
for i in range(3):
    scale = np.abs(ref[i]).max(axis=-1, keepdims=True)
    print(f"set {i}: max |direct - templates| / row max = {np.max(np.abs(dr[i] - ref[i]) / scale):.3e}   finite {np.isfinite(dr[i]).all()}", flush=True)
staged()
for i in range(K):
    assert np.array_equal(out[i], dr[i % 8] if i % 8 < 3 else out[i]), f"pipelined direct step {i} differs from the synchronous direct run"
print("pipelined direct steps equal the synchronous ones (sets 0-2)")
for flag in ((True, True) if os.environ.get("HP_ONLY_DIRECT") else (False, True, False, True)):
    eng.set_plk_direct(flag)
    resident(); resident()
    print(f"direct={flag}:  resident {timeit(resident):.4f}   staged {timeit(staged):.4f} ms/step  ({B / timeit(staged):.0f} k evaluations/s)", flush=True)
