#!/usr/bin/env python3
"""PCIe-inclusive rates of the cfg-2 workload (host inputs in, results out), B = 128 walkers per call.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.marginal import MarginalLikelihood, data_index
from eftpipe_amd.parambasis import bias_row, gaussian_rows
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, 128
k = synth.survey_kgrid(512)
cfg = EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B)
d = synth.draw_batch(B, z=Z)
BS = [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49]
bias = np.stack([bias_row(float(f), BS, None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
rows = np.stack([gaussian_rows(float(f), (2.14, 0.55, 0.55), None, 0.7, 0.25, 4.5e-5) for f in d["f"]])
sel = np.arange(7, 512)[(k[7:] <= 0.2)][::8]
index = np.concatenate([l * 512 + sel for l in range(3)]).astype(np.int32)
templ, plk = eng.eval_batch(d["Pin"], d["f"], d["DA"], d["H"], bias=bias)
data = plk.reshape(B, -1)[0, index] * 1.01
sig = 0.05 * np.abs(data) + 10.0
like = MarginalLikelihood(eng, index, data, np.diag(1 / sig**2), np.zeros(7), np.full(7, 2.0))
pinned = eng.pinned_empty(templ.shape)


def rate(fn, n=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e3, B / dt


args = (d["Pin"], d["f"], d["DA"], d["H"])
for name, fn in (("templates + P_l -> pageable host", lambda: eng.eval_batch(*args, bias=bias)),
                 ("templates + P_l -> pinned host", lambda: eng.eval_batch(*args, bias=bias, out=pinned)),
                 ("P_l only", lambda: eng.eval_batch(*args, bias=bias, templates=False)),
                 ("marginalised ln P only (eftb_eval_logp_batch)", lambda: like.eval_logp(*args, rows))):
    ms, r = rate(fn)
    print(f"{name:48s} {ms:7.3f} ms per {B} -> {r:9.0f} evaluations/s", flush=True)
# pipelined: stage the next step, launch, fetch the previous step's results -- new inputs every step, nothing waits for the step in flight
from eftpipe_amd import _lib as L

draws = [synth.draw_batch(B, z=Z, seed=100 + i) for i in range(4)]
for name, mask, out, shape, kw in (("pipelined, P_l back (eftb_stage_inputs / run_staged / fetch_previous)", eng.full_mask(reduce=True), "PLK", (B, 3, 512), dict(bias=bias)),
                                  ("pipelined, marginalised ln P back", eng.full_mask() | L.S_LOGP, "LOGP", (B, 26), dict(rows=rows))):
    n = 40
    dd = draws[0]
    eng.stage_inputs(dd["Pin"], dd["f"], dd["DA"], dd["H"], **kw)
    eng.run_staged(mask, B)
    t0 = time.perf_counter()
    for i in range(1, n + 1):
        dd = draws[i % 4]
        eng.stage_inputs(dd["Pin"], dd["f"], dd["DA"], dd["H"], **kw)
        eng.run_staged(mask, B)
        res = eng.fetch_previous(out, shape)
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    assert np.all(np.isfinite(res[:, :2]))
    print(f"{name:48s} {dt * 1e3:7.3f} ms per {B} -> {B / dt:9.0f} evaluations/s", flush=True)
eng.close()
