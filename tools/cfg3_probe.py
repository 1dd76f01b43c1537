#!/usr/bin/env python3
"""BASELINE cfg 3 on one GPU: three tracers per likelihood point (LRG, ELG chained, X cross), Nk = 512, window + with_interp data k,
joint marginalised log-posterior.  Prints likelihood points per second, inputs resident (run + LOGP) and PCIe-inclusive
(eftb_eval_logp_batch).  GPU box.  Production window settings of the reference's yamls (accboost 4, windowk 0.1) with the per-tracer
DR16 windows win_NGC_{LRG,ELG,X}; the window matrices come from the device precompute (eftb_window_precompute)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eftpipe_amd import _lib as L
from eftpipe_amd import synth
from eftpipe_amd import tables as TB
from eftpipe_amd.engine import Engine
from eftpipe_amd.marginal import MarginalLikelihood, data_index
from eftpipe_amd.parambasis import gaussian_rows
from eftpipe_amd.tables import EngineConfig

NW, NTR, NK = int(os.environ.get("CFG3_NW", 42)), 3, 512
ZS = (0.696, 0.849, 0.763)
k = synth.survey_kgrid(NK)
cfg = EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, APst=True, DA_AP=float(synth.da_func(synth.OM_AP, 0.7)), H_AP=float(synth.hubble(synth.OM_AP, 0.7)))
eng = Engine(cfg, max_batch=NW * NTR)
from eftpipe_amd.window import window_matrix_device

kdata = np.arange(0.02, 0.2, 0.005)
interp = TB.interp_operator(k, kdata)
nd = kdata.size
ops_host = []
for name, chained in (("LRG", False), ("ELG", True), ("X", False)):
    tab = np.load(os.path.join(ROOT, "tests", "golden", f"win_NGC_{name}_sQ024.npy"))
    timing = {}
    _, p, _, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, windowk=0.1, accboost=4, timing=timing)
    print(f"window {name}: Np = {p.size}, host tables {timing['host_tables_s']:.2f} s, device kernels {timing['device_kernels_ms']:.2f} ms")
    op = np.zeros((3, 3, nd, NK))
    o = TB.compose_operator(3, NK, Wfold=Wfold, binning=interp, chained=chained)
    op[: o.shape[0]] = o
    ops_host.append(op)
ops = [eng.add_operator(o) for o in ops_host]
eng.set_tracers(NTR, ops)
Pin, f, DA, H = [], [], [], []
for w in range(NW):
    for t, z in enumerate(ZS):
        d = synth.draw_batch(1, z=z, seed=1000 + w)
        Pin.append(d["Pin"][0]); f.append(d["f"][0])
        # per-tracer AP fiducial through the engine's single one (include/eftbird.h eftb_set_tracers)
        DA.append(d["DA"][0] * cfg.DA_AP / synth.da_func(synth.OM_AP, z)); H.append(d["H"][0] * cfg.H_AP / synth.hubble(synth.OM_AP, z))
Pin, f, DA, H = np.stack(Pin), np.array(f), np.array(DA), np.array(H)
ngL, ngE = (2.1, 0.5, 0.3), (1.3, -0.2, 0.6)
nG = 17
rows = np.zeros((NW * NTR, nG + 1, 24))
for w in range(NW):
    eL, eE, eX = w * NTR, w * NTR + 1, w * NTR + 2
    rL, rE = gaussian_rows(f[eL], ngL, None, 0.7, 0.25, 4.5e-5), gaussian_rows(f[eE], ngE, None, 0.7, 0.25, 2.3e-4)
    rX = gaussian_rows(f[eX], ngL, ngE, 0.7, 0.25, 4.5e-5, 0.7, 0.25, 2.3e-4)
    rows[eL, 0], rows[eL, 1:8] = rL[0], rL[1:]
    rows[eE, 0], rows[eE, 8:15] = rE[0], rE[1:]
    rows[eX, 0], rows[eX, 1:5], rows[eX, 8:12], rows[eX, 15:18] = rX[0], rX[1:5], rX[5:9], rX[9:12]
index = np.concatenate([data_index([0, 2, 4], None, nd, tracer=0, nl=3), data_index([0, 2], None, nd, tracer=1, nl=3),
                        data_index([0, 2, 4], None, nd, tracer=2, nl=3)])
templ = eng.eval_batch(Pin, f, DA, H)
model = np.concatenate([np.einsum("r,lrx->lx", rows[t, 0], templ[t]) for t in range(NTR)]).reshape(-1)[index]
sig = 0.05 * np.abs(model) + 10.0
like = MarginalLikelihood(eng, index, model * 1.01, np.diag(1.0 / sig**2), np.zeros(nG), np.full(nG, 2.0))
lp = like.eval_logp(Pin, f, DA, H, rows)
assert np.all(np.isfinite(lp))
B = NW * NTR
eng.put("GROWS", np.concatenate([rows, np.zeros((B, 25 - (nG + 1), 24))], axis=1))
mask = eng.full_mask() | L.S_LOGP
for _ in range(3):
    eng.run(mask, B, sync=False)
eng.sync()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    eng.run(mask, B, sync=False)
eng.sync()
dt = (time.perf_counter() - t0) / n
print(f"cfg 3 (3 tracers / point, {index.size} data points, {nG} marginalised parameters), {NW} points per step:")
print(f"  inputs resident : {dt * 1e3:.3f} ms/step -> {NW / dt:9.0f} likelihood points/s ({B / dt:9.0f} theory evaluations/s)")
theory = eng.full_mask() & ~L.S_PROJECT
print("  stage ms: theory %.3f, project %.3f, logp %.3f" % (eng.run_timed(theory, B, 5), eng.run_timed(theory | L.S_PROJECT, B, 5) - eng.run_timed(theory, B, 5),
                                                          eng.run_timed(L.S_LOGP, B, 5)))
t0 = time.perf_counter()
for _ in range(n):
    like.eval_logp(Pin, f, DA, H, rows)
dt = (time.perf_counter() - t0) / n
print(f"  PCIe-inclusive  : {dt * 1e3:.3f} ms/step -> {NW / dt:9.0f} likelihood points/s")
eng.close()
