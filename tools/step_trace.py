#!/usr/bin/env python3
"""Undistorted timeline of the pipelined, coalescing direct-P_l loop (GPU box): EFTB_O_STEP_TRACE timing events at nine points of every launch
(rocprofv3 doubles the host's launch cost -- under it the host sets the pace).  Prints, for the last launches of a 90-step loop, when each
stage of each launch started / ended (us, relative to the first printed launch's upload) and the gaps between dependent stages."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig

Z, B, K, DEPTH, CO = 0.7, 128, int(os.environ.get("ST_K", 90)), int(os.environ.get("ST_DEPTH", 12)), int(os.environ.get("ST_COALESCE", 4))
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
eng = Engine(cfg, max_batch=B, coalesce=CO)
eng.set_latency_mode(False)
eng.set_plk_direct(True)
sets = []
for i in range(8):
    d = synth.draw_batch(B, z=Z, seed=100 + i)
    d["bias"] = np.stack([bias_row(float(f), [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49], None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)


def loop(n):
    for i in range(n):
        d = sets[i % 8]
        eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=DEPTH if i >= DEPTH else -1, shape=(B, 3, 512))
    for back in range(min(DEPTH, n) - 1, -1, -1):
        eng.fetch_previous("PLK", (B, 3, 512), back=back, copy=False)
    eng.sync()


loop(30)
t0 = time.perf_counter()
loop(K)
plain = (time.perf_counter() - t0) / K
eng.step_trace(True)
t0 = time.perf_counter()
loop(K)
traced = (time.perf_counter() - t0) / K
tr = eng.step_trace()
print(f"ms per step: {plain * 1e3:.4f} without the trace events, {traced * 1e3:.4f} with them; {len(tr)} launches kept")
names = ["upload", "front>", "synth<", "prep>", "resum<", "resum>", "spline<", "AP>", "copy>"]
sel = tr[len(tr) // 2 - 6: len(tr) // 2 + 6] if len(tr) > 14 else tr   # (a short loop, ST_K=20: every launch, i.e. fill and drain)
base = sel[0][2]
print("launch  B " + " ".join(f"{n:>8s}" for n in names) + "   | front synth+prep resum  AP  copy | gaps: front->synth prep->resum resum->spline")
for r in sel:
    t = r[2:] - base
    print(f"{int(r[0]):5d} {int(r[1]):4d} " + " ".join(f"{x:8.1f}" for x in t[:9]) + f" [front: upload {t[9] - t[0]:4.0f} rows+gemm {t[10] - t[9]:4.0f} antidiag {t[11] - t[10]:4.0f} build {t[1] - t[11]:4.0f}]" +
          f"   | {t[1] - t[0]:5.0f} {t[3] - t[2]:5.0f} {t[5] - t[4]:5.0f} {t[7] - t[6]:5.0f} {t[8] - t[7]:5.0f} | {t[2] - t[1]:6.0f} {t[4] - t[3]:6.0f} {t[6] - t[5]:6.0f}")
if len(tr) < 8:
    print("first upload to last copy-out: %.0f us for %d steps" % (tr[-1, 10] - tr[0, 2], K))
    eng.close()
    sys.exit(0)
per = np.diff(tr[:, 10])  # copy-out end
print("launch period (copy-out end to copy-out end), us: mean %.0f, steps per launch %.2f" % (per[5:].mean(), tr[5:, 1].mean() / B))
eng.close()
