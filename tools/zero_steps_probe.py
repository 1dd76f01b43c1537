#!/usr/bin/env python3
"""Fixed cost of the pair-GEMM launches: time them with every K slice emptied (plans with zero steps), i.e. tile staging,
linear terms, reduction/expansion epilogue and workgroup turnaround only.  GPU box."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import engine as E, synth, _lib as L
from eftpipe_amd.tables import EngineConfig

zero = "--zero" in sys.argv
for a in sys.argv[1:]:
    if a.startswith("--lib="):
        L.LIB_PATH = os.path.abspath(a[6:])
        print("using", L.LIB_PATH)
if zero:
    w4, w16 = E.wave_plan_2run, E.wave_plan
    def z4(s, n):
        p = w4(s, n); p[:, 3] = 0; return p
    def z16(s, n=8):
        p = w16(s, n); p[:, 3] = 0; return p
    E.wave_plan_2run, E.wave_plan = z4, z16
    _sp = E.split_plans
    E.split_plans = lambda steps, nw, plan=None: _sp(steps, nw, z4 if plan is w4 else z16)
B = 128
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True,
                   DA_AP=float(synth.da_func(synth.OM_AP, 0.7)), H_AP=float(synth.hubble(synth.OM_AP, 0.7)))
eng = E.Engine(cfg, max_batch=B)
d = synth.draw_batch(B, z=0.7)
eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
eng.run(L.S_PREP, B)
for name, m in (("K_P22", L.K_P22), ("LOOPS", L.S_LOOPS), ("K_C22", L.K_C22), ("CF", L.S_CF)):
    eng.run_timed(m, B, 2)
    print("zero" if zero else "full", name, round(eng.run_timed(m, B, 10) * 1e3, 1), "us")
eng.close()
