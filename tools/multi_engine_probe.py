#!/usr/bin/env python3
"""Several engines in one process: does the second one run as fast as the first?  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.tables import EngineConfig

Z, B = 0.7, 128
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
d = synth.draw_batch(B, z=Z)


def loop(eng, tag):
    m = eng.full_mask()
    for _ in range(3):
        eng.run(m, B, sync=False)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.run(m, B, sync=False)
    eng.sync()
    print(f"{tag}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)


e1 = Engine(cfg, max_batch=B)
e1.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
loop(e1, "engine 1")
e2 = Engine(cfg, max_batch=B)
e2.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
loop(e2, "engine 2 (1 alive)")
loop(e1, "engine 1 again")
loop(e2, "engine 2 again")
e1.close()
loop(e2, "engine 2 after closing 1")
e3 = Engine(cfg, max_batch=B)
e3.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
loop(e3, "engine 3 (2 alive)")
loop(e2, "engine 2 again")
# --- which of close() / run_timed() before creating the next engine matters?
from eftpipe_amd import _lib as L

e2.close(); e3.close()
for label, timed, closed in (("after close only", False, True), ("after run_timed + close", True, True), ("after run_timed, no close", True, False)):
    a = Engine(cfg, max_batch=B)
    a.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
    loop(a, f"  first engine [{label}]")
    if timed:
        a.run_timed(L.S_RESUM, B, 3)
    if closed:
        a.close()
    b = Engine(cfg, max_batch=B)
    b.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
    loop(b, f"  next engine  [{label}]")
    b.close()
    if not closed:
        a.close()
