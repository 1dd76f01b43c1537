import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from eftpipe_amd import synth, _lib as L
from eftpipe_amd.engine import Engine
from eftpipe_amd.tables import EngineConfig
Z=0.7; B=128
for Nl in (2,3):
    cfg = EngineConfig(Nl=Nl, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
    eng = Engine(cfg, max_batch=B)
    d = synth.draw_batch(B, z=Z)
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"])
    m = eng.full_mask()
    for _ in range(3): eng.run(m, B, sync=False)
    eng.sync(); t0=time.perf_counter()
    for _ in range(20): eng.run(m, B, sync=False)
    eng.sync(); dt=(time.perf_counter()-t0)/20
    st = {n: round(eng.run_timed(mm, B, 3),3) for n, mm in (("loops", L.S_LOOPS), ("cf", L.S_CF), ("resum", L.S_RESUM), ("ap", L.S_AP))}
    print(f"Nl={Nl}: {dt*1e3:.3f} ms/step -> {B/dt:.0f} evals/s  {st}", flush=True)
    eng.close()
