#!/usr/bin/env python3
"""What one Engine.step call costs the sampler's thread while nothing is fetched (GPU box): the pace at which a burst of steps reaches the queue."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from eftpipe_amd import synth
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig
B=128
cfg = EngineConfig(Nl=3, k=synth.survey_kgrid(512), with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, 0.7)), H_AP=float(synth.hubble(synth.OM_AP, 0.7)))
eng = Engine(cfg, max_batch=B, coalesce=4)
eng.set_latency_mode(False); eng.set_plk_direct(True)
sets=[]
for i in range(8):
    d = synth.draw_batch(B, z=0.7, seed=100+i)
    d["bias"] = np.stack([bias_row(float(f), bench.BS, None, bench.ES, kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
    sets.append(d)
mask = eng.full_mask(reduce=True)
out = eng.pinned_empty((64, B, 3, 512))
for rep in range(4):
    ts=[]
    for i in range(13):
        d=sets[i%8]
        t=time.perf_counter()
        eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=-1, shape=(B,3,512), out=out[i])
        ts.append((time.perf_counter()-t)*1e6)
    eng.flush(); eng.sync()
    print("step call us:", " ".join(f"{x:.1f}" for x in ts))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(13):
    d=sets[i%8]; eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=-1, shape=(B,3,512), out=out[i])
pr.disable(); eng.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
