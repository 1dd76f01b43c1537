import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from eftpipe_amd import synth, _lib as L
from eftpipe_amd.engine import Engine
from eftpipe_amd.parambasis import bias_row
from eftpipe_amd.tables import EngineConfig
Z, B = 0.7, 128
k = synth.survey_kgrid(512)
eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z))), max_batch=B)
draws = [synth.draw_batch(B, z=Z, seed=100 + i) for i in range(4)]
bias = np.stack([bias_row(float(f), [2.14, 0.55, 0.77, 0.55, -1.84, -1.89, -1.49], None, (0.26, 0.0, -0.93), kmA=0.7, krA=0.25, ndA=4.5e-5) for f in draws[0]["f"]])
mask = eng.full_mask(reduce=True)
dd = draws[0]
eng.stage_inputs(dd["Pin"], dd["f"], dd["DA"], dd["H"], bias=bias); eng.run_staged(mask, B)
ts = [0.0, 0.0, 0.0]
pin = eng.pinned_empty((B, 3, 512)) if os.environ.get("PINNED") else np.empty((B, 3, 512))
n = 60
t00 = time.perf_counter()
for i in range(1, n + 1):
    dd = draws[i % 4]
    t0 = time.perf_counter(); eng.stage_inputs(dd["Pin"], dd["f"], dd["DA"], dd["H"], bias=bias)
    t1 = time.perf_counter(); eng.run_staged(mask, B)
    t2 = time.perf_counter()
    if not os.environ.get("NOFETCH"):
        L.check(eng.lib.eftb_fetch_previous(eng._h, L.B["PLK"], L.dptr(pin), pin.size))
    t3 = time.perf_counter()
    ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2
eng.sync()
tot = (time.perf_counter() - t00) / n
print(f"per step {tot*1e3:.3f} ms: stage {ts[0]/n*1e3:.3f}  run_staged {ts[1]/n*1e3:.3f}  fetch {ts[2]/n*1e3:.3f}")
# resident-input async loop for comparison
eng.load_inputs(dd["Pin"], dd["f"], dd["DA"], dd["H"], bias)
for _ in range(3): eng.run(mask, B, sync=False)
eng.sync(); t0 = time.perf_counter()
for _ in range(n): eng.run(mask, B, sync=False)
t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
print(f"resident loop: {(t2-t0)/n*1e3:.3f} ms/step, host enqueue {(t1-t0)/n*1e3:.3f} ms/step")
