#!/usr/bin/env python3
"""Latency of the drop-in path: ONE cosmology per call through the PyBird-compatible classes, driven exactly as reference
eftpipe/theory.py:557-609 drives `eftpipe.pybird.pybird` -- the production shape of the reference (native 50-point k grid, Nl = 3,
IR-resummation, AP with APst, DR16 LRG window, binning onto 18 data bins) and the reduce_Plk that EFTLeaf does afterwards.
`bench.py` reports the numbers as extra keys; `python tools/dropin_probe.py` prints them (GPU box).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BS = [2.1401334, 0.77616816 / np.sqrt(2.0), 0.77003455, 0.77616816 / np.sqrt(2.0), -1.8396613, -1.8918368, -1.4856405]
ES = (0.26033594, 0.0, -0.92895016)
WIN = os.path.join(ROOT, "tests", "golden", "win_NGC_LRG_sQ024.npy")
KOUT = np.arange(0.025, 0.2, 0.01)
Z = 0.7


def build_plugins():
    """what EFTLeafKernel.initialize_with_provider builds once (reference theory.py:423-495)"""
    from eftpipe_amd import pybird, synth
    from eftpipe_amd.binning import Binning
    from eftpipe_amd.window import Window

    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    return dict(co=co, nonlinear=pybird.NonLinear(load=False, save=False, co=co), resum=pybird.Resum(co=co),
                ap=pybird.APeffect(Om_AP=synth.OM_AP, z_AP=Z, co=co, APst=True),
                window=Window(window_configspace_file=WIN, co=co, load=False, save=False), binning=Binning(kout=KOUT, co=co))


def evaluate(pl, cos):
    """one pass of reference theory.py:557-609 + the bias contraction of EFTLeaf.calculate (theory.py:829-874)"""
    from eftpipe_amd import pybird
    from eftpipe_amd.parambasis import reduce_Plk

    bird = pybird.Bird(cos["kin"], cos["Pin"], cos["f"], cos["DA"], cos["H"], Z, co=pl["co"])
    pl["nonlinear"].PsCf(bird)
    bird.setPsCfl()
    pl["resum"].Ps(bird)
    pl["ap"].AP(bird)
    pl["window"].Window(bird)
    binned = pl["binning"].transform(bird)
    return reduce_Plk(binned, BS, es=ES).sum()


def dropin_latency_ms(repeats=20):
    from eftpipe_amd import synth

    pl = build_plugins()
    draws = synth.draw_batch(8, z=Z)
    cosmos = [dict(kin=draws["kin"], Pin=draws["Pin"][i], f=float(draws["f"][i]), DA=float(draws["DA"][i]), H=float(draws["H"][i])) for i in range(8)]
    for c in cosmos[:3]:
        out = evaluate(pl, c)
    assert np.all(np.isfinite(out))
    ts = []
    for r in range(repeats):
        t0 = time.perf_counter()
        evaluate(pl, cosmos[r % 8])
        ts.append(time.perf_counter() - t0)
    ts = np.sort(ts)
    return {"dropin_ms_per_eval": float(np.median(ts) * 1e3), "dropin_ms_per_eval_min": float(ts[0] * 1e3),
            "dropin_config": "one cosmology per call through eftpipe_amd.pybird as reference theory.py:557-609 drives it: native 50-point grid, Nl=3, "
                             "resum + AP(APst) + DR16 LRG window + binning (18 bins) + reduce_Plk; host arrays in, host arrays out"}


def oracle_ms(repeats=3):
    """the CPU port on the same configuration (bounded sample)"""
    from eftpipe_amd import synth
    from oracle import OracleConfig, OracleEngine

    orc = OracleEngine(OracleConfig(Nl=3, ndA=4.5e-5, with_resum=True, with_ap=True, APst=True, Om_AP=synth.OM_AP, z_AP=Z, window_file=WIN, kout=KOUT))
    d = synth.draw_batch(8, z=Z)
    ts = []
    for i in range(repeats):
        t0 = time.perf_counter()
        st = orc.evaluate(d["kin"], d["Pin"][i], float(d["f"][i]), float(d["DA"][i]), float(d["H"][i]))
        st["Picc"] = np.zeros((3, orc.Nk))
        orc.reduce_plk(float(d["f"][i]), orc.binning(st), BS, None, ES)
        ts.append(time.perf_counter() - t0)
    return {"dropin_cpu_port_ms_per_eval": float(np.median(ts) * 1e3)}


if __name__ == "__main__":
    res = dropin_latency_ms()
    res.update(oracle_ms())
    for k, v in res.items():
        print(k, v)
