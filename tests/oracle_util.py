"""Build an OracleEngine matching a golden case (shared by oracle and GPU parity tests)."""
import os

import numpy as np

from eftpipe_amd import synth
from oracle import OracleConfig, OracleEngine

WINDOW_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "win_NGC_LRG_sQ024.npy")

CASE_FLAGS = {
    "caseA": dict(resum=False, ap=False),
    "caseB": dict(resum=False, ap=False),
    "caseC": dict(resum=True, ap=True, APst=True, window=True, binning=True),
    "caseD": dict(resum=True, ap=True),
    "caseE": dict(resum=True, ap=True),
    "caseF": dict(resum=True, ap=True),
    "caseG": dict(resum=True, ap=True, APst=True, window=True),
    "nnlo": dict(resum=True, ap=True, APst=True, window=True, binning=True, with_NNLO=True),
}


def oracle_config(g, name, **over):
    fl = CASE_FLAGS[name]
    native = g["k"].size == 50
    cfg = OracleConfig(
        Nl=int(g["Nl"]), k=None if native else g["k"], kmA=0.7, krA=0.25, ndA=4.5e-5,
        with_resum=fl.get("resum", False), with_ap=fl.get("ap", False), APst=fl.get("APst", False),
        Om_AP=synth.OM_AP, z_AP=float(g["z"]),
        window_file=WINDOW_FIXTURE if fl.get("window") else None,
        kout=g["kout"] if fl.get("binning") else None, with_NNLO=fl.get("with_NNLO", False),
    )
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def oracle_engine(g, name, **over):
    return OracleEngine(oracle_config(g, name, **over))
