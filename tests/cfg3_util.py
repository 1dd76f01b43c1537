"""BASELINE cfg 3 as the reference ships it (tests/golden/cfg3.npz, tools/make_fixtures.py cfg3): LRG + chained ELG + their cross
spectrum per likelihood point, DR16 windows at accboost 4 / windowk 0.1, the reference's own data vector and covariance.
Shared by the CPU (oracle / host tables) and the GPU parity tests."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACERS = ("LRG_NGC", "ELG_NGC", "X_NGC")
NAMES = ("P11l", "Pctl", "Ploopl", "Pstl")
CHAINED = {"LRG_NGC": False, "ELG_NGC": True, "X_NGC": False}
CROSS = {"X_NGC": ("LRG_NGC", "ELG_NGC")}


def window_table(t):
    """s, Q0, Q2, Q4 of the tracer's DR16 window (columns of the reference's data/DR16_noric/win_NGC_{LRG,ELG,X}.txt)"""
    return np.load(os.path.join(GOLD, "win_NGC_%s_sQ024.npy" % t.split("_")[0]))


def params(g):
    """non-Gaussian parameter values by name, b2 / b4 derived from c2 / c4 as in the yaml"""
    p = dict(zip((str(n) for n in g["ng_names"]), (float(v) for v in g["ng_values"])))
    for t in ("LRG_NGC_", "ELG_NGC_"):
        p[t + "b2"] = (p[t + "c2"] + p[t + "c4"]) / np.sqrt(2.0)
        p[t + "b4"] = (p[t + "c2"] - p[t + "c4"]) / np.sqrt(2.0)
    return p


def bases():
    from eftpipe_amd.parambasis import WestCoastBasis

    return [WestCoastBasis(prefix=t + "_", cross_prefix=[x + "_" for x in CROSS[t]] if t in CROSS else []) for t in TRACERS]


def scales(g):
    out = []
    for t in TRACERS:
        kmA, krA, ndA, kmB, krB, ndB = (float(x) for x in g[t + "_co"])
        out.append(dict(kmA=kmA, krA=krA, ndA=ndA, kmB=kmB, krB=krB, ndB=ndB))
    return out


def masks(g, t):
    return {int(l): slice(int(a), int(b)) for l, (a, b) in zip(g[t + "_ls"], g[t + "_mask"])}


def final_templates(g, t):
    """what the likelihood sees of tracer t: binned, and chained for ELG"""
    tag = "chained_" if CHAINED[t] else "binned_"
    return {n: g[f"{t}_{tag}{n}"] for n in NAMES}
