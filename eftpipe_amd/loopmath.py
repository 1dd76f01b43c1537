"""Closed-form pieces of the one-loop engine evaluated on the host at init time.

Formula data (factored M22b / M13b rational functions, Q polynomials, mu weights, native grids)
lives in ``eftpipe_amd/data`` (extracted by tools/make_tables.py); this module only evaluates it.
Reference counterparts: eftpipe/pybird/pybird.py:89-173 (mu, M13a/b, M22a/b, MPC), :472-482 (grids).
"""
from __future__ import annotations

import json
import os
from functools import lru_cache

import numpy as np
from scipy.special import loggamma

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# mu-power of each term (reference pybird.py:568-582)
MU_POWER_11 = (0, 2, 4)
MU_POWER_CT = (0, 2, 4, 2, 4, 6)
MU_POWER_22 = 6 * (0,) + 7 * (2,) + (4, 2, 4, 2, 4, 2) + 3 * (4,) + (6, 4, 6, 4, 6, 8)
MU_POWER_13 = 2 * (0,) + 4 * (2,) + 3 * (4,) + (6,)

# regrouping of the 28 + 10 loop pieces into 12 bias groups: (group, power of f)
# (reference pybird.py:762-803)
GROUP_22 = {20: (0, 2), 23: (0, 3), 24: (0, 3), 25: (0, 4), 26: (0, 4), 27: (0, 4),
            9: (1, 1), 14: (1, 2), 15: (1, 2), 21: (1, 3), 22: (1, 3),
            10: (2, 1), 16: (2, 2), 17: (2, 2),
            11: (4, 1), 18: (4, 2), 19: (4, 2),
            0: (5, 0), 6: (5, 1), 12: (5, 2), 13: (5, 2),
            1: (6, 0), 7: (6, 1), 2: (8, 0), 8: (8, 1), 3: (9, 0), 4: (10, 0), 5: (11, 0)}
GROUP_13 = {7: (0, 2), 8: (0, 3), 9: (0, 3), 3: (1, 1), 5: (1, 2), 6: (1, 2), 4: (3, 1),
            0: (5, 0), 2: (5, 1), 1: (7, 0)}


@lru_cache(maxsize=None)
def formula_data():
    with open(os.path.join(_DATA, "pt_tables.json")) as fh:
        return json.load(fh)


@lru_cache(maxsize=None)
def q_polynomials(Nl):
    """[2, Nl, Nl, Nn, 15] ascending-power coefficients in f; index 0 = the reference's
    ``Qa[0]`` / ``Qawithhex[0]`` table (pybird.py:460-469, resumfactor.py:4632-4643)."""
    z = np.load(os.path.join(_DATA, "q_tables.npz"))
    return np.ascontiguousarray(z["Qa_Nl2" if Nl == 2 else "Qa_Nl3"])


def native_k():
    return np.array(formula_data()["kbird"])


def native_s():
    return np.array(formula_data()["sbird"])


def _poly(terms, vals):
    tot = 0.0
    for mon, c in terms:
        t = float(c)
        for v, p in zip(vals, mon):
            if p:
                t = t * v**p
        tot = tot + t
    return tot


def _rational(entry, vals):
    r = entry["scale"][0] / entry["scale"][1]
    for terms, m in entry["num"]:
        r = r * _poly(terms, vals) ** m
    for terms, m in entry["den"]:
        r = r / _poly(terms, vals) ** m
    return r


def gamma_ratio_22(n1, n2):
    """Gamma-function factor shared by all 22 matrices (pybird.py:152-156)."""
    lg = loggamma(1.5 - n1) + loggamma(1.5 - n2) + loggamma(-1.5 + n1 + n2)
    lg_den = loggamma(n1) + loggamma(3.0 - n1 - n2) + loggamma(n2)
    return np.exp(lg) / (8.0 * np.pi**1.5 * np.exp(lg_den))


def tan_factor_13(n1):
    """Common factor of the 13 vectors (pybird.py:112-114)."""
    return np.tan(n1 * np.pi) / (14.0 * (n1 - 3.0) * (n1 - 2.0) * (n1 - 1.0) * n1 * np.pi)


def bessel_weight(l, pn):
    """pi^-3/2 2^-2p Gamma(3/2 + l/2 - p) / Gamma(l/2 + p) (pybird.py:159-173)."""
    return np.pi**-1.5 * 2.0 ** (-2.0 * pn) * np.exp(loggamma(1.5 + l / 2.0 - pn) - loggamma(l / 2.0 + pn))


def matrices_22(nu):
    """M22[28, N+1, N+1] complex (pybird.py:1005-1016)."""
    a = gamma_ratio_22(nu[:, None], nu[None, :])
    d = formula_data()["M22b"]
    return np.stack([a * (_rational(d[b], (nu[:, None], nu[None, :])) + 0.0 * a) for b in range(28)])


def vectors_13(nu):
    """M13[10, N+1] complex (pybird.py:1018-1023)."""
    a = tan_factor_13(nu)
    d = formula_data()["M13b"]
    return np.stack([a * (_rational(d[b], (nu,)) + 0.0 * a) for b in range(10)])


def mu_weights(Nl):
    """l11[Nl,3], lct[Nl,6], lctNNLO[Nl,3], l22[Nl,28], l13[Nl,10] (pybird.py:562-582), incl. 48/148 as written."""
    mu = {int(p): v for p, v in formula_data()["mu"].items()}
    pick = lambda plist: np.array([[mu[p][i] for p in plist] for i in range(Nl)], dtype=np.float64)
    return dict(l11=pick(MU_POWER_11), lct=pick(MU_POWER_CT), lctNNLO=pick((4, 6, 8)), l22=pick(MU_POWER_22), l13=pick(MU_POWER_13))
