"""CPU restatement of the fibre-collision correction (effective-window method).

TEST INFRASTRUCTURE: imported only by tests/.  Follows reference eftpipe/pybird/pybird.py
  * :44-86     W2D, Hllp, fllp_IR, fllp_UV
  * :1703-1757 FiberCollision.dPcorr  (linear interp1d onto 1024 log-spaced q, masked q-sums per (l, l', k))
  * :1760-1810 FiberCollision.fibcolWindow
Pinned by tests/test_oracle_golden.py against tests/golden/fiber.npz (outputs of the real reference, tools/make_fixtures.py fiber).
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import interp1d
from scipy.special import j1


def w2d(x):
    return (2.0 * j1(x)) / x


def hllp(l, lp, x):
    if l == 2 and lp == 0:
        return x**2 - 1.0
    if l == 4 and lp == 0:
        return 1.75 * x**4 - 2.5 * x**2 + 0.75
    if l == 4 and lp == 2:
        return x**4 - x**2
    return x * 0.0


def f_ir(l, lp, k, q, Dfc):
    if l == lp:
        return (q / k) * w2d(q * Dfc) * (q / k) ** l
    return (q / k) * w2d(q * Dfc) * (2.0 * l + 1.0) / 2.0 * hllp(max(l, lp), min(l, lp), q / k)


def f_uv(l, lp, k, q, Dfc):
    if l == lp:
        return w2d(q * Dfc) * (k / q) ** l
    return w2d(q * Dfc) * (2.0 * l + 1.0) / 2.0 * hllp(max(l, lp), min(l, lp), k / q)


def dpcorr(kout, kPS, PS, Nl, ktrust=0.25, fs=0.6, Dfc=0.43 / 0.6777):
    """PS [Nl, n, len(kPS)] -> dPcorr [Nl, n, len(kout)]"""
    q = np.geomspace(min(kPS), ktrust, num=1024)
    dq = np.concatenate([[0], q[1:] - q[:-1]])
    Pq = interp1d(kPS, PS, axis=-1, bounds_error=False, fill_value="extrapolate")(q)
    out = np.zeros((PS.shape[0], PS.shape[1], len(kout)))
    for l in range(Nl):
        for lp in range(Nl):
            for i, k in enumerate(kout):
                if lp <= l:
                    m = q < k
                    out[l, :, i] += -0.5 * fs * Dfc**2 * np.einsum("q,q,jq,q->j", q[m], dq[m], Pq[lp][:, m], f_ir(2 * l, 2 * lp, k, q[m], Dfc))
                if lp >= l:
                    m = (q > k) & (q < ktrust)
                    out[l, :, i] += -0.5 * fs * Dfc**2 * np.einsum("q,q,jq,q->j", q[m], dq[m], Pq[lp][:, m], f_uv(2 * l, 2 * lp, k, q[m], Dfc))
    return out


def fibcol_window(st, k, Nl, fs, Dfc, ktrust=0.25, fiberst=False):
    """st: dict of template arrays -> new dict with the correlated correction added (Pstl only if fiberst)"""
    out = dict(st)
    for n in ("P11l", "Pctl", "Ploopl") + (("Pstl",) if fiberst else ()):
        out[n] = st[n] + dpcorr(k, k, st[n], Nl, ktrust=ktrust, fs=fs, Dfc=Dfc)
    return out
