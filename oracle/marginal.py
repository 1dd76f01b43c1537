"""CPU restatement of the Gaussian (derivative) table and the analytically marginalised log-posterior.

TEST INFRASTRUCTURE: imported only by tests/, smoke() and bench.py's cpu_baseline.  Follows
  * reference eftpipe/parambasis.py:249-316  (WestCoastBasis.derivative_table: dP_l/d(gaussian parameter)), :378-444 (EastCoastBasis)
  * reference eftpipe/likelihood.py:167-195  (flatten: multipoles x masked k bins -> data-vector order)
  * reference eftpipe/marginal.py:79-203     (Marginalizable.marginalized_logp and calc_F0 / calc_F1i / calc_F2ij)
Pinned by tests/test_oracle_golden.py against tests/golden/marg.npz (outputs of the real reference, tools/make_fixtures.py marg).
"""
from __future__ import annotations

import numpy as np

GAUSSIAN = ("b3", "cct", "cr1", "cr2")
STOCHASTIC = ("ce0", "cemono", "cequad")


def gaussian_names(prefix="", cross_prefix=()):
    """parambasis.py:209-223 (without the NNLO names, which never enter the table when with_NNLO is off)"""
    if cross_prefix:
        return [x + p for x in cross_prefix for p in GAUSSIAN] + [prefix + p for p in STOCHASTIC]
    return [prefix + p for p in GAUSSIAN + STOCHASTIC]


def derivative_table(st, f, b1A, b1B=None, kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None, ndB=None):
    """parambasis.py:249-316.  st: dict with Ploopl [No,12,nx], Pctl [No,6,nx], Pstl [No,3,nx]; b1B None = auto spectrum.
    -> list of [No, nx] arrays in the order of gaussian_names()."""
    Ploopl, Pctl, Pstl = st["Ploopl"], st["Pctl"], st["Pstl"]
    cross = b1B is not None
    kmB, krB, ndB = (kmA if kmB is None else kmB), (krA if krB is None else krB), (ndA if ndB is None else ndB)
    out = []
    if cross:
        for b1o, km, kr in ((b1B, kmA, krA), (b1A, kmB, krB)):  # parameters of tracer A see b1 of B and vice versa
            out.append(0.5 * Ploopl[:, 3] + 0.5 * b1o * Ploopl[:, 7])
            out.append(b1o / km**2 * Pctl[:, 0] + f / km**2 * Pctl[:, 3])
            out.append(b1o / kr**2 * Pctl[:, 1] + f / kr**2 * Pctl[:, 4])
            out.append(b1o / kr**2 * Pctl[:, 2] + f / kr**2 * Pctl[:, 5])
    else:
        out.append(Ploopl[:, 3] + b1A * Ploopl[:, 7])
        out.append(2.0 * b1A / kmA**2 * Pctl[:, 0] + 2.0 * f / kmA**2 * Pctl[:, 3])
        out.append(2.0 * b1A / krA**2 * Pctl[:, 1] + 2.0 * f / krA**2 * Pctl[:, 4])
        out.append(2.0 * b1A / krA**2 * Pctl[:, 2] + 2.0 * f / krA**2 * Pctl[:, 5])
    x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
    x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
    out += [Pstl[:, 0] * x1, Pstl[:, 1] * x2, Pstl[:, 2] * x2]
    return out


EAST_GAUSSIAN = ("bGamma3", "c0", "c2", "c4", "Pshot", "a0", "a2")


def eastcoast_bs(f, b1, b2, bG2, bGamma3=0.0, c0=0.0, c2=0.0, c4=0.0, Pshot=0.0, a0=0.0, a2=0.0):
    """EastCoastBasis.reduce_Plk's parameter map (parambasis.py:378-397) -> (bsA, es) for reduce_plk(counterform='eastcoast')"""
    bsA = [b1, b1 + 7 / 2 * bG2, b1 + 15 * bG2 + 6 * bGamma3, 1 / 2 * b2 - 7 / 2 * bG2,
           c0 - f / 3 * c2 + 3 / 35 * f**2 * c4, c2 - 6 / 7 * f * c4, c4]
    return bsA, [Pshot, a0 + 1 / 3 * a2, 2 / 3 * a2]


def eastcoast_derivative_table(st, f, b1, kmA=0.7, ndA=3e-4):
    """EastCoastBasis.reduce_Plk_gaussian_table (parambasis.py:399-444), order EAST_GAUSSIAN; auto spectra only."""
    Ploopl, Pctl, Pstl = st["Ploopl"], st["Pctl"], st["Pstl"]
    x1, x2 = 1.0 / ndA, 1.0 / ndA / kmA**2
    return [
        6.0 * (Ploopl[:, 3] + b1 * Ploopl[:, 7]),
        -2.0 * Pctl[:, 0],
        2 / 3 * f * Pctl[:, 0] - 2.0 * f * Pctl[:, 1],
        -6 / 35 * f**2 * Pctl[:, 0] + 12 / 7 * f**2 * Pctl[:, 1] - 2.0 * f**2 * Pctl[:, 2],
        x1 * Pstl[:, 0],
        x2 * Pstl[:, 1],
        x2 / 3 * (Pstl[:, 1] + 2.0 * Pstl[:, 2]),
    ]


def flatten(ls, array, masks):
    """likelihood.py:167-195: array [No, nx], masks {ell: slice} -> 1-d in data-vector order"""
    return np.hstack([array[ell // 2, masks[ell]] for ell in ls])


def marginalized_logp(PG, PNG, D, invcov, loc, scale, jeffreys=False, return_best=False):
    """marginal.py:79-140.  PG [nG, ndata], PNG/D [ndata], invcov [ndata, ndata], Gaussian prior (loc, scale) per
    marginalised parameter (scale = inf for all: flat)."""
    scale = np.asarray(scale, dtype=float)
    nG = PG.shape[0]
    sigma_inv = np.zeros((nG, nG)) if np.any(np.isinf(scale)) else np.diag(1.0 / scale**2)
    mu = np.asarray(loc, dtype=float)
    res = PNG - D
    F2 = np.einsum("ia,ab,jb->ij", PG, invcov, PG, optimize=True) + sigma_inv
    F1 = -np.einsum("ia,ab,b->i", PG, invcov, res, optimize=True) + sigma_inv @ mu
    F0 = res @ invcov @ res + mu @ sigma_inv @ mu
    sign, logdet = np.linalg.slogdet(F2 / (2 * np.pi))
    if sign <= 0:
        raise RuntimeError("det of F2ij <= 0")
    best = np.linalg.solve(F2, F1)
    chi2 = -F1 @ best + F0 + (0.0 if jeffreys else logdet)
    if not return_best:
        return -0.5 * chi2
    r = best @ PG + PNG - D
    return -0.5 * chi2, float(r @ invcov @ r), best, dict(F2=F2, F1=F1, F0=F0)
