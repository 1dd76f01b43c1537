"""Bias contraction onto P_l(k): host mirror of reference eftpipe/parambasis.py:42-136 (west coast).

``bias_vectors`` builds the 3 + 6 + 12 + 3 coefficient vectors; the
contraction itself runs on the device (``reduce_kernel``) when driven through ``Engine`` and is a
24-term dot product per (l, k).  ``reduce_Plk`` keeps the reference signature for BirdLike objects.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class BirdComponent:
    """Same record as reference parambasis.py:30-39."""

    Plin: np.ndarray
    Ploop: np.ndarray
    Pct: np.ndarray
    Pst: np.ndarray
    Picc: np.ndarray

    def sum(self):
        return self.Plin + self.Ploop + self.Pct + self.Pst + self.Picc


def bias_vectors(f, bsA, bsB=None, es=(0.0, 0.0, 0.0), kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None, ndB=None):
    """-> b11[3], bct[6], bloop[12], bst[3] (reference parambasis.py:69-126, counterform='westcoast')."""
    kmB = kmA if kmB is None else kmB
    krB = krA if krB is None else krB
    ndB = ndA if ndB is None else ndB
    b1A, b2A, b3A, b4A, cctA, cr1A, cr2A = bsA
    b1B, b2B, b3B, b4B, cctB, cr1B, cr2B = bsB if bsB is not None else bsA
    ce0, cemono, cequad = es
    b11 = np.array([b1A * b1B, (b1A + b1B) * f, f * f])
    bct = np.array([
        b1A * cctB / kmB**2 + b1B * cctA / kmA**2,
        b1B * cr1A / krA**2 + b1A * cr1B / krB**2,
        b1B * cr2A / krA**2 + b1A * cr2B / krB**2,
        (cctA / kmA**2 + cctB / kmB**2) * f,
        (cr1A / krA**2 + cr1B / krB**2) * f,
        (cr2A / krA**2 + cr2B / krB**2) * f,
    ])
    bloop = np.array([
        1.0, 0.5 * (b1A + b1B), 0.5 * (b2A + b2B), 0.5 * (b3A + b3B), 0.5 * (b4A + b4B), b1A * b1B,
        0.5 * (b1A * b2B + b1B * b2A), 0.5 * (b1A * b3B + b1B * b3A), 0.5 * (b1A * b4B + b1B * b4A),
        b2A * b2B, 0.5 * (b2A * b4B + b2B * b4A), b4A * b4B,
    ])
    x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
    x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
    bst = np.array([ce0 * x1, cemono * x2, cequad * x2])
    return b11, bct, bloop, bst


def bias_row(f, bsA, bsB=None, es=(0.0, 0.0, 0.0), **scales):
    """The 24 coefficients in template-row order (P11l, Pctl, Ploopl, Pstl) for the device reduce."""
    return np.concatenate(bias_vectors(f, bsA, bsB, es, **scales))


def reduce_Plk(bird, bsA, bsB=None, es=(0.0, 0.0, 0.0)):
    """BirdLike -> BirdComponent (reference parambasis.py:42-136; NNLO and east-coast not on the path)."""
    co = bird.co
    b11, bct, bloop, bst = bias_vectors(bird.f, list(bsA), None if bsB is None else list(bsB), tuple(es),
                                        kmA=co.kmA, krA=co.krA, ndA=co.ndA, kmB=co.kmB, krB=co.krB, ndB=co.ndB)
    No = co.No
    return BirdComponent(
        Plin=np.einsum("b,lbx->lx", b11, bird.P11l[:No]),
        Ploop=np.einsum("b,lbx->lx", bloop, bird.Ploopl[:No]),
        Pct=np.einsum("b,lbx->lx", bct, bird.Pctl[:No]),
        Pst=np.einsum("b,lbx->lx", bst, bird.Pstl[:No]),
        Picc=bird.Picc[:No],
    )


# ----------------------------------------------------------------------------- Gaussian table
GAUSSIAN = ("b3", "cct", "cr1", "cr2")
STOCHASTIC = ("ce0", "cemono", "cequad")


def gaussian_params(prefix="", cross_prefix=()):
    """Names of the analytically marginalisable parameters, in table order (reference parambasis.py:209-223)."""
    if cross_prefix:
        return [x + p for x in cross_prefix for p in GAUSSIAN] + [prefix + p for p in STOCHASTIC]
    return [prefix + p for p in GAUSSIAN + STOCHASTIC]


def gaussian_rows(f, ngA, ngB=None, kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None, ndB=None):
    """Coefficient rows over the 24 template rows (P11l[3], Pctl[6], Ploopl[12], Pstl[3]) of

        row 0        P_NG: the model with every Gaussian parameter at 0 (reference likelihood.py:524-549 via reduce_Plk)
        rows 1..nG   dP/d(gaussian parameter) in the order of ``gaussian_params`` (reference parambasis.py:249-316)

    for the non-Gaussian parameters ngA = (b1, b2, b4) [and ngB for a cross spectrum].  Everything downstream of the
    templates is linear in these rows, so the device builds P_NG and P_G with one small contraction per walker."""
    cross = ngB is not None
    kmB = kmA if kmB is None else kmB
    krB = krA if krB is None else krB
    ndB = ndA if ndB is None else ndB
    b1A, b2A, b4A = ngA
    b1B, b2B, b4B = ngB if cross else ngA
    bsA = [b1A, b2A, 0.0, b4A, 0.0, 0.0, 0.0]
    bsB = [b1B, b2B, 0.0, b4B, 0.0, 0.0, 0.0] if cross else None
    rows = [bias_row(f, bsA, bsB, (0.0, 0.0, 0.0), kmA=kmA, krA=krA, ndA=ndA, kmB=kmB, krB=krB, ndB=ndB)]
    CT, LOOP, ST = 3, 9, 21  # first template row of Pctl, Ploopl, Pstl

    def row(pairs):
        r = np.zeros(24)
        for i, c in pairs:
            r[i] = c
        return r

    if cross:
        for b1o, km, kr in ((b1B, kmA, krA), (b1A, kmB, krB)):
            rows.append(row([(LOOP + 3, 0.5), (LOOP + 7, 0.5 * b1o)]))
            rows.append(row([(CT + 0, b1o / km**2), (CT + 3, f / km**2)]))
            rows.append(row([(CT + 1, b1o / kr**2), (CT + 4, f / kr**2)]))
            rows.append(row([(CT + 2, b1o / kr**2), (CT + 5, f / kr**2)]))
    else:
        rows.append(row([(LOOP + 3, 1.0), (LOOP + 7, b1A)]))
        rows.append(row([(CT + 0, 2.0 * b1A / kmA**2), (CT + 3, 2.0 * f / kmA**2)]))
        rows.append(row([(CT + 1, 2.0 * b1A / krA**2), (CT + 4, 2.0 * f / krA**2)]))
        rows.append(row([(CT + 2, 2.0 * b1A / krA**2), (CT + 5, 2.0 * f / krA**2)]))
    x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
    x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
    rows += [row([(ST + 0, x1)]), row([(ST + 1, x2)]), row([(ST + 2, x2)])]
    return np.stack(rows)
