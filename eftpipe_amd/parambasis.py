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
