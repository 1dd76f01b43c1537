"""Chained multipoles Q_l = P_l - A_l P_{l+2} (same surface as reference eftpipe/chained.py:13-68),
applied on the device as a registered linear operator (gemm_rows_kernel)."""
from __future__ import annotations

import numpy as np

from .tables import chained_matrix
from .transformer import PlainBird, apply_operator_to_birdlike


def chain_coeff(l: int) -> float:
    """(reference chained.py:13-29)"""
    return float(-chained_matrix(l // 2 + 2)[l // 2, l // 2 + 1])


class Chained:
    def __init__(self):
        self._ops = {}

    def chained_matrix(self, Nl: int):
        """(reference chained.py:32-54)"""
        if Nl not in (2, 3, 4):
            raise NotImplementedError
        return chained_matrix(Nl)

    def transform(self, birdlike):
        """(reference chained.py:56-68)"""
        from .pybird import engine_for

        eng = engine_for(birdlike.co)
        Nl, nx = birdlike.P11l.shape[0], birdlike.P11l.shape[-1]
        key = (id(eng), Nl, nx)
        if key not in self._ops:
            self._ops[key] = eng.add_operator(np.einsum("al,xk->alxk", self.chained_matrix(Nl), np.eye(nx)))
        out = apply_operator_to_birdlike(eng, self._ops[key], birdlike)
        mat = self.chained_matrix(Nl)
        out.setdefault("PctNNLOl", None)
        return PlainBird(f=birdlike.f, co=birdlike.co, Picc=np.einsum("al,l...->a...", mat, birdlike.Picc), **out)
