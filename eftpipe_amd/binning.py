"""k-binning of the theory templates (same surface as reference eftpipe/binning.py:17-162): the cubic
spline onto the quadrature points, the k^2-weighted trapezoid rule and the bin volume are folded into one
[nbins, Nk] matrix on the host and applied on the device (gemm_rows_kernel)."""
from __future__ import annotations

import numpy as np

from ._log import HasLogger
from .tables import binning_operator
from .transformer import PlainBird, apply_operator_to_birdlike


class Binning(HasLogger):
    def __init__(self, kout, accboost: int = 1, decimals: int = 2, co=None, name: str = "pybird.binning",
                 kstart=None, kend=None, nbins=None) -> None:
        from . import pybird

        self.set_logger(name=name)
        self.kout = np.array(kout)
        self.co = pybird.common if co is None else co
        self.accboost, self.decimals = accboost, decimals
        self.mpi_info("binning correction: on")
        self.matrix, self.keff, self.binmin, self.binmax = binning_operator(
            self.co.k, self.kout, accboost=accboost, decimals=decimals, kstart=kstart, kend=kend, nbins=nbins)
        self.binvol = (self.binmax**3 - self.binmin**3) / 3.0
        self._ops = {}

    def integrBinning(self, P):
        """Bin average of any array [..., Nk] (reference binning.py:131-144): the host form of the operator ``kbinning`` applies on the
        device (cubic spline onto 100 x accboost points per bin, k^2-weighted trapezoid rule, divided by the bin volume)."""
        return np.asarray(P) @ self.matrix.T

    def kbinning(self, bird):
        """(reference binning.py:146-159)"""
        from .pybird import engine_for

        eng = engine_for(bird.co)
        Nl = bird.P11l.shape[0]
        key = (id(eng), Nl)
        if key not in self._ops:
            self._ops[key] = eng.add_operator(np.einsum("al,xk->alxk", np.eye(Nl), self.matrix))
        out = apply_operator_to_birdlike(eng, self._ops[key], bird)
        out.setdefault("PctNNLOl", None)
        return PlainBird(f=bird.f, co=bird.co, Picc=bird.Picc @ self.matrix.T, **out)

    def transform(self, birdlike):
        return self.kbinning(birdlike)
