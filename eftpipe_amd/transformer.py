"""BirdLike records and the copy transformer (same surface as reference eftpipe/transformer.py:10-38)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

from . import _lib as L
from .engine import ROWS

TEMPLATES = ("P11l", "Pctl", "Ploopl", "Pstl")


@dataclass
class PlainBird:
    """(reference transformer.py:14-24)"""

    f: float
    co: Any
    P11l: np.ndarray
    Ploopl: np.ndarray
    Pctl: np.ndarray
    Pstl: np.ndarray
    Picc: np.ndarray
    PctNNLOl: Optional[np.ndarray] = None


class BirdCopier:
    """(reference transformer.py:27-38)"""

    def transform(self, birdlike):
        return PlainBird(f=birdlike.f, co=birdlike.co, P11l=birdlike.P11l.copy(), Ploopl=birdlike.Ploopl.copy(),
                         Pctl=birdlike.Pctl.copy(), Pstl=birdlike.Pstl.copy(), Picc=birdlike.Picc.copy(),
                         PctNNLOl=None if birdlike.PctNNLOl is None else birdlike.PctNNLOl.copy())


def _templates_in_place(eng, birdlike):
    """Make the device template block hold `birdlike`'s four template arrays; -> (nl, nx).  A Bird whose last stage left them on the
    device (eftpipe_amd.pybird.Bird: lazy attributes) needs no transfer; anything else is uploaded, after the results of whichever
    bird still owns the engine have been brought to the host."""
    from .pybird import claim, release

    if hasattr(birdlike, "_on_device"):
        claim(eng, birdlike)
        if birdlike._on_device(eng, "TEMPL"):
            return birdlike._shape
    else:
        release(eng)
    nl, nx = birdlike.P11l.shape[0], birdlike.P11l.shape[-1]
    T = np.empty((nl, 24, nx))
    for n, sl in ROWS.items():
        T[:, sl] = getattr(birdlike, n)
    eng.set_template_dims(nl, nx)
    eng.put("TEMPL", T)
    return nl, nx


def apply_operator_in_place(eng, op_id, bird):
    """A plugin stage that re-binds the bird's templates (Window.Window, FiberCollision.fibcolWindow): run one registered operator on
    the device block.  For a lazy Bird nothing crosses PCIe -- the outputs stay on the device as its pending attributes; any other
    BirdLike gets host arrays back.  The operator's own stochastic matrix decides what happens to Pstl (window_st / fiberst)."""
    nnlo = getattr(bird, "PctNNLOl", None)
    if hasattr(bird, "_templates_pending") and nnlo is None:
        _templates_in_place(eng, bird)
        bird.__dict__["_engine"] = eng
        eng.apply_operator(op_id, 1, sync=False)
        bird._dev.add("TEMPL")
        bird._templates_pending(TEMPLATES, shape=tuple(eng.dims))
        return
    out = apply_operator_to_birdlike(eng, op_id, bird)
    for n in TEMPLATES:
        setattr(bird, n, out[n])
    if "PctNNLOl" in out:
        bird.PctNNLOl = out["PctNNLOl"]


def apply_operator_to_birdlike(eng, op_id, birdlike):
    """Run one registered operator on the templates of a BirdLike and download the result (Binning / Chained: the input object stays
    as it is).  -> dict(P11l, Pctl, Ploopl, Pstl[, PctNNLOl]) with the operator's output shape.  Templates that a lazy Bird still holds
    on the device are not uploaded again: they are fetched for the bird (its later consumers read them) and used where they are."""
    nnlo = getattr(birdlike, "PctNNLOl", None)
    if hasattr(birdlike, "_materialize") and birdlike._on_device(eng, "TEMPL"):
        nl, nx = birdlike._shape
        birdlike._materialize("TEMPL")   # the block itself is still in place; the bird just stops claiming it
    else:
        nl, nx = _templates_in_place(eng, birdlike)
        if hasattr(birdlike, "_dev"):
            birdlike._dev.discard("TEMPL")
    eng.apply_operator(op_id, 1)
    nlo, nxo = eng.dims
    out = eng.get("TEMPL", (nlo, 24, nxo))
    res = {n: np.ascontiguousarray(out[:, sl]) for n, sl in ROWS.items()}
    if nnlo is not None:  # the NNLO counter-terms ride through the same operator in the Pctl slots of a second block
        T = np.zeros((nl, 24, nx))
        T[:, 3:6] = nnlo
        eng.set_template_dims(nl, nx)
        eng.put("TEMPL", T)
        eng.apply_operator(op_id, 1)
        res["PctNNLOl"] = np.ascontiguousarray(eng.get("TEMPL", (nlo, 24, nxo))[:, 3:6])
    return res


class PlkInterpolator:
    """Same surface as reference theory.py:75-106: ``PlkInterpolator(ls, kgrid, Plk)(l, k)`` interpolates k P_l(k) with a
    cubic spline through (0, 0) and the grid and divides by k.  The interpolation is the linear operator
    ``tables.interp_operator``; for a batch resident on the device register it with ``Engine.add_operator`` (as Binning
    does) -- this host class serves the single-spectrum calls of the reference's provider API."""

    def __init__(self, ls, kgrid, Plk):
        self.ls = list(ls)
        self._k = np.array(kgrid, dtype=np.float64)
        self._P = np.array(Plk, dtype=np.float64)

    def __call__(self, l, k):
        from .tables import interp_operator

        ll = [int(l)] if np.ndim(l) == 0 else [int(x) for x in l]
        try:
            idx = [self.ls.index(x) for x in ll]
        except ValueError as ex:
            raise ValueError(f"l={ll} not in {self.ls}") from ex
        out = self._P[idx] @ interp_operator(self._k, k).T
        out = out.reshape((len(idx),) + np.shape(k))
        return out[0] if len(idx) == 1 else out
