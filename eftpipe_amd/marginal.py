"""Analytically marginalised log-posterior on the device.

Host mirror of what ``EFTLike`` + ``Marginalizable`` do per likelihood call in the reference
(eftpipe/likelihood.py:483-549 ``PNG``/``PG`` -> eftpipe/marginal.py:79-140 ``marginalized_logp``), for a whole batch
of walkers whose templates are already resident on the GPU: only ``B`` log-posteriors (and, on request, the best-fit
Gaussian parameters) cross PCIe instead of 0.3 MB of templates per walker.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .parambasis import gaussian_params, gaussian_rows

MAXG = 24


def data_index(ls, masks, nx, tracer=0, nl=None):
    """Flat indices l * nx + x of the data vector in the order of reference likelihood.py:167-195 (``flatten``):
    multipoles ``ls`` (even), each restricted to ``masks[ell]`` (a slice, or None for all bins).  With several tracers per
    likelihood point (``Engine.set_tracers``) the block of tracer t starts at multipole t * nl: pass ``tracer`` and ``nl`` (the
    multipoles per entry of the template block) and concatenate the tracers' indices in data-vector order."""
    out = []
    for ell in ls:
        sl = masks[ell] if masks and masks.get(ell) is not None else slice(0, nx)
        out.append((tracer * (nl or 0) + ell // 2) * nx + np.arange(nx)[sl])
    return np.concatenate(out).astype(np.int32)


def joint_gaussian_rows(bases, fs, params, names, scales):
    """Coefficient rows of a joint likelihood over several tracers (the device form of ``EFTLike.PNG`` / ``EFTLike.PG``, reference
    likelihood.py:430-530): tracer t contributes to marginalised parameter ``names[i]`` iff its basis lists that parameter.
    bases   one ``WestCoastBasis`` / ``EastCoastBasis`` per tracer (a cross spectrum's basis carries ``cross_prefix``)
    fs      growth rate per tracer entry; params: the non-Gaussian parameter values by name; scales: per tracer dict(kmA=, krA=, ndA=[, kmB=, ...])
    -> rows [ntr, len(names) + 1, 24] for one walker (row 0: the model at zero Gaussian parameters)."""
    out = np.zeros((len(bases), len(names) + 1, 24))
    for t, (basis, f, sc) in enumerate(zip(bases, fs, scales)):
        r = basis.gaussian_rows(float(f), params, **sc)
        own = gaussian_params(basis.prefix, tuple(basis.cross_prefix)) if basis.get_name() == "westcoast" else basis.gaussian_params()
        out[t, 0] = r[0]
        for i, n in enumerate(names):
            if n in own:
                out[t, 1 + i] = r[1 + own.index(n)]
    return out


class MarginalLikelihood:
    """Gaussian likelihood of one data vector with the linear bias parameters marginalised analytically.

    engine        an ``Engine`` whose template block has the shape the data were measured on (set the pipeline operator
                  -- window / binning / chained -- before constructing this object)
    index         ``data_index(...)`` or any int array of l * nx + x
    data, invcov  data vector [ndata] and inverse covariance [ndata, ndata]
    loc, scale    Gaussian prior of the marginalised parameters (scale = inf for all of them: flat prior)
    """

    def __init__(self, engine, index, data, invcov, loc, scale, jeffreys=False):
        self.eng = engine
        self.index = np.ascontiguousarray(index, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.float64)
        invcov = np.ascontiguousarray(invcov, dtype=np.float64)
        loc = np.ascontiguousarray(loc, dtype=np.float64)
        scale = np.asarray(scale, dtype=np.float64)
        self.nG = loc.size
        if self.nG > MAXG:
            raise ValueError(f"at most {MAXG} marginalised parameters")
        if np.any(np.isinf(scale)) and not np.all(np.isinf(scale)):
            raise ValueError("only support setting infinite scale for all parameters")  # reference marginal.py:222-226
        sinv = np.ascontiguousarray(np.zeros(self.nG) if np.all(np.isinf(scale)) else 1.0 / scale**2)
        if invcov.shape != (data.size, data.size) or self.index.size != data.size:
            raise ValueError("index, data and invcov disagree on the data-vector length")
        L.check(engine.lib.eftb_set_likelihood(engine._h, data.size, self.index.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(data),
                                               L.dptr(invcov), self.nG, L.dptr(loc), L.dptr(sinv)))
        L.check(engine.lib.eftb_set_option(engine._h, 1, int(bool(jeffreys))))

    def logp(self, rows, return_best=False, rows_nnlo=None):
        """rows [B, nG + 1, 24] (``parambasis.gaussian_rows`` per walker) -> ln P_marg [B]
        (+ full chi2 [B] and best-fit Gaussian parameters [B, nG]).  Raises like the reference when det F2 <= 0.
        With ``Engine.set_tracers(ntr)``: B = walkers * ntr entries (each tracer its own rows, zero rows for parameters that do
        not act on it) -> results per walker [B / ntr]."""
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        B = rows.shape[0]
        if rows.shape[1:] != (self.nG + 1, 24):
            raise ValueError(f"rows must be [B, {self.nG + 1}, 24]")
        buf = np.zeros((B, MAXG + 1, 24))
        buf[:, : self.nG + 1] = rows
        self.eng.put("GROWS", buf)
        if self.eng.cfg.with_NNLO:  # coefficients of PctNNLOl per row (zeros unless given): [B, nG + 1, 3]
            bn = np.zeros((B, MAXG + 1, 3))
            if rows_nnlo is not None:
                bn[:, : self.nG + 1] = np.asarray(rows_nnlo, dtype=np.float64).reshape(B, self.nG + 1, 3)
            self.eng.put("GROWSN", bn)
        elif rows_nnlo is not None:
            raise ValueError("rows_nnlo needs an engine built with with_NNLO")
        self.eng.run(L.S_LOGP, B)
        out = self.eng.get("LOGP", (B // self.eng.ntracers, 2 + MAXG))
        if np.any(np.isnan(out[:, 0])):
            raise RuntimeError("det of F2ij <= 0")
        if return_best:
            return out[:, 0], out[:, 1], out[:, 2 : 2 + self.nG]
        return out[:, 0]


    def eval_logp(self, Pin, f, DA, H, rows, return_best=False):
        """Theory + likelihood in one call (``eftb_eval_logp_batch``): Pin [B, Nkin], f/DA/H [B], rows [B, nG + 1, 24] ->
        ln P_marg [B]; only the inputs and B floats cross PCIe.  The engine's pipeline operator must bring the templates to
        the shape the data index refers to."""
        B, Pin, f, DA, H = self.eng._inputs(Pin, f, DA, H)
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.shape != (B, self.nG + 1, 24):
            raise ValueError(f"rows must be [{B}, {self.nG + 1}, 24]")
        nw = B // self.eng.ntracers
        logp, full, best = np.empty(nw), np.empty(nw), np.empty((nw, self.nG))
        L.check(self.eng.lib.eftb_eval_logp_batch(self.eng._h, B, L.dptr(Pin), L.dptr(f), L.dptr(DA), L.dptr(H), L.dptr(rows),
                                                  L.dptr(logp), L.dptr(full), L.dptr(best)))
        if np.any(np.isnan(logp)):
            raise RuntimeError("det of F2ij <= 0")
        return (logp, full, best) if return_best else logp


__all__ = ["MarginalLikelihood", "data_index", "gaussian_params", "gaussian_rows", "joint_gaussian_rows"]
