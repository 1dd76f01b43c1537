"""Survey-window convolution (same surface as reference eftpipe/window.py:40-415).

Init: W_{al}(k, p) from the configuration-space window on the device (window_matrix_device: k-independent tables from
tables.window_tables on the host, Bessel factor + FP64-MFMA products + mask / dp / spline fold in eftb_window_precompute),
or from an existing ``*.npy`` cache written by the reference (same array layout [Na, Nl, Nk, Np]); the band mask, the
dp weights and the cubic spline k -> p are folded into one dense [Na, Nl, Nk, Nk] operator.
Per evaluation (device): one FP64-MFMA GEMM over the template block (gemm_rows_kernel)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import NamedTuple

import numpy as np

from ._log import HasLogger
from .tables import spline_matrix, window_fold, window_pgrid, window_tables
from .transformer import apply_operator_in_place, apply_operator_to_birdlike


class MetaInfoError(Exception):
    pass


def window_matrix_device(k, sw, Qq, Na, Nl, withmask=True, windowk=0.05, device=0, timing=None, **kw):
    """Wal [Na, Nl, Nk, Np], p, Waldk, Wfold [Na, Nl, Nk, Nk] through eftb_window_precompute (reference window.py:262-359);
    kw as tables.window_tables.  ``timing`` (a dict) receives the host-table and device-kernel times."""
    import ctypes as C
    import time

    from . import _lib as L

    lib = L.load()
    k = np.ascontiguousarray(k, dtype=np.float64)
    t0 = time.perf_counter()
    x, Qt, T, p = window_tables(k, sw, Qq, Na, Nl, **kw)
    S = np.ascontiguousarray(spline_matrix(k, p))
    t1 = time.perf_counter()
    Wal = np.empty((Na, Nl, k.size, p.size))
    Waldk = np.empty_like(Wal)
    Wfold = np.empty((Na, Nl, k.size, k.size))
    ms = C.c_double()
    T = np.ascontiguousarray(T)
    p = np.ascontiguousarray(p, dtype=np.float64)
    L.check(lib.eftb_window_precompute(device, Na, Nl, k.size, x.size, p.size, L.dptr(k), L.dptr(x), L.dptr(Qt), L.dptr(T), L.dptr(p),
                                       int(bool(withmask)), float(windowk), L.dptr(S), L.dptr(Wal), L.dptr(Waldk), L.dptr(Wfold),
                                       C.cast(C.byref(ms), C.POINTER(C.c_double))))
    if timing is not None:
        timing.update(host_tables_s=t1 - t0, device_kernels_ms=ms.value, call_s=time.perf_counter() - t1)
    return Wal, p, Waldk, Wfold


class Window(HasLogger):
    def __init__(self, window_fourier_file=None, window_configspace_file=None, co=None, load=True, save=True,
                 check_meta=True, Na=None, Nl=None, Nq=3, pmax=None, accboost=1, withmask=True, windowk=0.05,
                 Nmax=4096, xmin_factor=1.0, xmax_factor=100.0, bias=-1.6, window_param=1, window_st=True, icc=None,
                 name="pybird.window", snapshot=False):
        from . import pybird

        self.set_logger(name=name)
        self.co = pybird.common if co is None else co
        if window_fourier_file is None and window_configspace_file is None:
            raise ValueError("Window requires window_fourier_file or window_configspace_file or both")
        self.icc = icc  # an eftpipe_amd.icc.IntegralConstraint: its matrix is folded into the operator (reference window.py:393-406)
        self.window_fourier_file = Path(window_fourier_file).resolve() if window_fourier_file else None
        self.window_configspace_file = Path(window_configspace_file).resolve() if window_configspace_file else None
        self.window_st, self.withmask, self.windowk = window_st, withmask, windowk
        Na = Na if Na else self.co.Nl
        Nl = Nl if Nl else self.co.Nl
        if Na > self.co.Nl or Nl > self.co.Nl:
            raise ValueError(f"request Na={Na}, Nl={Nl} while bird only compute Nl up to {self.co.Nl}")
        if Na > Nl:
            raise ValueError(f"dangerous settings Na={Na}, Nl={Nl}")
        if pmax is None:
            pmax = float(self.co.k.max())
        self.p = window_pgrid(pmax, accboost)
        cfile = str(self.window_configspace_file) if self.window_configspace_file else None
        self.meta = dict(Na=Na, Nl=Nl, Nq=Nq, pmax=pmax, accboost=accboost, Nmax=Nmax, xmin_factor=xmin_factor,
                         xmax_factor=xmax_factor, bias=bias, window_param=window_param, window_configspace_file=cfile,
                         k=self.co.k.tolist())
        self.Wal = self._load_Wal(load, check_meta)
        computed = self.Wal is None
        if computed:
            self.Wal, self.Waldk, self.Wfold = self._compute_Wal()
        else:  # cached matrix (possibly written by the reference): only the fold is left to do
            self.Wfold, self.Waldk = window_fold(self.co.k, self.Wal, self.p, windowk=windowk, withmask=withmask)
        if save and computed and self.window_fourier_file is not None:
            self._save_Wal()
        self.snapshot = snapshot
        self._op = None

    # ---- cache files in the reference's format (window.py:204-260, 361-369)
    def _load_Wal(self, load, check_meta):
        f = self.window_fourier_file
        if not load or f is None or not f.exists():
            return None
        Wal = np.load(f)
        if Wal.shape != (self.meta["Na"], self.meta["Nl"], self.co.Nk, self.p.size):
            self.mpi_warning("cached window %s has shape %s, recomputing", f, Wal.shape)
            return None
        meta_file = f.with_suffix(".json")
        if check_meta and meta_file.exists():
            with meta_file.open() as fh:
                meta = json.load(fh)
            if self.meta["window_configspace_file"] is None:
                self.meta["window_configspace_file"] = meta.get("window_configspace_file")
            if meta != self.meta:
                raise MetaInfoError(f"inconsistent meta info\nloaded matrix's meta:\n{meta}\nexpect:\n{self.meta}")
        return Wal

    def _save_Wal(self):
        np.save(self.window_fourier_file, self.Wal)
        with self.window_fourier_file.with_suffix(".json").open("w") as fh:
            json.dump(self.meta, fh, indent=2)

    def _compute_Wal(self):
        if self.window_configspace_file is None:
            raise ValueError("please specify a configuration space mask file")
        f = self.window_configspace_file
        tab = np.load(f) if f.suffix == ".npy" else np.loadtxt(f)
        while tab[0, 0] == 0.0:
            tab = tab[1:]
        m = self.meta
        tab = tab[:, : 1 + m["Nq"]]
        Wal, p, Waldk, Wfold = window_matrix_device(self.co.k, tab[:, 0], tab[:, 1:].T, m["Na"], m["Nl"], withmask=self.withmask,
                                                    windowk=self.windowk, device=0, accboost=m["accboost"],
                                                    Nmax=m["Nmax"], xmin_factor=m["xmin_factor"], xmax_factor=m["xmax_factor"],
                                                    bias=m["bias"], window_param=m["window_param"], pmax=m["pmax"])
        assert np.array_equal(p, self.p)
        return Wal, Waldk, Wfold

    def Window(self, bird):
        """Convolve P11l, Pctl, Ploopl (and Pstl if window_st) in place (reference window.py:389-415)."""
        from .pybird import engine_for

        eng = engine_for(bird.co)
        if self.Wfold.shape[1] != bird.co.Nl:  # Window(Nl != co.Nl): the reference's einsum fails the same way (window.py:387)
            raise ValueError(f"operands could not be broadcast together: window matrix has {self.Wfold.shape[1]} input multipoles, the bird {bird.co.Nl}")
        if self._op is None or self._op[0] is not eng:
            Na, Nl, Nk = self.Wfold.shape[0], self.Wfold.shape[1], self.Wfold.shape[2]
            op = self.Wfold
            if self.icc is not None:  # P -> W P - W_ic P as one matrix
                if self.icc.Wfold.shape != op.shape:
                    raise ValueError(f"icc matrix {self.icc.Wfold.shape} does not match the window matrix {op.shape}")
                op = op - self.icc.Wfold
            keep = None
            if not self.window_st and Na == Nl:  # Pstl passes through untouched: an identity matrix for the stochastic rows
                keep = np.einsum("al,xk->alxk", np.eye(Nl), np.eye(Nk))
            self._op = (eng, eng.add_operator(op, stochastic=keep), keep is not None or self.window_st)
        if self._op[2]:
            apply_operator_in_place(eng, self._op[1], bird)
        else:  # window_st off with Na != Nl: the reference keeps the old Pstl array as it is
            keep = bird.Pstl
            apply_operator_in_place(eng, self._op[1], bird)
            bird.Pstl = keep
        if self.icc is not None:
            bird.Picc = bird.Picc - self.icc.PSN
        if self.snapshot:
            bird.create_snapshot("window")

    def integrWindow(self, P, interp=True):
        """Host form of the convolution of one array of rows [Nl, n, Nk] (reference window.py:371-387, same signature): cubic interpolation
        onto the p grid unless ``interp=False`` (P already sampled on p), then the masked dp-weighted matrix.  ``Window`` applies the same
        operator on the device; kept for callers of the helper.  A 2-D [Nl, Nk] input is treated as one row per multipole."""
        from scipy.interpolate import interp1d

        P = np.asarray(P)
        flat = P.ndim == 2
        if flat:
            P = P[:, None]
        Pp = interp1d(self.co.k, P, axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")(self.p) if interp else P
        out = np.einsum("alkp,lsp->ask", self.Waldk, Pp, optimize=True)
        return out[:, 0] if flat else out


# ----------------------------------------------------------------------------- window as a ready-made matrix
class PInfo(NamedTuple):
    """Layout of one axis of a stacked window-matrix file (reference window.py:418-423)."""

    ells: tuple
    kmin: float
    kmax: float
    nbins: int


class PolesInfo(NamedTuple):
    """(reference window.py:470-474)"""

    nells: int
    kstart: float
    kend: float
    nbin: int


def _axis_mask(info, ells_keep, lo, hi):
    """rows / columns of the stacked [len(ells) * nbins] axis that belong to the kept multipoles and to bin-centre range [lo, hi)"""
    edges = np.linspace(info.kmin, info.kmax, info.nbins + 1)
    centres = 0.5 * (edges[1:] + edges[:-1])
    a, b = int(np.searchsorted(centres, lo)), int(np.searchsorted(centres, hi))
    mask = np.zeros(info.nbins * len(info.ells), dtype=bool)
    for n, ell in enumerate(info.ells):
        if ell in ells_keep:
            mask[n * info.nbins + a : n * info.nbins + b] = True
    return mask


def to_window_matrix(matrix, inpoles, outpoles, ells_in, kmax_in, ells_out, kmin_out, kmax_out):
    """Stacked window matrix [out rows, in columns] -> [len(ells_out), len(ells_in), nk_out, nk_in] restricted to the requested
    multipoles and k ranges (reference window.py:426-467; like the reference, the multipoles keep the file's order)."""
    m_in = _axis_mask(inpoles, ells_in, -np.inf, kmax_in)
    m_out = _axis_mask(outpoles, ells_out, kmin_out, kmax_out)
    sub = np.asarray(matrix)[np.ix_(m_out, m_in)]
    no, ni = len(ells_out), len(ells_in)
    return np.ascontiguousarray(sub.reshape(no, sub.shape[0] // no, ni, sub.shape[1] // ni).transpose(0, 2, 1, 3))


class WindowMatrix(HasLogger):
    """Window convolution with a ready-made matrix (same surface as reference window.py:479-586): cubic interpolation of
    the templates onto ``kavg`` folded with ``matrix`` into one dense device operator."""

    def __init__(self, matrix, inpoles, outpoles, co=None, window_st=False, icc=None, name="pybird.WindowMatrix", snapshot=False):
        from . import pybird
        from .tables import spline_matrix

        self.set_logger(name=name)
        self.matrix, self.inpoles, self.outpoles = np.asarray(matrix, dtype=np.float64), inpoles, outpoles
        self.co = pybird.common if co is None else co
        self.window_st, self.snapshot = window_st, snapshot
        if icc:
            raise NotImplementedError("ICC not implemented for WindowMatrix")
        if self.matrix.shape != (outpoles.nells, inpoles.nells, outpoles.nbin, inpoles.nbin):
            raise ValueError("matrix shape does not match meta information")
        if inpoles.nells != self.co.Nl:
            raise ValueError("input poles do not match self.co.Nl")
        if self.matrix.shape[3] != self.kavg.size:
            raise ValueError("matrix columns do not match kavg")
        self.operator = np.einsum("alxp,pk->alxk", self.matrix, spline_matrix(self.co.k, self.kavg))
        self._op = None

    @classmethod
    def load(cls, path, ells, kmin, kmax, co=None, window_st=False, icc=None, name="pybird.WindowMatrix", snapshot=False):
        """(reference window.py:511-545, with its hard-coded file layout)"""
        from . import pybird

        co = pybird.common if co is None else co
        m = to_window_matrix(np.loadtxt(path), PInfo((0, 2, 4), 0, 0.4, 400), PInfo((0, 1, 2, 3, 4), 0, 0.4, 40),
                             ells_in=tuple(2 * i for i in range(co.Nl)), kmax_in=co.k.max(), ells_out=tuple(ells), kmin_out=kmin, kmax_out=kmax)
        return cls(m, PolesInfo(co.Nl, 0, co.k.max(), m.shape[3]), PolesInfo(len(ells), kmin, kmax, m.shape[2]), co=co, window_st=window_st,
                   icc=icc, name=name, snapshot=snapshot)

    @property
    def kavg(self):
        return np.linspace(0, 0.4, 400)[:300]  # as hard-coded in the reference (window.py:548-550)

    def convolve(self, Plk):
        """host form (reference window.py:557-564)"""
        return np.einsum("alxk,l...k->a...x", self.operator, np.asarray(Plk))

    def Window(self, bird):
        """(reference window.py:566-586)"""
        from .pybird import engine_for

        eng = engine_for(bird.co)
        if self._op is None or self._op[0] is not eng:
            self._op = (eng, eng.add_operator(self.operator))
        keep = bird.Pstl
        apply_operator_in_place(eng, self._op[1], bird)
        if not self.window_st:
            bird.Pstl = keep
        bird.Picc = np.zeros((self.outpoles.nells, self.outpoles.nbin))
        if self.snapshot:
            bird.create_snapshot("window")
