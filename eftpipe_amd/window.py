"""Survey-window convolution (same surface as reference eftpipe/window.py:40-415).

Init (host): W_{al}(k, p) from the configuration-space window by FFTLog (tables.window_matrix), or from an
existing ``*.npy`` cache written by the reference (same array layout [Na, Nl, Nk, Np]); the band mask, the
dp weights and the cubic spline k -> p are folded into one dense [Na, Nl, Nk, Nk] operator.
Per evaluation (device): one FP64-MFMA GEMM over the template block (gemm_rows_kernel)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from ._log import HasLogger
from .tables import window_fold, window_matrix, window_pgrid
from .transformer import apply_operator_to_birdlike


class MetaInfoError(Exception):
    pass


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
        if icc is not None:
            raise NotImplementedError("integral-constraint correction is outside the accelerated hot path")
        self.window_fourier_file = Path(window_fourier_file).resolve() if window_fourier_file else None
        self.window_configspace_file = Path(window_configspace_file).resolve() if window_configspace_file else None
        self.window_st, self.withmask, self.windowk = window_st, withmask, windowk
        Na = Na if Na else self.co.Nl
        Nl = Nl if Nl else self.co.Nl
        if Na > self.co.Nl or Nl > self.co.Nl:
            raise ValueError(f"request Na={Na}, Nl={Nl} while bird only compute Nl up to {self.co.Nl}")
        if Na > Nl:
            raise ValueError(f"dangerous settings Na={Na}, Nl={Nl}")
        if Nl != self.co.Nl:
            raise NotImplementedError("Window(Nl != co.Nl) is not supported by the device operator")
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
            self.Wal = self._compute_Wal()
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
        Wal, p = window_matrix(self.co.k, tab[:, 0], tab[:, 1:].T, m["Na"], m["Nl"], accboost=m["accboost"], Nmax=m["Nmax"],
                               xmin_factor=m["xmin_factor"], xmax_factor=m["xmax_factor"], bias=m["bias"],
                               window_param=m["window_param"], pmax=m["pmax"])
        assert np.array_equal(p, self.p)
        return Wal

    def Window(self, bird):
        """Convolve P11l, Pctl, Ploopl (and Pstl if window_st) in place (reference window.py:389-415)."""
        from .pybird import engine_for

        eng = engine_for(bird.co)
        if self._op is None or self._op[0] is not eng:
            self._op = (eng, eng.add_operator(self.Wfold))
        keep = bird.Pstl
        out = apply_operator_to_birdlike(eng, self._op[1], bird)
        bird.P11l, bird.Pctl, bird.Ploopl = out["P11l"], out["Pctl"], out["Ploopl"]
        bird.Pstl = out["Pstl"] if self.window_st else keep
        if "PctNNLOl" in out:
            bird.PctNNLOl = out["PctNNLOl"]
        if self.snapshot:
            bird.create_snapshot("window")
