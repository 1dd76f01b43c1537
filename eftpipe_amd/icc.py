"""Integral-constraint correction (same surface as reference eftpipe/icc.py:119-500, eftpipe/fftlog2d.py).

Per evaluation the correction is one more linear operator on the templates plus a constant term: with an ``IntegralConstraint`` the
window plugin applies  P -> W P - W_ic P  and  Picc -= Pshot * P_SN  (reference window.py:393-406).  Both matrices are folded with the
k -> p cubic spline into ONE dense operator, so ``Window(icc=...)`` costs exactly what ``Window`` costs: one FP64-MFMA product on the
device (gemm_rows_kernel).

Init (host, once; the results are cached in the reference's own ``.npz`` + ``.json`` format, keys ``PSN`` [Na, Nk], ``Wal`` [Na, Nl, Nk, Np]):
``_compute_PSN`` is a 1-D FFTLog; ``_compute_Wal`` is the 2-D FFTLog of the configuration-space panel.  The reference resamples that panel
with ``scipy.interpolate.interp2d`` (fftlog2d.py:75), removed from SciPy 1.14 -- the reference itself cannot run this step on a current
SciPy.  This module uses the replacement SciPy's migration guide prescribes for regular grids, ``RectBivariateSpline(kx=3, ky=3, s=0)``
(the same FITPACK surface fit interp2d called); PARITY of that one step is therefore UNPINNED: it is checked against
the oracle's restatement only, everything else (cache files written by the reference, PSN, mask, fold, convolution) is ordinary arithmetic.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
from scipy.interpolate import RectBivariateSpline
from scipy.special import loggamma

from ._log import HasLogger
from .tables import FFTLogOperator, edge_window, window_fold, window_pgrid
from .window import MetaInfoError


def bessel_matrix(p, l):
    """int_0^inf s^(2+p) j_l(s) ds; the k dependence is k^(-3-p) (reference fftlog2d.py:13-40)"""
    return np.exp((1.0 + p) * np.log(2.0) + loggamma(0.5 * (3.0 + l + p)) - loggamma(0.5 * (l - p))) * np.sqrt(np.pi)


class FFTLog2D:
    """2-D FFTLog (same surface as reference fftlog2d.py:43-166)"""

    def __init__(self, Nxmax, Nymax, xmin, xmax, ymin, ymax, xbias, ybias):
        self.Nxmax, self.Nymax = Nxmax, Nymax
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.xbias, self.ybias = xbias, ybias
        self.dx = np.log(xmax / xmin) / (Nxmax - 1)
        self.dy = np.log(ymax / ymin) / (Nymax - 1)
        self.x = np.geomspace(xmin, xmax, Nxmax, dtype=np.float64)
        self.y = np.geomspace(ymin, ymax, Nymax, dtype=np.float64)
        self.xPow = xbias + 2j * np.pi * np.fft.fftfreq(Nxmax, d=self.dx)
        self.yPow = ybias + 2j * np.pi * np.fft.fftfreq(Nymax, d=self.dy)

    def Coef(self, xin, yin, zin, extrap="padding", window=None):
        if extrap == "extrap":
            raise NotImplementedError
        if extrap != "padding":
            raise ValueError("extra should be 'extrap' or 'padding'")
        # interp2d(xin, yin, zin, kind="cubic") read zin as z[j, i] = f(x_i, y_j) and returned [len(y), len(x)]
        f = RectBivariateSpline(xin, yin, np.asarray(zin, dtype=np.float64).T, kx=3, ky=3, s=0)
        farr = np.zeros((self.Nxmax, self.Nymax))
        maskx = (self.x >= xin[0]) & (self.x <= xin[-1])
        masky = (self.y >= yin[0]) & (self.y <= yin[-1])
        farr[np.outer(maskx, masky)] = f(self.x[maskx], self.y[masky]).T.reshape(-1)  # same flattening as the reference (fftlog2d.py:85)
        out = (np.fft.fft2(farr * np.outer((self.x / self.x[0]) ** (-self.xbias), (self.y / self.y[0]) ** (-self.ybias))) / (self.Nxmax * self.Nymax)
               / np.outer(self.x[0] ** self.xPow, self.y[0] ** self.yPow))
        if window is not None:
            out = out * self.window(window)
        return out

    def spherical_transform(self, xin, yin, zin, extrap="padding", window=None, *, k1, k2, l1, l2):
        C = self.Coef(xin, yin, zin, extrap=extrap, window=window)
        M1 = np.asarray(k1)[:, None] ** (-3.0 - self.xPow)[None, :] * bessel_matrix(self.xPow, l1)
        M2 = np.asarray(k2)[:, None] ** (-3.0 - self.yPow)[None, :] * bessel_matrix(self.yPow, l2)
        return np.einsum("mn,pm,qn->pq", C, M1, M2, optimize=True).real

    def window(self, window):
        def one(N):
            f = np.fft.fftfreq(N, d=1.0)
            nf = int((1 - window) * N / 2)
            if nf >= N // 2:
                nf -= 1
            left, right, fmin = f[-nf], f[nf], np.min(f)
            w = np.ones(N)
            tl = (f[f < left] - fmin) / (left - fmin)
            tr = (-fmin - f[f > right]) / (-fmin - right)
            w[f < left] = tl - np.sin(2 * np.pi * tl) / (2 * np.pi)
            w[f > right] = tr - np.sin(2 * np.pi * tr) / (2 * np.pi)
            return w

        return np.outer(one(self.Nxmax), one(self.Nymax))


def read_configspace_IC_file(file, info=print, warning=print):
    """(reference icc.py:81-105)"""
    file = Path(file)
    if not file.exists():
        raise FileNotFoundError(f"File {file} does not exist")
    if file.suffix in (".txt", ".TXT", ".dat", ".ascii"):
        warning("reading csv file is very slow, please consider using npy file instead")
        return np.loadtxt(file)
    if file.suffix == ".npy":
        return np.load(file)
    raise ValueError(f"File {file} has unsupported suffix {file.suffix}")


def ICpannel_to_ndarray(arr, inorder=False, info=print):
    """rows (l1, l2, s1, s2, value) -> panel [l1, l2, s1, s2] and its axes (reference icc.py:108-121)"""
    l1, l2, s1, s2 = (np.unique(c) for c in arr.T[:4])
    assert l1.size * l2.size * s1.size * s2.size == arr.shape[0]
    meta = dict(l1=l1, l2=l2, s1=s1, s2=s2)
    if inorder:
        return arr.T[4].reshape((l1.size, l2.size, s1.size, s2.size)), meta
    idx = [np.searchsorted(ax, col) for ax, col in zip((l1, l2, s1, s2), arr.T[:4])]
    ret = np.zeros((l1.size, l2.size, s1.size, s2.size))
    ret[tuple(idx)] = arr.T[4]
    return ret, meta


class IntegralConstraint(HasLogger):
    def __init__(self, Pshot, icc_fourier_file=None, icc_configspace_SN_file=None, icc_configspace_IC_file=None, inorder=False, co=None,
                 load=True, save=True, check_meta=True, Na=None, Nl=None, pmax=0.3, accboost=1, withmask=True, windowk=0.05, Nmax=4096,
                 bias=-2.1, window_param=1, Nxmax=4096, Nymax=4096, xbias=-2.0, ybias=-2.0, windowxy_param=1, name="eftpipe.icc",
                 snapshot=False):
        from . import pybird

        self.set_logger(name=name)
        self.co = pybird.common if co is None else co
        if all(x is None for x in (icc_fourier_file, icc_configspace_SN_file, icc_configspace_IC_file)):
            raise ValueError("No ICC file specified")
        res = lambda p: Path(p).resolve() if p else None
        self.icc_fourier_file, self.icc_configspace_SN_file, self.icc_configspace_IC_file = res(icc_fourier_file), res(icc_configspace_SN_file), res(icc_configspace_IC_file)
        self.inorder, self._load, self.check_meta = inorder, load, check_meta
        self._save = save if self.icc_fourier_file else False
        self._create_meta = True
        self.withmask, self.windowk = withmask, windowk
        Na = Na or self.co.Nl
        Nl = Nl or self.co.Nl
        if Na > self.co.Nl or Nl > self.co.Nl:
            raise ValueError(f"request Na={Na}, Nl={Nl} while bird only compute Nl up to {self.co.Nl}")
        if Na > Nl:
            raise ValueError(f"dangerous settings Na={Na} > Nl={Nl}")
        self.p = window_pgrid(pmax, accboost)
        s_or_none = lambda x: str(x) if x else None
        self.meta = dict(Na=Na, Nl=Nl, pmax=pmax, accboost=accboost, Nmax=Nmax, bias=bias, window_param=window_param, Nxmax=Nxmax, Nymax=Nymax,
                         xbias=xbias, ybias=ybias, windowxy_param=windowxy_param, icc_configspace_SN_file=s_or_none(self.icc_configspace_SN_file),
                         icc_configspace_IC_file=s_or_none(self.icc_configspace_IC_file), k=self.co.k.tolist())
        self.Pshot = Pshot
        self.PSN, self.Wal = self._loadicc()
        if self.PSN is None:
            self.PSN = self._compute_PSN()
            self.Wal = self._compute_Wal()
        self.Wfold, self.Waldk = window_fold(self.co.k, self.Wal, self.p, windowk=windowk, withmask=withmask)
        if self._save:
            self._saveicc()
        self.PSN = self.PSN * Pshot  # always need Pshot (reference icc.py:278)
        self.snapshot = snapshot

    # ---- cache files in the reference's format (icc.py:283-357, 465-474)
    def _loadicc(self):
        PSN = Wal = None
        f = self.icc_fourier_file
        if self._load and f is not None:
            try:
                data = np.load(f)
                PSN, Wal = data["PSN"], data["Wal"]
            except (OSError, TypeError):
                self.mpi_warning("Cannot load icc from %s", f)
            else:
                if Wal.shape[1] != self.meta["Nl"]:  # retry with the suffix the reference uses for another Nl
                    PSN = Wal = None
                    f = self.icc_fourier_file = f.with_name(f.stem + f"_Nl{self.meta['Nl']}.npz")
                    try:
                        data = np.load(f)
                        PSN, Wal = data["PSN"], data["Wal"]
                    except OSError:
                        self.mpi_warning("Cannot load icc from %s", f)
            if PSN is not None and self.check_meta:
                meta_file = f.with_suffix(".json")
                if not meta_file.exists():
                    self._create_meta = False
                else:
                    with meta_file.open() as fh:
                        meta = json.load(fh)
                    for key in ("icc_configspace_SN_file", "icc_configspace_IC_file"):
                        if self.meta[key] is None:
                            self.meta[key] = meta[key]
                    if meta != self.meta:
                        raise MetaInfoError(f"inconsistent meta info\nloaded icc's meta:\n{meta}\nexpect:\n{self.meta}")
        if PSN is not None:
            if Wal.shape != (self.meta["Na"], self.meta["Nl"], self.co.Nk, self.p.size) or PSN.shape != (self.meta["Na"], self.co.Nk):
                raise MetaInfoError(f"cached icc has shapes {PSN.shape}, {Wal.shape}")
            self._save = False
        return PSN, Wal

    def _saveicc(self):
        np.savez(self.icc_fourier_file, PSN=self.PSN, Wal=self.Wal)
        if self._create_meta:
            with self.icc_fourier_file.with_suffix(".json").open("w") as fh:
                json.dump(self.meta, fh, indent=2)

    def _compute_PSN(self):
        """(reference icc.py:359-403)"""
        Na = self.meta["Na"]
        if self.icc_configspace_SN_file is None:
            raise ValueError("please specify icc_configspace_SN_file")
        data = np.loadtxt(self.icc_configspace_SN_file)
        while data[0, 0] == 0.0:
            data = data[1:, :]
        try:
            data = data[:, : 1 + Na]
            s, xi = data[:, 0], data[:, 1:].T
            if xi.shape[0] != Na:
                raise IndexError
        except IndexError as ex:
            raise TypeError("loaded icc_configspace_SN_file has unexpected shape") from ex
        op = FFTLogOperator(self.meta["Nmax"], s[0], s[-1], self.meta["bias"], s, self.meta["window_param"], extrap=("padding", "padding"))
        coef = xi @ op.G.T                                               # [a, n]
        power = self.co.k[:, None] ** (-op.Pow[None, :] - 3.0)           # [k, n]
        mat = np.array([bessel_matrix(op.Pow, ell) for ell in range(0, 2 * Na, 2)])
        PSN = np.einsum("an,kn,an->ak", coef, power, mat, optimize=True).real
        return PSN * 4 * np.pi * np.array([(-1j) ** ell for ell in range(0, 2 * Na, 2)]).real[:, None]

    def _compute_Wal(self):
        """(reference icc.py:405-450)"""
        Na, Nl = self.meta["Na"], self.meta["Nl"]
        if self.icc_configspace_IC_file is None:
            raise ValueError("please specify icc_configspace_IC_file")
        data = read_configspace_IC_file(self.icc_configspace_IC_file, self.mpi_info, self.mpi_warning)
        data, pm = ICpannel_to_ndarray(data, self.inorder, self.mpi_info)
        s1, s2 = pm["s1"], pm["s2"]
        m = self.meta
        fft2d = FFTLog2D(m["Nxmax"], m["Nymax"], 1e-3, s1[-1], 1e-3, s2[-1], m["xbias"], m["ybias"])
        Wal = np.empty((Na, Nl, self.co.k.size, self.p.size))
        for a in range(Na):
            for l in range(Nl):
                Wal[a, l] = fft2d.spherical_transform(s1, s2, data[a, l], extrap="padding", window=m["windowxy_param"], k1=self.co.k, k2=self.p,
                                                      l1=2 * a, l2=2 * l)
                Wal[a, l] *= 8.0 * np.real((-1j) ** (2 * a) * (1j) ** (2 * l)) / (2 * (2 * l) + 1) * self.p**2
        return Wal

    def integrWindow(self, P):
        """Host form (reference icc.py:476-489): [Nl, n, Nk] -> [Na, n, Nk]"""
        return np.einsum("alxk,lnk->anx", self.Wfold, np.asarray(P))

    def icc(self, bird):
        """(reference icc.py:491-500: "this method is wrong and deprecated") -- kept with the reference's arithmetic"""
        self.mpi_warning("This method is wrong and deprecated, please don't use it!")
        for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
            setattr(bird, n, getattr(bird, n) - self.integrWindow(getattr(bird, n)))
        if bird.co.with_NNLO:
            bird.PctNNLOl = bird.PctNNLOl - self.integrWindow(bird.PctNNLOl)
        bird.Picc = bird.Picc - self.PSN
        if self.snapshot:
            bird.create_snapshot("icc")
