"""Oracle: integral-constraint correction (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED.  The reference's ``FFTLog2D.Coef`` resamples the configuration-space panel with ``scipy.interpolate.interp2d``
(reference eftpipe/fftlog2d.py:75), which no longer exists in the SciPy of this image (removed in 1.14): the reference itself cannot run
this stage here, so no reference-generated fixture can exist.  This file restates reference eftpipe/icc.py:119-500 and
eftpipe/fftlog2d.py:13-166 line by line with the ONE documented substitution SciPy's own migration guide prescribes for data on a
regular grid, ``RectBivariateSpline(x, y, z.T, kx=3, ky=3, s=0)`` (interp2d(kind="cubic") called the same FITPACK ``regrid_smth`` surface
fit).  The product (eftpipe_amd/icc.py) is tested against this restatement only; everything that does not pass through the removed
function (PSN: a 1-D FFTLog; the mask / dp weights; the convolution; the cache format) is ordinary, checkable arithmetic.
"""
from __future__ import annotations

import numpy as np
from scipy.interpolate import RectBivariateSpline, interp1d
from scipy.special import loggamma

from .fftlog import FFTLogGrid


def bessel_matrix(p, l):
    """int_0^inf s^(2+p) j_l(s) ds (reference fftlog2d.py:13-40)"""
    return np.exp((1.0 + p) * np.log(2.0) + loggamma(0.5 * (3.0 + l + p)) - loggamma(0.5 * (l - p))) * np.sqrt(np.pi)


def taper2d(Nx, Ny, window):
    """(reference fftlog2d.py:131-166)"""
    def one(N):
        f = np.fft.fftfreq(N, d=1.0)
        nf = int((1 - window) * N / 2)
        if nf >= N // 2:
            nf -= 1
        left, right = f[-nf], f[nf]
        fmin = np.min(f)
        fmax = -fmin
        w = np.ones(N)
        tl = (f[f < left] - fmin) / (left - fmin)
        tr = (fmax - f[f > right]) / (fmax - right)
        w[f < left] = tl - np.sin(2 * np.pi * tl) / (2 * np.pi)
        w[f > right] = tr - np.sin(2 * np.pi * tr) / (2 * np.pi)
        return w

    return np.outer(one(Nx), one(Ny))


class FFTLog2D:
    """(reference fftlog2d.py:43-129)"""

    def __init__(self, Nxmax, Nymax, xmin, xmax, ymin, ymax, xbias, ybias):
        self.Nxmax, self.Nymax, self.xbias, self.ybias = Nxmax, Nymax, xbias, ybias
        self.dx, self.dy = np.log(xmax / xmin) / (Nxmax - 1), np.log(ymax / ymin) / (Nymax - 1)
        self.x, self.y = np.geomspace(xmin, xmax, Nxmax), np.geomspace(ymin, ymax, Nymax)
        self.xPow = xbias + 2j * np.pi * np.fft.fftfreq(Nxmax, d=self.dx)
        self.yPow = ybias + 2j * np.pi * np.fft.fftfreq(Nymax, d=self.dy)

    def coef(self, xin, yin, zin, window=None):
        """zin[iy, ix] on (xin, yin), as interp2d took it (reference fftlog2d.py:67-104; extrap='padding')"""
        f = RectBivariateSpline(xin, yin, np.asarray(zin).T, kx=3, ky=3, s=0)  # <- the substitution for interp2d(kind="cubic")
        farr = np.zeros((self.Nxmax, self.Nymax))
        mx = (self.x >= xin[0]) & (self.x <= xin[-1])
        my = (self.y >= yin[0]) & (self.y <= yin[-1])
        # interp2d returned [len(y), len(x)] and the reference wrote its flattened values into the (x, y)-ordered mask (fftlog2d.py:85):
        # for a panel that is symmetric in (s1, s2) the two orders coincide; the same flattening is kept here
        vals = f(self.x[mx], self.y[my]).T
        farr[np.outer(mx, my)] = vals.reshape(-1)
        out = (np.fft.fft2(farr * np.outer((self.x / self.x[0]) ** (-self.xbias), (self.y / self.y[0]) ** (-self.ybias))) / (self.Nxmax * self.Nymax)
               / np.outer(self.x[0] ** self.xPow, self.y[0] ** self.yPow))
        if window is not None:
            out = out * taper2d(self.Nxmax, self.Nymax, window)
        return out

    def spherical_transform(self, xin, yin, zin, window, k1, k2, l1, l2):
        """(reference fftlog2d.py:106-129)"""
        C = self.coef(xin, yin, zin, window)
        M1 = k1[:, None] ** (-3.0 - self.xPow)[None, :] * bessel_matrix(self.xPow, l1)
        M2 = k2[:, None] ** (-3.0 - self.yPow)[None, :] * bessel_matrix(self.yPow, l2)
        return np.einsum("mn,pm,qn->pq", C, M1, M2, optimize=True).real


def compute_psn(k, s, xi, Na, Nmax=4096, bias=-2.1, window_param=1):
    """P_SN[a, k] from the configuration-space shot-noise term xi[a, s] (reference icc.py:359-403), without the Pshot factor"""
    fft = FFTLogGrid(Nmax, s[0], s[-1], bias)
    coef = fft.coef(s, xi[:Na], extrap="padding", window=window_param)
    power = k[:, None] ** (-fft.Pow[None, :] - 3.0)
    mat = np.array([bessel_matrix(fft.Pow, ell) for ell in range(0, 2 * Na, 2)])
    psn = np.einsum("an,kn,an->ak", coef, power, mat, optimize=True).real
    return psn * 4 * np.pi * np.array([(-1j) ** ell for ell in range(0, 2 * Na, 2)]).real[:, None]


def compute_wal(k, p, s1, s2, panel, Na, Nl, Nxmax=4096, Nymax=4096, xbias=-2.0, ybias=-2.0, windowxy_param=1):
    """W^ic_al(k, p) from the panel[l1, l2, s1, s2] (reference icc.py:405-450)"""
    fft2d = FFTLog2D(Nxmax, Nymax, 1e-3, s1[-1], 1e-3, s2[-1], xbias, ybias)
    Wal = np.empty((Na, Nl, k.size, p.size))
    for a in range(Na):
        for l in range(Nl):
            Wal[a, l] = fft2d.spherical_transform(s1, s2, panel[a, l], windowxy_param, k, p, 2 * a, 2 * l)
            Wal[a, l] *= 8.0 * np.real((-1j) ** (2 * a) * (1j) ** (2 * l)) / (2 * (2 * l) + 1) * p**2
    return Wal


def waldk(k, p, Wal, windowk=0.05, withmask=True):
    """(reference icc.py:452-463)"""
    W = Wal
    if withmask:
        pg, kg = np.meshgrid(p, k, indexing="ij")
        W = np.einsum("alkp,pk->alkp", Wal, (pg > kg - windowk) & (pg < kg + windowk))
    return np.einsum("alkp,p->alkp", W, np.concatenate([[0.0], np.diff(p)]))


def integr_window(k, p, Waldk, P):
    """(reference icc.py:476-489)"""
    Pk = interp1d(k, P, axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")(p)
    return np.einsum("alkp,lsp->ask", Waldk, Pk, optimize=True)


def window_with_icc(st, k, win_p, win_Waldk, icc_p, icc_Waldk, psn, window_st=True):
    """Window.Window with an IntegralConstraint (reference window.py:393-406): P -> W P - W_ic P, Picc -= PSN"""
    out = dict(st)
    for n in ("P11l", "Pctl", "Ploopl") + (("Pstl",) if window_st else ()):
        out[n] = integr_window(k, win_p, win_Waldk, st[n]) - integr_window(k, icc_p, icc_Waldk, st[n])
    out["Picc"] = st["Picc"] - psn
    return out
