"""Oracle: FFTLog power-law decomposition (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows reference eftpipe/pybird/fftlog.py:17-166.
"""
from __future__ import annotations

import numpy as np
from numpy.fft import rfft
from scipy.interpolate import CubicSpline


def edge_window(N, window=1):
    """Taper sending the outermost FFTLog coefficients to zero (reference fftlog.py:17-40).

    ``n_cut = int(window*N//2.)`` -- note the floor division happens on ``window*N`` (fftlog.py:23).
    """
    n = np.arange(-N // 2, N // 2 + 1)
    n_cut = N // 2 if window == 1 else int(window * N // 2.0)
    hi, lo = n[-1] - n_cut, n[0] + n_cut
    W = np.ones(n.size)
    sel = n > hi
    th = (n[-1] - n[sel]) / float(n[-1] - hi - 1)
    W[sel] = th - np.sin(2 * np.pi * th) / (2 * np.pi)
    sel = n < lo
    th = (n[sel] - n[0]) / float(lo - n[0] - 1)
    W[sel] = th - np.sin(2 * np.pi * th) / (2 * np.pi)
    return W


class FFTLogGrid:
    """Log grid, complex powers and normalisation (reference fftlog.py:59-82)."""

    def __init__(self, Nmax, xmin, xmax, bias):
        if Nmax % 2:
            raise ValueError(f"expected even Nmax, instead of Nmax={Nmax}")
        self.Nmax, self.xmin, self.xmax, self.bias = Nmax, xmin, xmax, bias
        self.dx = np.log(xmax / xmin) / (Nmax - 1.0)
        i = np.arange(Nmax)
        self.x = xmin * np.exp(i * self.dx)
        m = np.arange(Nmax + 1)
        self.Pow = bias + 1j * 2.0 * np.pi / (Nmax * self.dx) * (m - Nmax / 2.0)
        self.coef_factor = xmin ** (-self.Pow) / float(Nmax)

    def coef(self, xin, f, extrap="extrap", window=1, kernel=None):
        """Power-law coefficients of ``f`` sampled on ``xin`` (reference fftlog.py:84-166).

        ``f`` may carry leading batch axes (only with ``extrap='padding'``, as on the hot path).
        """
        f = np.asarray(f, dtype=float)
        if not isinstance(extrap, tuple):
            extrap = (extrap, extrap)
        if any(e not in ("padding", "extrap") for e in extrap):
            raise ValueError(f"unexpected extrap = {extrap}")
        N = self.Nmax
        spline = CubicSpline(xin, f, axis=-1, extrapolate=False)
        fx = np.zeros(f.shape[:-1] + (N,))
        lo = np.searchsorted(self.x, xin[0])
        hi = np.searchsorted(self.x, xin[-1], side="right")
        tilt = np.exp(-self.bias * np.arange(lo, hi) * self.dx)
        if kernel is not None:
            tilt = tilt * kernel(self.x[lo:hi])
        fx[..., lo:hi] = spline(self.x[lo:hi]) * tilt
        if extrap[0] == "extrap" and xin[0] > self.x[0]:
            slope = (np.log(f[1]) - np.log(f[0])) / (np.log(xin[1]) - np.log(xin[0]))
            amp = f[0] / xin[0] ** slope
            fx[..., :lo] = amp * self.x[:lo] ** slope * np.exp(-self.bias * np.arange(0, lo) * self.dx)
        if extrap[1] == "extrap" and xin[-1] < self.x[-1]:
            slope = (np.log(f[-1]) - np.log(f[-2])) / (np.log(xin[-1]) - np.log(xin[-2]))
            amp = f[-1] / xin[-1] ** slope
            fx[..., hi:] = amp * self.x[hi:] ** slope * np.exp(-self.bias * np.arange(hi, N) * self.dx)
        half = rfft(fx, axis=-1)
        c = np.empty(f.shape[:-1] + (N + 1,), dtype=complex)
        c[..., : N // 2] = np.conj(half[..., 1:][..., ::-1])
        c[..., N // 2 :] = half
        c *= self.coef_factor
        if window is not None:
            c *= edge_window(N, window)
        else:
            c[..., 0] /= 2.0
            c[..., N] /= 2.0
        return c
