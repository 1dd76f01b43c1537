"""Host-side, init-time construction of every constant table the HIP engine consumes.

The reference evaluates each stage as spline -> FFTLog -> complex einsum.  All of those steps are
*linear* in the sampled function once the grids are fixed, and every complex sum on the path is
conjugate-symmetric (Pow[N-n] = conj(Pow[n])), so the engine pre-multiplies them here into REAL
operators and real-reduced pair tables; the device then only runs real FP64 contractions:

  * ``Sk``        cubic not-a-knot spline  kin -> k          (Bird.__init__, pybird.py:694-695)
  * ``G*``/``E*`` FFTLog.Coef as a matrix + power-law tails  (fftlog.py:84-166)
  * ``ad``        the loop matrices in anti-diagonal form: sum_{nm} x_n M[n,m] x_m = sum_j k^{..j} S[j] with the
                  k-independent S[j] = sum_{n+m=j} c_n c_m M[n,m]           (pybird.py:1074-1078, 1103-1125)
  * ``syn_*``     the 513-term (257 for the single sums) real synthesis bases k^{..}{1, cos, sin}  (pybird.py:1058-1064)
  * ``H``         resum FFTLog(192) + Bessel sum as a real [Na, Nk, Ns] operator (pybird.py:1361-1365,1409-1411)
  * ``BX/BY``     IR filters X(s), Y(s) as real [Ns, Nkin] operators + tails (pybird.py:1316-1353)
  * spline LU     tridiagonal factors of the not-a-knot system on the k grid (AP, pybird.py:1586-1593)
  * ``Wfold``     window / binning / chained projections folded with their splines (window.py:371-387,
                  binning.py:131-144, chained.py:32-68)

This is host *setup* code (NumPy/SciPy, float64/complex128); nothing here runs per evaluation.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
from scipy.interpolate import CubicSpline
from scipy.special import legendre, spherical_jn

from . import loopmath as lm

NS = 80          # |sbird|
NPOW = 257       # NFFT + 1 for the loop FFTLog
NHALF = 128


# ----------------------------------------------------------------------------- FFTLog as an operator
def edge_window(N, window):
    """Coefficient taper (reference fftlog.py:17-40); ``window=None`` -> endpoints halved (:162-164)."""
    if window is None:
        w = np.ones(N + 1)
        w[0] = w[N] = 0.5
        return w
    n = np.arange(-N // 2, N // 2 + 1)
    n_cut = N // 2 if window == 1 else int(window * N // 2.0)
    hi, lo = n[-1] - n_cut, n[0] + n_cut
    w = np.ones(n.size)
    sel = n > hi
    th = (n[-1] - n[sel]) / float(n[-1] - hi - 1)
    w[sel] = th - np.sin(2 * np.pi * th) / (2 * np.pi)
    sel = n < lo
    th = (n[sel] - n[0]) / float(lo - n[0] - 1)
    w[sel] = th - np.sin(2 * np.pi * th) / (2 * np.pi)
    return w


class FFTLogOperator:
    """``Coef = G @ f + E_hi @ (A_hi x_hi^n_hi) [+ E_lo @ ...]`` for samples ``f`` on fixed ``xin``.

    Same arithmetic as reference fftlog.py:59-166: log grid, tilt by exp(-bias i dx), real FFT with
    conjugate mirroring, ``xmin^-Pow / N`` normalisation and the edge window; the spline resampling
    (CubicSpline, not-a-knot, no extrapolation) is folded in, and the power-law extrapolations keep
    their two scalar parameters (slope, amplitude) outside the matrices.
    """

    def __init__(self, N, xmin, xmax, bias, xin, window, extrap=("extrap", "extrap"), kernel=None):
        self.N, self.bias = N, bias
        self.dx = np.log(xmax / xmin) / (N - 1.0)
        i = np.arange(N)
        self.x = xmin * np.exp(i * self.dx)
        m = np.arange(N + 1)
        self.dpow = 2.0 * np.pi / (N * self.dx)
        self.Pow = bias + 1j * self.dpow * (m - N / 2.0)
        coef_factor = xmin ** (-self.Pow) / float(N)
        lo = int(np.searchsorted(self.x, xin[0]))
        hi = int(np.searchsorted(self.x, xin[-1], side="right"))
        self.lo, self.hi = lo, hi
        tilt = np.exp(-bias * i * self.dx)
        phase = np.exp(-2j * np.pi * (((m[:, None] - N // 2) * i[None, :]) % N) / N)
        D = phase * (coef_factor * edge_window(N, window))[:, None]
        if xin is not None:
            S = CubicSpline(xin, np.eye(len(xin)), axis=0, extrapolate=False)(self.x[lo:hi])
            t = tilt[lo:hi]
            if kernel is not None:
                t = t * kernel(self.x[lo:hi])
            self.G = D[:, lo:hi] @ (t[:, None] * S)
        self.low_active = extrap[0] == "extrap" and xin[0] > self.x[0]
        self.high_active = extrap[1] == "extrap" and xin[-1] < self.x[-1]
        self.E_lo = D[:, :lo] * tilt[:lo] if self.low_active else np.zeros((N + 1, 0), complex)
        self.E_hi = D[:, hi:] * tilt[hi:] if self.high_active else np.zeros((N + 1, 0), complex)
        self.lnx_lo = np.log(self.x[:lo]) if self.low_active else np.zeros(0)
        self.lnx_hi = np.log(self.x[hi:]) if self.high_active else np.zeros(0)


# ----------------------------------------------------------------------------- real reduction
def loop_basis(M22, tol=1e-11):
    """The 28 loop matrices M22[b] span a space of dimension 7 only (they are built from a handful of F2/G2
    angular structures): pick a well-conditioned subset by column-pivoted QR and express every M22[b] as a real
    combination of it.  -> (basis indices [nb], comb [28, nb]);  max |comb| ~ 1, residual ~ 1e-15 (asserted)."""
    from scipy.linalg import qr

    A = M22.reshape(M22.shape[0], -1)
    _, Rq, piv = qr(A.T, mode="economic", pivoting=True)
    d = np.abs(np.diag(Rq))
    rank = int(np.sum(d > tol * d[0]))
    basis = np.sort(piv[:rank])
    X = np.linalg.lstsq(A[basis].T, A.T, rcond=None)[0].T
    resid = np.max(np.abs(X @ A[basis] - A)) / np.max(np.abs(A))
    if resid > 1e-13 or np.max(np.abs(X.imag)) > 1e-12 or rank > 16:
        raise ValueError(f"loop-matrix basis reduction failed: rank {rank}, residual {resid:.2e}")
    return basis, np.ascontiguousarray(X.real)


def antidiagonal_tables(M22b, M13b):
    """The one-loop double sums in anti-diagonal form.

    With x_n(k) = c_n k^{Pow_n} and Pow_n = bias + i dpow (n - N/2) the k dependence of a pair (n, m) is
    k^{2 bias + i dpow (n + m - N)}: it depends on n + m only.  Hence for any loop matrix M

        sum_{n,m} x_n M[n,m] x_m  =  sum_{j'=-N..N}  k^{2 bias + i dpow j'}  S[j'],     S[j'] = sum_{n+m=N+j'} c_n c_m M[n,m],

    i.e. ONE k-independent pass over the 257^2 matrix entries per cosmology (the anti-diagonal sums S) followed by a
    513-term synthesis per k -- instead of 257^2 terms per k (reference pybird.py:1074-1078, 1103-1125).  The same S
    serve P22 at every k and, multiplied by Ml(n + m), C22 / C13 at every s and l.  S[-j'] = conj S[j'], so only
    j' = 0..N is kept.  Exact regrouping of the same sum (verified to 1e-15 against the reference in tests).

    -> AD[c, j', t] complex, t = 0..128: the symmetrised weight of the pair (n, m) = (j' + t, N - t), n <= m, of matrix c
       (M22 basis first: M[n,m] + M[m,n], or M[n,n]; then the row-broadcast M13 basis: V[n] + V[m], or V[n]); zero
       beyond the end of the anti-diagonal."""
    N = NPOW - 1
    mats = [np.asarray(M22b)]
    if M13b is not None:
        mats.append(np.asarray(M13b)[:, :, None] * np.ones((1, 1, NPOW)))
    M = np.concatenate(mats)
    Ms = M + np.swapaxes(M, 1, 2)
    idx = np.arange(NPOW)
    Ms[:, idx, idx] = M[:, idx, idx]
    AD = np.zeros((M.shape[0], NPOW, NHALF + 2), dtype=complex)
    for jp in range(NPOW):
        cnt = ((N + jp) >> 1) - jp + 1
        tt = np.arange(cnt)
        AD[:, jp, :cnt] = Ms[:, jp + tt, N - tt]
    return AD


def synthesis_table(lnx, power, dpow, nharm, sign):
    """Real basis of  x^power [ Z_0 + 2 Re sum_{j=1..nharm} Z_j e^{sign i dpow j ln x} ]  for coefficient rows stored as
    (Re Z_0, Re Z_1, Im Z_1, Re Z_2, Im Z_2, ...):  -> [Kpad, len(x)], Kpad = 1 + 2 nharm rounded up to 48 (zero rows: the LDS
    chunk of synth_kernel)."""
    K = 1 + 2 * nharm
    out = np.zeros(((K + 47) // 48 * 48, lnx.size))
    amp = np.exp(power * lnx)
    out[0] = amp
    j = np.arange(1, nharm + 1)
    th = dpow * np.outer(j, lnx)
    out[1:K:2] = 2.0 * amp * np.cos(th)
    out[2:K:2] = -sign * 2.0 * amp * np.sin(th)
    return out


# ----------------------------------------------------------------------------- cubic spline (not-a-knot)
def spline_factors(x):
    """Tridiagonal system of scipy's not-a-knot CubicSpline for knot derivatives, pre-factored.

    Unknowns s_i = S'(x_i).  Interior rows: dx_i s_{i-1} + 2(dx_{i-1}+dx_i) s_i + dx_{i-1} s_{i+1}
    = 3(dx_i slope_{i-1} + dx_{i-1} slope_i); end rows are the not-a-knot conditions.  Returns the
    Thomas-algorithm factors (lower, inv_pivot, cprime) so the device only substitutes.
    """
    x = np.asarray(x, dtype=float)
    n = x.size
    dx = np.diff(x)
    lower, diag, upper = np.zeros(n), np.zeros(n), np.zeros(n)
    diag[1:-1] = 2.0 * (dx[:-1] + dx[1:])
    upper[1:-1] = dx[:-1]
    lower[1:-1] = dx[1:]
    diag[0], upper[0] = dx[1], x[2] - x[0]
    diag[-1], lower[-1] = dx[-2], x[-1] - x[-3]
    inv, cp = np.zeros(n), np.zeros(n)
    piv = diag[0]
    inv[0], cp[0] = 1.0 / piv, upper[0] / piv
    for i in range(1, n):
        piv = diag[i] - lower[i] * cp[i - 1]
        inv[i] = 1.0 / piv
        cp[i] = upper[i] / piv
    return dict(dx=dx, lower=lower, inv=inv, cp=cp)


SPL_HB = 32  # half-width of the banded knot-derivative operator (must match csrc/eftb_kernels.hpp)


def spline_derivative_band(x):
    """Knot derivatives of the not-a-knot cubic spline as a banded operator on the values:
    s_i = sum_d band[d, i] * y[i + d - SPL_HB].  The exact operator is  A^-1 R  (A, R tridiagonal); its entries
    decay like (2 - sqrt 3)^|i-j| = 0.27^|i-j|, so half-width 32 truncates below 1e-18 of the peak."""
    x = np.asarray(x, dtype=float)
    n = x.size
    S = CubicSpline(x, np.eye(n), axis=0)(x, 1)            # [n(i), n(j)] exact dense operator
    band = np.zeros((2 * SPL_HB + 1, n))
    for d in range(2 * SPL_HB + 1):
        j = np.arange(n) + d - SPL_HB
        ok = (j >= 0) & (j < n)
        band[d, ok] = S[np.arange(n)[ok], j[ok]]
    return band


def bspline_tables(x):
    """The same not-a-knot cubic spline in B-spline form (round 3: ONE number per knot instead of the Hermite pair -- the AP kernels move and
    keep half the bytes).  ``cband[d, i]``: coefficient c_i = sum_d cband[d, i] y[i + d - SPL_HB] (the inverse collocation matrix: same decay
    as the slope operator, same band).  ``local[i, e, p]``: on the interval [x_i, x_i+1] (end pieces extrapolate) the spline is
    sum_e c[J_i + e] sum_p local[i, e, p] (x - x_i)^p with J_i = clip(i - 1, 0, n - 4): not-a-knot removes x_1 and x_n-2 as break points, so
    the first two and the last two intervals share a piece (scipy make_interp_spline: t = [x_0]*4 + x[2:-2] + [x_n-1]*4)."""
    from scipy.interpolate import BSpline, make_interp_spline

    x = np.asarray(x, dtype=float)
    n = x.size
    if n < 6:
        raise ValueError("the B-spline form of the AP stage needs at least six k points")
    spl = make_interp_spline(x, np.eye(n), k=3)          # coefficients [n, n]: column j = response to y = e_j
    C = np.asarray(spl.c)                                 # c = C @ y
    t = spl.t
    cband = np.zeros((2 * SPL_HB + 1, n))
    for d in range(2 * SPL_HB + 1):
        j = np.arange(n) + d - SPL_HB
        ok = (j >= 0) & (j < n)
        cband[d, ok] = C[np.arange(n)[ok], j[ok]]
    J = np.clip(np.arange(n) - 1, 0, n - 4)
    local = np.zeros((n, 4, 4))
    fact = (1.0, 1.0, 2.0, 6.0)
    for i in range(n - 1):
        xm = 0.5 * (x[i] + x[i + 1])                     # a point inside the interval: Taylor coefficients about x_i from there (exact for a cubic)
        for e in range(4):
            b = BSpline.basis_element(t[J[i] + e : J[i] + e + 5], extrapolate=True)
            d = [float(b(xm, nu)) for nu in range(4)]    # value and derivatives at xm
            h = x[i] - xm
            # shift the expansion point from xm to x_i
            a3 = d[3] / 6.0
            a2 = d[2] / 2.0 + 3.0 * a3 * h
            a1 = d[1] + d[2] * h + 3.0 * a3 * h * h
            a0 = d[0] + d[1] * h + d[2] / 2.0 * h * h + a3 * h**3
            local[i, e] = (a0, a1, a2, a3)
    local[n - 1] = local[n - 2]                           # (never addressed: the last interval is n - 2)
    return cband, local, J


def spline_matrix(x, xe):
    """Dense operator of the same spline, end pieces extrapolated (interp1d 'extrapolate')."""
    return CubicSpline(x, np.eye(len(x)), axis=0, extrapolate=True)(xe)


# ----------------------------------------------------------------------------- configuration
@dataclass
class EngineConfig:
    """What the hot path needs from ``Common`` + the plugin constructors
    (reference pybird.py:498-514, 907-915, 1230, 1503-1518; window.py:121-145; binning.py:43-53)."""

    Nl: int = 2
    k: Optional[np.ndarray] = None          # None -> native grid (pybird.py:472-479)
    kin: Optional[np.ndarray] = None        # None -> logspace(-5, 0, 200) (theory.py:562)
    NFFT: int = 256
    fft_window: Optional[float] = 0.2       # NonLinear.PsCf(window=0.2) (pybird.py:1143)
    irf_soffset: float = 1.0                # Resum.IRFilters(soffset, RescaleIR, window) (pybird.py:1316-1353); Resum.Ps uses the defaults
    irf_rescale: float = 1.0
    irf_window: Optional[float] = None
    with_resum: bool = False
    LambdaIR: float = 0.2
    NFFT_resum: int = 192
    resum_window: Optional[float] = None    # Resum.Ps(bird, window=...): coefficient taper of the resummation FFTLog (pybird.py:1409-1411)
    with_ap: bool = False
    DA_AP: Optional[float] = None
    H_AP: Optional[float] = None
    nbinsmu: int = 200
    APst: bool = False
    with_NNLO: bool = False                 # Common(with_NNLO=True): the k^4 P11 counter-terms (pybird.py:741-748)
    optiresum: bool = False                 # Common(optiresum=True): resum only the BAO peak (pybird.py:553-556, 1382-1400)
    IRcutoff: object = False                # Common(IRcutoff=False | True | "all" | "loop" | "resum", kIR) (pybird.py:528-533)
    kIR: Optional[float] = None
    extra: dict = field(default_factory=dict)


def legendre_table(Nl, mu):
    return np.array([legendre(2 * l)(mu) for l in range(Nl)])


PYEGG_KEYS = ("Pow", "M22", "M13", "Mcf11", "Mcf22", "Mcf13", "Mcfct", "McfctNNLO")


def pyegg_path(path, NFFT, Nl):
    """File name of the reference's loop-matrix cache (reference pybird.py:925)."""
    import os

    return os.path.join(path, f"pyegg{NFFT}_Nl{Nl}.npz")


def loop_matrices(Nl, NFFT=256, bias=-1.6, xmin=1.5e-5, xmax=1000.0):
    """The loop matrices in the reference's own cache layout (reference pybird.py:968-981):
    Pow[N+1], M22[28,N+1,N+1], M13[10,N+1], Mcf11/Mcfct/McfctNNLO[Nl,N+1], Mcf22[28,Nl,N+1,N+1], Mcf13[10,Nl,N+1,N+1]."""
    dx = np.log(xmax / xmin) / (NFFT - 1.0)
    Pow = bias + 1j * 2.0 * np.pi / (NFFT * dx) * (np.arange(NFFT + 1) - NFFT / 2.0)
    nu = -0.5 * Pow
    ells = 2 * np.arange(Nl)
    M22, M13 = lm.matrices_22(nu), lm.vectors_13(nu)
    Ml = lm.bessel_weight(ells[:, None, None], nu[None, :, None] + nu[None, None, :] - 1.5)
    return dict(Pow=Pow, M22=M22, M13=M13, Mcf11=lm.bessel_weight(ells[:, None], nu[None, :]),
                Mcf22=np.einsum("lnm,bnm->blnm", Ml, M22), Mcf13=np.einsum("lnm,bn->blnm", Ml, M13),
                Mcfct=lm.bessel_weight(ells[:, None], nu[None, :] - 1.0), McfctNNLO=lm.bessel_weight(ells[:, None], nu[None, :] - 2.0))


def build_tables(cfg: EngineConfig, loop_cache=None) -> dict:
    """-> {name: ndarray} of every constant table, float64 / int32, C-contiguous, device layout.
    ``loop_cache``: the arrays of a reference ``pyegg*.npz`` (M22, M13, Mcf11, Mcfct are used) instead of recomputing them."""
    if cfg.NFFT != 256:
        raise ValueError("the HIP engine is specialised for NFFT=256 (the reference default, pybird.py:912)")
    Nl = cfg.Nl
    if Nl not in (2, 3):
        raise ValueError("Nl must be 2 or 3")
    k = lm.native_k() if cfg.k is None else np.ascontiguousarray(cfg.k, dtype=np.float64)
    kin = np.logspace(-5, 0, 200) if cfg.kin is None else np.ascontiguousarray(cfg.kin, dtype=np.float64)
    s = lm.native_s()
    sr_idx = np.arange(NS)                      # the s slots that enter the resummation
    if cfg.optiresum:
        # the s axis keeps its 80 device slots: the first 52 carry the optiresum grid, the rest repeat the last value and
        # get zero weight everywhere they could matter (H columns, BAO mask)
        s_opt = np.arange(70.0, 200.0, 2.5)
        idlow, idhigh = int(np.where(s_opt > 70.0)[0][0]), int(np.where(s_opt > 190.0)[0][0])   # pybird.py:1236-1239
        s = np.concatenate([s_opt, np.full(NS - s_opt.size, s_opt[-1])])
        sr_idx = np.arange(idlow, idhigh)
        # Resum.extractBAO: linear interpolation of s^2 xi(s) between the last point below and the first point above the window
        ilo, ihi = idlow - 1, idhigh
        tt = (s[sr_idx] - s[ilo]) / (s[ihi] - s[ilo])
        bao = np.zeros(2 * NS + 4)
        bao[sr_idx] = (1.0 - tt) * s[ilo] ** 2 / s[sr_idx] ** 2
        bao[NS + sr_idx] = tt * s[ihi] ** 2 / s[sr_idx] ** 2
        bao[2 * NS :] = [ilo, ihi, idlow, idhigh]
    Nk, Nkin = k.size, kin.size
    t = dict(k=k, kin=kin, s=s)
    w = lm.mu_weights(Nl)
    t.update(w)
    grp22 = np.array([lm.GROUP_22[b] for b in range(28)], dtype=np.int32)
    grp13 = np.array([lm.GROUP_13[b] for b in range(10)], dtype=np.int32)
    t["grp22"], t["grp13"] = grp22, grp13

    # ---- P11 = Sk @ Pin
    t["Sk"] = np.ascontiguousarray(CubicSpline(kin, np.eye(Nkin), axis=0)(k))

    # ---- loop FFTLog as an operator on Pin; only the 129 independent coefficients are kept
    ircut = "all" if cfg.IRcutoff is True else cfg.IRcutoff
    if ircut not in (False, "all", "loop", "resum"):
        raise ValueError(f"unexpected IRcutoff option: {ircut}")
    if ircut and cfg.kIR is None:
        raise ValueError("kIR must be specified when doing IRcutoff")
    icut = int(np.searchsorted(kin, cfg.kIR)) if ircut else 0

    def cut_operator(N, xmin, xmax, bias, window):
        """The FFTLog operator of the samples at kin >= kIR, zero padded below (reference pybird.py:1136-1140, 1327-1335),
        as a matrix on the full input grid."""
        o = FFTLogOperator(N, xmin, xmax, bias, kin[icut:], window, extrap=("padding", "extrap"))
        o.G = np.hstack([np.zeros((o.G.shape[0], icut), dtype=complex), o.G])
        return o

    op = FFTLogOperator(256, 1.5e-5, 1000.0, -1.6, kin, cfg.fft_window)
    if op.low_active:
        raise ValueError("kin[0] must not exceed the FFTLog xmin=1.5e-5 (low-k extrapolation unsupported)")
    nh = NHALF + 1
    # coefficient sets of the k-space (P22, P13) and xi-space (C11, Cct, C22, C13) pieces (reference pybird.py:1151-1160)
    op_cut = cut_operator(256, 1.5e-5, 1000.0, -1.6, cfg.fft_window) if ircut else None
    op_k = op_cut if ircut in ("all", "loop") else op
    op_s = op_cut if ircut in ("all", "resum") else op
    t["Gc"] = np.ascontiguousarray(np.stack([op_k.G[:nh].real, op_k.G[:nh].imag]))        # [2,129,Nkin]
    if op_s is not op_k:
        t["Gc2"] = np.ascontiguousarray(np.stack([op_s.G[:nh].real, op_s.G[:nh].imag]))
    t["Ec"] = np.ascontiguousarray(np.stack([op.E_hi[:nh].real, op.E_hi[:nh].imag]))      # [2,129,Ntail]
    t["lnx_tail"] = op.lnx_hi
    Pow = op.Pow
    nu = -0.5 * Pow
    # ---- one-loop pieces through the anti-diagonal form (see antidiagonal_tables)
    if loop_cache is not None:
        if np.shape(loop_cache["Pow"]) != Pow.shape or np.any(np.asarray(loop_cache["Pow"]) != Pow):
            raise ValueError("loop-matrix cache was written for a different FFTLog configuration")
        M22, M13 = np.asarray(loop_cache["M22"]), np.asarray(loop_cache["M13"])
        if M22.shape != (28, NPOW, NPOW) or M13.shape != (10, NPOW):
            raise ValueError("loop-matrix cache has the wrong shapes")
    else:
        M22 = lm.matrices_22(nu)
        M13 = lm.vectors_13(nu)
    ells = 2 * np.arange(Nl)
    basis, comb = loop_basis(M22)
    t["basis22"], t["comb22"] = basis.astype(np.int32), comb                 # M22[b] = sum_c comb[b, c] M22[basis[c]]
    vecs = [M13]
    if cfg.with_resum:
        basis13, comb13 = loop_basis(M13)
        t["comb13"] = comb13                                                  # M13[b] = sum_c comb13[b, c] M13[basis13[c]]
        # expansion of the synthesised basis rows (l, c) -> C22[l, b], C13[l, b]: dense [Nl*38, Nl*(nb + nb13)]
        nb, nb13 = comb.shape[1], comb13.shape[1]
        expc = np.zeros((Nl * 38, Nl * (nb + nb13)))
        for l in range(Nl):
            expc[l * 28 : (l + 1) * 28, l * (nb + nb13) : l * (nb + nb13) + nb] = comb
            expc[Nl * 28 + l * 10 : Nl * 28 + (l + 1) * 10, l * (nb + nb13) + nb : (l + 1) * (nb + nb13)] = comb13
        t["expand_c"] = expc
        t["ad"] = antidiagonal_tables(M22[basis], M13[basis13])
        jp = np.arange(NPOW)
        t["mlj"] = lm.bessel_weight(ells[:, None], -op.bias - 0.5j * op.dpow * jp[None, :] - 1.5)   # Ml depends on n + m only
        if loop_cache is not None and np.shape(loop_cache["Mcf11"]) == (Nl, NPOW):
            vecs += [np.asarray(loop_cache["Mcf11"]), np.asarray(loop_cache["Mcfct"])]
        else:
            vecs += [lm.bessel_weight(ells[:, None], nu[None, :]), lm.bessel_weight(ells[:, None], nu[None, :] - 1.0)]
        if cfg.with_NNLO:  # McfctNNLO (reference pybird.py:1054-1056)
            if loop_cache is not None and np.shape(loop_cache.get("McfctNNLO")) == (Nl, NPOW):
                vecs.append(np.asarray(loop_cache["McfctNNLO"]))
            else:
                vecs.append(lm.bessel_weight(ells[:, None], nu[None, :] - 2.0))
        t["syn_s"] = synthesis_table(np.log(s), -2.0 * op.bias - 6.0, op.dpow, NPOW - 1, sign=-1.0)
        t["lin_s"] = synthesis_table(np.log(s), -op.bias - 3.0, op.dpow, NHALF, sign=+1.0)
    else:
        t["ad"] = antidiagonal_tables(M22[basis], None)
    t["linvec"] = np.ascontiguousarray(np.concatenate(vecs)[:, :nh])          # [10 (+ 2|3 Nl), 129] complex: M13, Mcf11, Mcfct (, McfctNNLO)
    if cfg.with_NNLO:
        t["lctn"] = np.concatenate([w["lctNNLO"], np.zeros((Nl, 3))], axis=1)  # the NNLO block keeps lctNNLO in the first 3 Pctl slots
    t["syn_k"] = synthesis_table(np.log(k), 3.0 + 2.0 * op.bias, op.dpow, NPOW - 1, sign=+1.0)
    t["lin_k"] = synthesis_table(np.log(k), 3.0 + op.bias, op.dpow, NHALF, sign=-1.0)

    # ---- IR-resummation
    if cfg.with_resum:
        NIR = 16 if Nl == 3 else 8
        Na = 3 if Nl == 3 else 2
        kr_mask = k >= 0.02
        kr = k[kr_mask]
        rop = FFTLogOperator(cfg.NFFT_resum, 0.1, 10000.0, -0.6, s[sr_idx], cfg.resum_window, extrap=("padding", "padding"))
        rM = np.stack([8.0 * np.pi**3 * lm.bessel_weight(2 * l, -0.5 * rop.Pow) for l in range(Na)])
        rk = np.exp(np.outer(-rop.Pow - 3.0, np.log(kr)))                # [193,Nkr]
        H = np.zeros((Na, Nk, NS))
        for v in range(Na):
            H[np.ix_([v], np.where(kr_mask)[0], sr_idx)] = np.real((rM[v][:, None] * rk).T @ rop.G)[None]
        t["H"] = H
        if cfg.optiresum:
            t["bao"] = bao
        # IR filters: q = Pin * exp(-k^2/L^2)/k^2 -> FFTLog(32) -> j0/j2 sums -> X, Y
        wq = np.exp(-(kin**2) / cfg.LambdaIR**2) / kin**2
        xop = (cut_operator(32, 1.5e-5, 10.0, -2.6, cfg.irf_window) if ircut in ("all", "resum")
               else FFTLogOperator(32, 1.5e-5, 10.0, -2.6, kin, cfg.irf_window))
        if xop.low_active:
            raise ValueError("kin[0] must not exceed 1.5e-5")
        XM = np.stack([lm.bessel_weight(2 * l, -0.5 * xop.Pow) for l in range(2)])
        XsPow = np.exp(np.outer(-xop.Pow - 3.0, np.log(s)))              # [33,Ns]
        K = lambda D: np.real(np.einsum("ln,ns,ni->lsi", XM, XsPow, D))   # [2,Ns,cols]
        soff = cfg.irf_soffset ** (-xop.Pow - 3.0)                       # X0offset: the j0 sum at s = soffset (pybird.py:1339-1346)
        Koff = lambda D: np.real(np.einsum("n,ni->i", XM[0] * soff, D))
        body, tail = K(xop.G), K(xop.E_hi)
        boff, toff = Koff(xop.G), Koff(xop.E_hi)
        t["BX"] = np.ascontiguousarray(cfg.irf_rescale * 2.0 / 3.0 * (boff[None, :] - body[0] - body[1]) * wq[None, :])
        t["BY"] = np.ascontiguousarray(2.0 * body[1] * wq[None, :])
        t["TX"] = np.ascontiguousarray(cfg.irf_rescale * 2.0 / 3.0 * (toff[None, :] - tail[0] - tail[1]))
        t["TY"] = np.ascontiguousarray(2.0 * tail[1])
        t["lnx_xtail"] = xop.lnx_hi
        t["wq_last2"] = wq[-2:].copy()
        t["Qpoly"] = lm.q_polynomials(Nl)
        t["resum_dims"] = np.array([NIR, Na, int(np.sum(~kr_mask))], dtype=np.int32)
        t.update(resum_mfma_tables(t["Qpoly"], NIR, Na))

    # ---- AP
    if cfg.with_ap:
        mu = np.linspace(0.0, 1.0, cfg.nbinsmu)
        wmu = np.full(cfg.nbinsmu, mu[1] - mu[0])
        wmu[0] = wmu[-1] = 0.5 * (mu[1] - mu[0])
        t["mu"], t["wmu"] = mu, wmu
        t["legmu"] = np.ascontiguousarray(((2 * ells + 1) / 2.0)[:, None] * legendre_table(Nl, mu))
        t["sp_band"] = spline_derivative_band(k)
        t["sp_cband"], t["sp_local"], _ = bspline_tables(k)
        t["ap_fid"] = np.array([cfg.DA_AP, cfg.H_AP], dtype=np.float64)
    return t


# ----------------------------------------------------------------------------- IR-resummation on the matrix cores
RS_ZS = 8.0     # the polynomials are evaluated in t = k^2 X / RS_ZS (must match csrc/eftb_kernels.hpp RS_ZS)
RS_NB = 8       # dimension of the span of all resummation polynomials
RS_ROWS = 80    # Nl = 2: row table length (two MFMA row tiles used)
RS3_ROWS = 64   # Nl = 3: four MFMA row tiles x 16


def resum_mfma_tables(Qpoly, NIR, Na):
    """Tables of resum_mfma_kernel (Nl = 3).

    Every polynomial  sum_p Q_a[l,l',(half,p,v)](f) z^p  of the IR-resummation (there are 108 per cosmology, degree 15)
    lies, for every f, in one fixed 8-dimensional space: q_p (p+1)! (-2)^(p+1) is a degree-7 polynomial in p.  With an
    orthonormal basis V8 of that space (in the scaled variable t = z / RS_ZS) the polynomials of all rows at 16 (k, s) points
    become one [rows x 8] x [8 x 16] matrix product on the FP64 matrix cores, A = Q . diag(RS_ZS^p) . V8^T.

    The kernel's operand is built PER s (resum_as_kernel, from the inputs alone): the weight of an (a, l, l') block is
        W = k^2 sum_v H_v(k,s) [ delta(v,l') X(s) D(half 0, v) + Y(s) D(half 1, v) ]
    (the X^(p+1) series only couples v = l'), and D is linear in the rows of A, so X(s), Y(s) are folded into the rows:
        A_s[row] = X(s) A[half 0, v = l'] delta(v,l') + Y(s) A[half 1, v]
    THREE rows (v = 0, 1, 2) per block instead of four: 18 blocks = 54 rows fit four row tiles (ten MFMAs per step became eight, and
    the lanes no longer form z H, y H products or select between roles).

    Row layout (tile tau, row i):  lane group jg = i % 4, slot = i // 4  -- the 4 values a lane holds of one v_mfma_f64_16x16x4
    result (rows (lane >> 4) + 4 q).
      tau < 3, jg < 3:  slots 0-2 -> (a = 1, l = jg, l' = tau), v = slot;   slot 3 -> (a = 0, l = 1, l' = tau), v = jg
      tau < 3, jg = 3:  slots 0-2 -> (a = 0, l = 0, l' = tau), v = slot;   slot 3 -> zero
      tau = 3, jg < 3:  slots 0-2 -> (a = 0, l = 2, l' = jg),  v = slot;   slot 3 -> zero;   jg = 3 -> zero

    -> rs_basis [8,16] (V8), rs_basis_scaled [8,16] (V8 diag(RS_ZS^p)), rs_rows int32[2 * 64]: per row the offset of the p = 0
       coefficient of its X part and of its Y part inside one cosmology's Q block [2,Nl,Nl,Nn] (stride Na per p; -1 = none)."""
    if NIR == 8:
        return _resum_mfma_tables_nl2(Qpoly, NIR, Na)
    Nl = 3
    NN = 2 * NIR * Na
    Qr = Qpoly.reshape(2, Nl, Nl, 2, NIR, Na, Qpoly.shape[-1])          # table, l, l', half, p, v, f-power
    scale = RS_ZS ** np.arange(NIR)
    M = Qr.transpose(0, 1, 2, 3, 5, 6, 4).reshape(-1, NIR) * scale[None, :]
    M = M[np.abs(M).max(axis=1) > 0]
    _, sv, Vt = np.linalg.svd(M, full_matrices=False)
    if sv[RS_NB] > 1e-13 * sv[0]:
        raise ValueError(f"resummation polynomials do not span {RS_NB} dimensions (sv ratio {sv[RS_NB] / sv[0]:.2e})")
    V8 = np.ascontiguousarray(Vt[:RS_NB])
    rows = np.full((2, RS3_ROWS), -1, dtype=np.int32)
    used = np.zeros((2, Nl, Nl, 2, Na), dtype=bool)                     # a (device convention), l, l', half, v
    for tau in range(4):
        for i in range(16):
            jg, slot = i % 4, i // 4
            blk = resum_block(tau, jg, slot)
            if blk is None:
                continue
            a, l, lp, v = blk
            base = ((a * Nl + l) * Nl + lp) * NN
            r = 16 * tau + i
            if v == lp:
                rows[0, r] = base + v
                used[a, l, lp, 0, v] = True
            rows[1, r] = base + NIR * Na + v
            used[a, l, lp, 1, v] = True
    # device Q[a] = table[1 - a] (reference pybird.py:1374-1376): every nonzero polynomial must have a row
    nz = np.abs(Qr).max(axis=(4, 6)) > 0                                 # table, l, l', half, v
    if np.any(nz[::-1] & ~used):
        raise ValueError("resummation table has entries outside the (v = l' | half 1) slot pattern")
    return dict(rs_basis=V8, rs_basis_scaled=np.ascontiguousarray(V8 * scale[None, :]), rs_rows=rows.reshape(-1))


def resum_block(tau, jg, slot):
    """(a, l, l', v) of row (tile tau, lane group jg, slot) of the Nl = 3 resummation operand, or None for a zero row."""
    if tau < 3 and jg < 3:
        return (1, jg, tau, slot) if slot < 3 else (0, 1, tau, jg)
    if tau < 3:
        return (0, 0, tau, slot) if slot < 3 else None
    if jg < 3 and slot < 3:
        return (0, 2, jg, slot)
    return None


def _resum_mfma_tables_nl2(Qpoly, NIR, Na):
    """Nl = 2 (NIR = 8, Na = 2): the polynomials have degree 7, so the monomials of t = z / RS_ZS are the basis (V8 = identity) and
    A = Q diag(RS_ZS^p).  32 rows = two tiles: tile tau <-> l' = tau; row i: lane group i % 4 <-> (a, l) = (1, 0), (1, 1), (0, 0), (0, 1);
    slot i // 4: 0 -> (v = l', half 0), 1, 2 -> (v = slot - 1, half 1), 3 -> empty (resum_mfma2_kernel)."""
    Nl = 2
    NN = 2 * NIR * Na
    Qr = Qpoly.reshape(2, Nl, Nl, 2, NIR, Na, Qpoly.shape[-1])
    V8 = np.zeros((RS_NB, 16))
    V8[np.arange(RS_NB), np.arange(RS_NB)] = 1.0
    scale = np.zeros(16)
    scale[:NIR] = RS_ZS ** np.arange(NIR)
    rows = np.full(RS_ROWS, -1, dtype=np.int32)
    used = np.zeros((2, Nl, Nl, 2, Na), dtype=bool)
    for tau in range(2):
        for i in range(16):
            jg, slot = i % 4, i // 4
            a, l = ((1, 0), (1, 1), (0, 0), (0, 1))[jg]
            if slot == 3:
                continue
            v, half = (tau, 0) if slot == 0 else (slot - 1, 1)
            rows[16 * tau + i] = ((a * Nl + l) * Nl + tau) * NN + half * NIR * Na + v
            used[a, l, tau, half, v] = True
    nz = np.abs(Qr).max(axis=(4, 6)) > 0
    if np.any(nz[::-1] & ~used):
        raise ValueError("resummation table has entries outside the (v = l' | half 1) slot pattern")
    rows = np.concatenate([rows, np.full(2 * RS3_ROWS - RS_ROWS, -1, dtype=np.int32)])  # one table size for both kernels
    return dict(rs_basis=V8, rs_basis_scaled=np.ascontiguousarray(V8 * scale[None, :]), rs_rows=rows)


# ----------------------------------------------------------------------------- projections after AP
def window_pgrid(kmax, accboost=1):
    """(reference window.py:27-33)"""
    return np.concatenate(
        [np.geomspace(1e-5, 0.015, 100 * accboost, endpoint=False), np.arange(0.015, kmax, 1e-3 / accboost)]
    )


_CALQ = np.array(
    [
        [[1, 0, 0, 0], [0, 1 / 5, 0, 0], [0, 0, 1 / 9, 0], [0, 0, 0, 1 / 13]],
        [[0, 1, 0, 0], [1, 2 / 7, 2 / 7, 0], [0, 2 / 7, 100 / 693, 25 / 143], [0, 0, 25 / 143, 14 / 143]],
        [[0, 0, 1, 0], [0, 18 / 35, 20 / 77, 45 / 143], [1, 20 / 77, 162 / 1001, 20 / 143], [0, 45 / 143, 20 / 143, 252 / 2431]],
        [[0, 0, 0, 1], [0, 0, 5 / 11, 14 / 55], [0, 5 / 11, 20 / 99, 28 / 187], [1, 14 / 55, 28 / 187, 400 / 3553]],
    ]
)  # (2a+1) (a l q; 0 0 0)^2, reference window.py:286-303


def window_tables(k, sw, Qq, Na, Nl, accboost=1, Nmax=4096, xmin_factor=1.0, xmax_factor=100.0, bias=-1.6,
                  window_param=1, pmax=None):
    """k-independent tables of the window precompute (reference window.py:262-346).  The reference takes, per (a, l, k),
    the FFTLog(Nmax) of Q_{al}(s) j_{2a}(k s) and sums the coefficients against p^{-Pow-3}; the FFT and the sum are both
    linear in the sampled integrand, so they collapse into one real table per l,
        T_l[i, p] = p^2 Re sum_m e^{-2 pi i (m - N/2) i / N} w_l[m] p^{-Pow_m - 3},
    and  W_{al}(k, p) = sum_i (-1)^a Qt_{al}[i] j_{2a}(k x_i) T_l[i, p]  with Qt = Q_{al}(x_i) e^{-bias i dx}: a dense
    (Nk x nx) @ (nx x Np) product whose left factor is generated from Bessel functions (eftb_window_wal on the device).
    Returns x [nx], Qt [Na, Nl, nx] (sign included), T [Nl, nx, Np], p [Np]."""
    k = np.asarray(k, dtype=float)
    Nq = Qq.shape[0]
    Qal = np.einsum("alq,qs->als", _CALQ[:Na, :Nl, :Nq], Qq)
    p = window_pgrid(float(k.max()) if pmax is None else pmax, accboost)
    N = Nmax
    xmin, xmax = sw[0] * xmin_factor, sw[-1] * xmax_factor
    dx = np.log(xmax / xmin) / (N - 1.0)
    i = np.arange(N)
    x = xmin * np.exp(i * dx)
    Pow = bias + 1j * 2.0 * np.pi / (N * dx) * (np.arange(N + 1) - N / 2.0)
    cw = xmin ** (-Pow) / float(N) * edge_window(N, window_param)
    lo, hi = int(np.searchsorted(x, sw[0])), int(np.searchsorted(x, sw[-1], side="right"))
    Qx = CubicSpline(sw, Qal, axis=-1, extrapolate=False)(x[lo:hi])          # [Na,Nl,nx]
    sign = np.real((-1j) ** (2 * np.arange(Na)))                               # (-i)^{2a}
    Qt = Qx * np.exp(-bias * i[lo:hi] * dx) * sign[:, None, None]
    pPow = np.exp(np.outer(-Pow - 3.0, np.log(p))) * p**2                      # [N+1,Np]
    alt = np.where(i[lo:hi] % 2 == 0, 1.0, -1.0)[:, None]
    T = np.empty((Nl, hi - lo, p.size))
    for l in range(Nl):
        a = (cw * (1j) ** (2 * l) * 4.0 * np.pi * lm.bessel_weight(2 * l, -0.5 * Pow))[:, None] * pPow
        T[l] = alt * np.real(np.fft.fft(a[:N], axis=0)[lo:hi] + a[N])
    return x[lo:hi].copy(), np.ascontiguousarray(Qt), T, p


def window_matrix(k, sw, Qq, Na, Nl, **kw):
    """W_{al}(k, p) [Na, Nl, Nk, Np] and the p grid from a configuration-space window Q_q(s) (reference window.py:262-346)
    on the host, through window_tables: the table-building restatement the device version (window.window_matrix_device)
    is checked against."""
    x, Qt, T, p = window_tables(k, sw, Qq, Na, Nl, **kw)
    k = np.asarray(k, dtype=float)
    Wal = np.empty((Na, Nl, k.size, p.size))
    for a in range(Na):
        jk = spherical_jn(2 * a, x[None, :] * k[:, None])                    # [Nk,nx]
        for l in range(Nl):
            Wal[a, l] = (jk * Qt[a, l]) @ T[l]
    return Wal, p


def window_fold(k, Wal, p, windowk=0.05, withmask=True):
    """Mask + dp weights (reference window.py:348-359) folded with the k->p cubic spline
    (window.py:376-383):  P'_a(k) = sum_{l,k'} Wfold[a,l,k,k'] P_l(k')."""
    W = Wal
    if withmask:
        pp, kk = np.meshgrid(p, k, indexing="ij")
        mask = (pp < kk + windowk) & (pp > kk - windowk)
        W = np.einsum("alkp,pk->alkp", Wal, mask)
    dp = np.concatenate([[0.0], np.diff(p)])
    Waldk = W * dp
    return np.ascontiguousarray(np.einsum("alkp,pq->alkq", Waldk, spline_matrix(k, p), optimize=True)), Waldk


def binning_operator(k, kout=None, accboost=1, decimals=2, kstart=None, kend=None, nbins=None):
    """k^2-weighted bin average as a matrix [nbins, Nk] (+ keff, bin edges): cubic spline of the template
    onto 100*accboost points per bin, trapezoid rule, divided by the bin volume (reference binning.py:43-144)."""
    k = np.asarray(k, dtype=float)
    kout = np.asarray(kout, dtype=float)
    if kstart is None and kend is None and nbins is None:
        dk = np.round(kout[-1] - kout[-2], decimals)
        centre = (kout[-1] - dk * np.arange(len(kout)))[::-1]
        binmin, binmax = centre - dk / 2, centre + dk / 2
    else:
        if kstart is None or kend is None or nbins is None:
            raise ValueError("need specify kstart, kend and nbins together")
        edges = np.linspace(kstart, kend, nbins + 1)
        il = np.searchsorted(edges, kout[0]) - 1
        ir = np.searchsorted(edges, kout[-1], side="right") + 1
        edges = edges[il:ir]
        binmin, binmax = edges[:-1], edges[1:]
    binvol = (binmax**3 - binmin**3) / 3.0
    keff = (binmax**4 - binmin**4) / 4.0 / binvol
    npts = 100 * accboost
    op = np.empty((len(binmin), k.size))
    for b, (lo, hi) in enumerate(zip(binmin, binmax)):
        pts = np.linspace(lo, hi, npts)
        w = np.full(npts, pts[1] - pts[0])
        w[0] = w[-1] = 0.5 * (pts[1] - pts[0])
        op[b] = (w * pts**2) @ spline_matrix(k, pts) / binvol[b]
    return op, keff, binmin, binmax


def interp_operator(k, kout):
    """PlkInterpolator as a matrix [len(kout), Nk] (reference theory.py:75-106): cubic (not-a-knot) interpolation of
    k P(k) on the grid (0, k_0, ..., k_{N-1}) with the value 0 inserted at k = 0, evaluated at kout (end pieces
    extrapolated) and divided by kout."""
    k = np.asarray(k, dtype=float)
    kout = np.atleast_1d(np.asarray(kout, dtype=float))
    kg = np.concatenate([[0.0], k])
    S = spline_matrix(kg, kout)[:, 1:]  # the inserted node carries the value 0
    return np.ascontiguousarray(S * k[None, :] / kout[:, None])


def chained_matrix(Nl):
    """Q_l = P_l - A_l P_{l+2}, A_l = (2l+1) L_l(0) / ((2l+5) L_{l+2}(0)) (reference chained.py:13-54)."""
    m = np.zeros((Nl - 1, Nl))
    for a in range(Nl - 1):
        l = 2 * a
        m[a, a] = 1.0
        m[a, a + 1] = -((2 * l + 1) * legendre(l)(0)) / ((2 * l + 5) * legendre(l + 2)(0))
    return m


def linear_interp_matrix(x, xnew):
    """scipy interp1d(kind='linear', fill_value='extrapolate') as a matrix [len(xnew), len(x)] (end segments extended)."""
    x, xnew = np.asarray(x, dtype=float), np.asarray(xnew, dtype=float)
    lo = np.clip(np.searchsorted(x, xnew) - 1, 0, x.size - 2)
    t = (xnew - x[lo]) / (x[lo + 1] - x[lo])
    L = np.zeros((xnew.size, x.size))
    rows = np.arange(xnew.size)
    L[rows, lo] = 1.0 - t
    L[rows, lo + 1] += t
    return L


def _fiber_H(l, lp, x):
    """Legendre-overlap polynomials of the effective-window method, l > lp (reference pybird.py:49-65)"""
    if (l, lp) == (2, 0):
        return x**2 - 1.0
    if (l, lp) == (4, 0):
        return 1.75 * x**4 - 2.5 * x**2 + 0.75
    if (l, lp) == (4, 2):
        return x**4 - x**2
    return np.zeros_like(x)


def fiber_operator(k, Nl, fs, Dfc, ktrust=0.25, kout=None, nq=1024):
    """The correlated fibre-collision correction as a matrix (reference pybird.py:1703-1757, FiberCollision.dPcorr):
        dP_l(kout_i) = sum_{l', k'} F[l, l', i, k'] P_l'(k')
    The reference interpolates P linearly onto 1024 log-spaced q in [k_min, ktrust] and sums q dq P(q) f_ll'(k, q) over q < k
    (l' <= l) and k < q < ktrust (l' >= l); both steps are linear in P, so they fold into F = W @ L."""
    from scipy.special import j1

    k = np.asarray(k, dtype=float)
    kout = k if kout is None else np.asarray(kout, dtype=float)
    q = np.geomspace(k.min(), ktrust, num=nq)
    dq = np.concatenate([[0.0], np.diff(q)])
    Lq = linear_interp_matrix(k, q)
    w2d = 2.0 * j1(q * Dfc) / (q * Dfc)
    ir = q[None, :] < kout[:, None]
    uv = (q[None, :] > kout[:, None]) & (q[None, :] < ktrust)
    xi = q[None, :] / kout[:, None]      # q/k (IR side)
    xu = kout[:, None] / q[None, :]      # k/q (UV side)
    F = np.zeros((Nl, Nl, kout.size, k.size))
    for a in range(Nl):
        for b in range(Nl):
            l, lp = 2 * a, 2 * b
            W = np.zeros((kout.size, nq))
            if l == lp:
                fir, fuv = xi * w2d * xi**l, w2d * xu**l
            else:
                pre = (2.0 * l + 1.0) / 2.0
                fir = xi * w2d * pre * _fiber_H(max(l, lp), min(l, lp), xi)
                fuv = w2d * pre * _fiber_H(max(l, lp), min(l, lp), xu)
            if lp <= l:
                W += np.where(ir, fir, 0.0)
            if lp >= l:
                W += np.where(uv, fuv, 0.0)
            F[a, b] = (-0.5 * fs * Dfc**2 * W * (q * dq)[None, :]) @ Lq
    return F


def compose_operator(Nl, Nk, Wfold=None, binning=None, chained=False, fiber=None):
    """Fold window [Na,Nl,Nk,Nk], fibre collisions (F of ``fiber_operator``: P += F P), binning [nb,Nk] and the chained
    matrix into one [nl_out, Nl, nx_out, Nk], in the reference's order (theory.py:583-600)."""
    op = Wfold if Wfold is not None else np.einsum("al,xk->alxk", np.eye(Nl), np.eye(Nk))
    if fiber is not None:
        op = op + np.einsum("abxy,blyk->alxk", fiber, op)
    if binning is not None:
        op = np.einsum("bx,alxk->albk", binning, op)
    if chained:
        op = np.einsum("ca,alxk->clxk", chained_matrix(op.shape[0]), op)
    return np.ascontiguousarray(op)
