"""Oracle: the PyBird theory-vector pipeline on the CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

P_lin -> FFTLog -> P22/P13/C11/Cct/C22/C13 -> multipole regrouping -> IR-resummation -> AP ->
window -> binning -> chained -> bias contraction.  The timed stages use the same NumPy/SciPy calls
as the reference (same einsum subscripts and ``optimize='optimal'`` paths, ``rfft``, scipy cubic
``interp1d``, trapezoid rule) so that this file is a fair stand-in for "CPU eftpipe/pybird" on the
GPU box, where the reference itself is not available.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
from scipy.integrate import quad
from scipy.interpolate import interp1d
from scipy.special import legendre, spherical_jn

from . import tables as T
from .fftlog import FFTLogGrid

_trapz = getattr(np, "trapezoid", None) or np.trapz


# ----------------------------------------------------------------------------- helpers
def hubble(Om, z):
    """E(z) of flat LCDM (reference pybird.py:34-36)."""
    return (Om * (1 + z) ** 3.0 + (1 - Om)) ** 0.5


def da_func(Om, z):
    """Dimensionless angular-diameter distance (reference pybird.py:39-42)."""
    return quad(lambda x: 1.0 / hubble(Om, x), 0, z)[0] / (1 + z)


def survey_kgrid(Nk):
    """Non-native k grid of SURVEY.md 8(c): 7 native low-k points + linspace(0.02, 0.3, Nk-7)."""
    low = np.array([0.001, 0.005, 0.0075, 0.01, 0.0125, 0.015, 0.0175])
    return np.concatenate([low, np.linspace(0.02, 0.3, Nk - 7)])


def window_pgrid(kmax=0.3, accboost=1):
    """Integration grid of the window convolution (reference window.py:27-33)."""
    return np.concatenate(
        [np.geomspace(1e-5, 0.015, 100 * accboost, endpoint=False), np.arange(0.015, kmax, 1e-3 / accboost)]
    )


def cubic_to(k, P, x):
    """scipy cubic interp1d with end-piece extrapolation (pybird.py:1586-1593, window.py:376-383)."""
    return interp1d(k, P, axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")(x)


@dataclass
class OracleConfig:
    """Subset of the tracer_config keys that reach the hot path (reference theory.py:421-438)."""

    Nl: int = 2
    No: Optional[int] = None
    k: Optional[np.ndarray] = None  # None -> native 50-point grid (pybird.py:472-479)
    kmA: float = 0.7
    krA: float = 0.25
    ndA: float = 3e-4
    kmB: Optional[float] = None
    krB: Optional[float] = None
    ndB: Optional[float] = None
    NFFT: int = 256
    with_resum: bool = False
    LambdaIR: float = 0.2
    NFFT_resum: int = 192
    resum_window: Optional[float] = None  # Resum.Ps(bird, window=...) (pybird.py:1409-1411, 1413)
    with_ap: bool = False
    Om_AP: Optional[float] = None
    z_AP: Optional[float] = None
    DA_AP: Optional[float] = None
    H_AP: Optional[float] = None
    nbinsmu: int = 200
    APst: bool = False
    with_NNLO: bool = False  # Common(with_NNLO=True) (pybird.py:511, 741-748)
    optiresum: bool = False  # Common(optiresum=True): resum only the BAO peak (pybird.py:553-556, 1235-1244, 1382-1400)
    IRcutoff: object = False  # False | True (= "all") | "all" | "loop" | "resum"  (pybird.py:528-533)
    kIR: Optional[float] = None
    # window
    window_file: Optional[str] = None  # config-space window "s Q0 Q2 ..."
    window_accboost: int = 1
    windowk: float = 0.05
    window_Nmax: int = 4096
    window_st: bool = True
    # binning
    kout: Optional[np.ndarray] = None
    binning_accboost: int = 1
    extra: dict = field(default_factory=dict)


class OracleEngine:
    """All init-time tables + the per-evaluation stages, one method per reference stage."""

    # ------------------------------------------------------------------ init (Common, NonLinear)
    def __init__(self, cfg: OracleConfig):
        self.cfg = cfg
        Nl = self.Nl = cfg.Nl
        self.No = cfg.No or Nl
        pt = T.pt_tables()
        # Common (pybird.py:498-582)
        self.k = np.array(pt["kbird"]) if cfg.k is None else np.asarray(cfg.k, dtype=float)
        self.Nk = self.k.size
        self.s = np.arange(70.0, 200.0, 2.5) if cfg.optiresum else np.array(pt["sbird"])  # pybird.py:553-556
        self.Ns = self.s.size
        self.kr = self.k[0.02 <= self.k]
        self.Nkr = self.kr.size
        self.Nklow = self.Nk - self.Nkr
        self.kmA, self.krA, self.ndA = cfg.kmA, cfg.krA, cfg.ndA
        self.kmB = cfg.kmA if cfg.kmB is None else cfg.kmB
        self.krB = cfg.krA if cfg.krB is None else cfg.krB
        self.ndB = cfg.ndA if cfg.ndB is None else cfg.ndB
        w = T.mu_weights(Nl)
        self.l11, self.lct, self.l22, self.l13 = w["l11"], w["lct"], w["l22"], w["l13"]
        self.lctNNLO = w["lctNNLO"]
        self._init_loops()
        if cfg.with_resum:
            self._init_resum()
        if cfg.with_ap:
            self._init_ap()
        if cfg.window_file is not None:
            self._init_window()
        if cfg.kout is not None:
            self._init_binning()

    def _init_loops(self):
        """Loop matrices and power tables (reference pybird.py:907-1064)."""
        cfg, Nl = self.cfg, self.Nl
        self.fft = FFTLogGrid(cfg.NFFT, 1.5e-5, 1000.0, -1.6)
        nu = -0.5 * self.fft.Pow
        ma = T.m22a(nu[:, None], nu[None, :])
        self.M22 = np.stack([ma * T.m22b(b, nu[:, None], nu[None, :]) for b in range(28)])
        m13 = T.m13a(nu)
        self.M13 = np.stack([m13 * T.m13b(b, nu) for b in range(10)])
        ells = 2 * np.arange(Nl)
        self.Mcf11 = T.mpc(ells[:, None], nu[None, :])
        self.Ml = T.mpc(ells[:, None, None], nu[None, :, None] + nu[None, None, :] - 1.5)
        self.Mcfct = T.mpc(ells[:, None], nu - 1.0)
        self.McfctNNLO = T.mpc(ells[:, None], nu - 2.0)  # pybird.py:1054-1056
        self.Mcf22 = np.einsum("lnm,bnm->blnm", self.Ml, self.M22)
        self.Mcf13 = np.einsum("lnm,bn->blnm", self.Ml, self.M13)
        self.kPow = np.exp(np.einsum("n,k->nk", self.fft.Pow, np.log(self.k)))
        self.sPow = np.exp(np.einsum("n,s->ns", -self.fft.Pow - 3.0, np.log(self.s)))
        ep = lambda sub, *ops: np.einsum_path(sub, *ops, optimize="optimal")[0]
        self.path_P22 = ep("nk,mk,bnm->bk", self.kPow, self.kPow, self.M22)
        self.path_P22_pairwise = ["einsum_path", (0, 2), (0, 1)]
        self.path_P13 = ep("nk,bn->bk", self.kPow, self.M13)
        self.path_C11 = ep("ns,ln->ls", self.sPow, self.Mcf11)
        self.path_C22 = ep("ns,ms,blnm->lbs", self.sPow, self.sPow, self.Mcf22)
        self.path_C13 = ep("ns,ms,blnm->lbs", self.sPow, self.sPow, self.Mcf13)

    # ------------------------------------------------------------------ Bird + NonLinear.PsCf
    def linear(self, kin, Pin):
        """P11 on the engine k grid (reference pybird.py:694-695)."""
        return interp1d(kin, Pin, kind="cubic")(self.k)

    def loop_coef(self, kin, Pin, window=0.2, IRcut=False):
        """FFTLog of P_lin (reference pybird.py:1127-1141, 1143); IRcut drops the samples below kIR and pads with zeros."""
        extrap = ("extrap", "extrap")
        if IRcut:
            idx = np.searchsorted(kin, self.cfg.kIR)
            kin, Pin = kin[idx:], Pin[idx:]
            extrap = ("padding", "extrap")
        return self.fft.coef(kin, Pin, extrap=extrap, window=window)

    def _ircutoff(self):
        mode = self.cfg.IRcutoff
        if mode and self.cfg.kIR is None:
            raise ValueError("kIR must be specified when doing IRcutoff")
        return "all" if mode is True else mode

    def pscf(self, kin, Pin, pairwise=False):
        """One-loop P and xi pieces (reference pybird.py:1066-1125, 1143-1171)."""
        P11 = self.linear(kin, Pin)
        mode = self._ircutoff()  # which of the two coefficient sets is cut (pybird.py:1151-1160)
        if mode in ("all", False):
            coef = coef_cf = self.loop_coef(kin, Pin, IRcut=bool(mode))
        elif mode == "loop":
            coef, coef_cf = self.loop_coef(kin, Pin, IRcut=True), self.loop_coef(kin, Pin)
        elif mode == "resum":
            coef, coef_cf = self.loop_coef(kin, Pin), self.loop_coef(kin, Pin, IRcut=True)
        else:
            raise ValueError(f"unexpected IRcutoff option: {mode}")
        ck = coef[:, None] * self.kPow
        cs = coef_cf[:, None] * self.sPow
        path22 = self.path_P22_pairwise if pairwise else self.path_P22
        out = dict(P11=P11, coef=coef)
        out["P22"] = self.k**3 * np.real(np.einsum("nk,mk,bnm->bk", ck, ck, self.M22, optimize=path22))
        out["P13"] = self.k**3 * P11 * np.real(np.einsum("nk,bn->bk", ck, self.M13, optimize=self.path_P13))
        out["C11"] = np.real(np.einsum("ns,ln->ls", cs, self.Mcf11, optimize=self.path_C11))
        out["Cct"] = self.s**-2 * np.real(np.einsum("ns,ln->ls", cs, self.Mcfct, optimize=self.path_C11))
        if self.cfg.with_NNLO:  # makeCctNNLO (pybird.py:1098-1101)
            out["CctNNLO"] = self.s**-4 * np.real(np.einsum("ns,ln->ls", cs, self.McfctNNLO, optimize=self.path_C11))
        out["C22"] = np.real(np.einsum("ns,ms,blnm->lbs", cs, cs, self.Mcf22, optimize=self.path_C22))
        out["C13"] = np.real(np.einsum("ns,ms,blnm->lbs", cs, cs, self.Mcf13, optimize=self.path_C13))
        return out

    # ------------------------------------------------------------------ Bird.setPsCfl
    @staticmethod
    def _regroup(f, T22, T13):
        """28 + 10 loop pieces -> 12 bias groups (reference pybird.py:758-846)."""
        out = np.empty(T22.shape[:1] + (12,) + T22.shape[2:])
        out[:, 0] = (
            f**2 * T22[:, 20] + f**3 * T22[:, 23] + f**3 * T22[:, 24] + f**4 * T22[:, 25]
            + f**4 * T22[:, 26] + f**4 * T22[:, 27] + f**2 * T13[:, 7] + f**3 * T13[:, 8] + f**3 * T13[:, 9]
        )
        out[:, 1] = (
            f * T22[:, 9] + f**2 * T22[:, 14] + f**2 * T22[:, 15] + f**3 * T22[:, 21] + f**3 * T22[:, 22]
            + f * T13[:, 3] + f**2 * T13[:, 5] + f**2 * T13[:, 6]
        )
        out[:, 2] = f * T22[:, 10] + f**2 * T22[:, 16] + f**2 * T22[:, 17]
        out[:, 3] = f * T13[:, 4]
        out[:, 4] = f * T22[:, 11] + f**2 * T22[:, 18] + f**2 * T22[:, 19]
        out[:, 5] = T22[:, 0] + f * T22[:, 6] + f**2 * T22[:, 12] + f**2 * T22[:, 13] + T13[:, 0] + f * T13[:, 2]
        out[:, 6] = T22[:, 1] + f * T22[:, 7]
        out[:, 7] = T13[:, 1]
        out[:, 8] = T22[:, 2] + f * T22[:, 8]
        out[:, 9] = T22[:, 3]
        out[:, 10] = T22[:, 4]
        out[:, 11] = T22[:, 5]
        return out

    def set_pscfl(self, f, st):
        """Multipole weights, regrouping, shot-noise subtraction, stochastic templates
        (reference pybird.py:737-756, 850-866)."""
        k2 = self.k**2
        P11l = np.einsum("x,ln->lnx", st["P11"], self.l11)
        Pctl = np.einsum("x,x,ln->lnx", k2, st["P11"], self.lct)
        P22l = np.einsum("nx,ln->lnx", st["P22"], self.l22)
        P13l = np.einsum("nx,ln->lnx", st["P13"], self.l13)
        C22l = np.einsum("lnx,ln->lnx", st["C22"], self.l22)
        C13l = np.einsum("lnx,ln->lnx", st["C13"], self.l13)
        Ploopl = self._regroup(f, P22l, P13l)
        Cloopl = self._regroup(f, C22l, C13l)
        Ploopl = Ploopl - Ploopl[:, :, :1]
        Pstl = np.zeros((self.Nl, 3, self.Nk))
        Pstl[0, 0] = 1.0
        Pstl[0, 1] = k2
        if self.Nl >= 2:
            Pstl[1, 2] = k2
        out = dict(P11l=P11l, Pctl=Pctl, Ploopl=Ploopl, Cloopl=Cloopl, Pstl=Pstl, Picc=np.zeros((self.Nl, self.Nk)))
        if self.cfg.with_NNLO:  # pybird.py:741-748
            out["PctNNLOl"] = np.einsum("x,x,ln->lnx", self.k**4, st["P11"], self.lctNNLO)
        return out

    # ------------------------------------------------------------------ Resum
    def _init_resum(self):
        """IR-resummation tables (reference pybird.py:1230-1314, 1355-1359)."""
        cfg = self.cfg
        self.NIR = 16 if self.Nl == 3 else 8
        self.Na = 3 if self.NIR == 16 else 2
        self.Nn = self.NIR * self.Na * 2
        k2pi = np.array([self.kr ** (2 * (p + 1)) for p in range(self.NIR)])
        self.k2p = np.concatenate((k2pi, k2pi))
        self.rfft = FFTLogGrid(cfg.NFFT_resum, 0.1, 10000.0, -0.6)
        self.rM = np.stack([8.0 * np.pi**3 * T.mpc(2 * l, -0.5 * self.rfft.Pow) for l in range(self.Nl)])
        self.rkPow = np.exp(np.einsum("n,s->ns", -self.rfft.Pow - 3.0, np.log(self.kr)))
        self.xfft = FFTLogGrid(32, 1.5e-5, 10.0, -2.6)
        self.XM = np.stack([T.mpc(2 * l, -0.5 * self.xfft.Pow) for l in range(2)])
        if cfg.optiresum:  # pybird.py:1235-1244
            self.idlow = np.where(self.s > 70.0)[0][0]
            self.idhigh = np.where(self.s > 190.0)[0][0]
            self.sbao = self.s[self.idlow : self.idhigh]
            self.snobao = np.concatenate([self.s[: self.idlow], self.s[self.idhigh :]])
            self.sr = self.sbao
        else:
            self.sr = self.s
        self.XsPow = np.exp(np.einsum("n,s->ns", -self.xfft.Pow - 3.0, np.log(self.sr)))

    def extract_bao(self, cf):
        """(pybird.py:1382-1400): subtract the smooth part interpolated (linearly in s^2 xi) across the BAO window"""
        if not self.cfg.optiresum:
            return cf
        nobao_in = np.concatenate([cf[..., : self.idlow], cf[..., self.idhigh :]], axis=-1)
        nobao = interp1d(self.snobao, self.snobao**2 * nobao_in, kind="linear", axis=-1)(self.sbao) * self.sbao**-2
        return cf[..., self.idlow : self.idhigh] - nobao

    def ir_filters(self, kin, Pin):
        """X(s), Y(s) (reference pybird.py:1316-1353)."""
        L = self.cfg.LambdaIR
        if self._ircutoff() in ("loop", False):
            coef = self.xfft.coef(kin, Pin * np.exp(-(kin**2) / L**2) / kin**2, window=None)
        else:  # pybird.py:1327-1335
            idx = np.searchsorted(kin, self.cfg.kIR)
            kc, Pc = kin[idx:], Pin[idx:]
            coef = self.xfft.coef(kc, Pc * np.exp(-(kc**2) / L**2) / kc**2, window=None, extrap=("padding", "extrap"))
        cs = np.einsum("n,ns->ns", coef, self.XsPow)
        X02 = np.real(np.einsum("ns,ln->ls", cs, self.XM))
        X0off = np.real(np.einsum("n,n->", np.einsum("n,n->n", coef, 1.0 ** (-self.xfft.Pow - 3.0)), self.XM[0]))
        X02[0] = X0off - X02[0]
        X = 2.0 / 3.0 * (X02[0] - X02[1])
        Y = 2.0 * X02[1]
        return X, Y

    def _ir_block(self, XpYp, C):
        """FFTLog(192) of XpYp (x) C followed by the Bessel sum (pybird.py:1361-1365, 1409-1441)."""
        inp = np.einsum("jk,...k->...jk", XpYp, C)
        coef = self.rfft.coef(self.sr, inp, extrap="padding", window=self.cfg.resum_window)
        out = np.zeros(C.shape[:-1] + (self.Nn, self.Nk))
        flat_c = coef.reshape(-1, 2 * self.NIR, coef.shape[-1])
        flat_o = out.reshape(-1, self.Nn, self.Nk)
        for r in range(flat_c.shape[0]):
            for j in range(2 * self.NIR):
                ir = self.k2p[j] * np.real(self.rM[: self.Na] @ (flat_c[r, j][:, None] * self.rkPow))
                for v in range(self.Na):
                    flat_o[r, j * self.Na + v, self.Nklow :] = ir[v]
        return out

    def resum(self, f, kin, Pin, st):
        """IR-resummation of P11l, Pctl, Ploopl (reference pybird.py:1413-1464)."""
        Q = T.q_matrix(f, self.Nl)
        X, Y = self.ir_filters(kin, Pin)
        Xp = np.array([X ** (p + 1) for p in range(self.NIR)])
        XpY = np.array([Y * X**p for p in range(self.NIR)])
        XpYp = np.concatenate((Xp, XpY))
        IR11 = self._ir_block(XpYp, self.extract_bao(st["C11"]))
        IRct = self._ir_block(XpYp, self.extract_bao(st["Cct"]))
        IRloop = self._ir_block(XpYp, self.extract_bao(st["Cloopl"]))
        out = dict(st)
        out["P11l"] = st["P11l"] + np.einsum("lpn,pnk,pi->lik", Q[0], IR11, self.l11)
        out["Pctl"] = st["Pctl"] + np.einsum("lpn,pnk,pi->lik", Q[1], IRct, self.lct)
        out["Ploopl"] = st["Ploopl"] + np.einsum("lpn,pink->lik", Q[1], IRloop)
        if self.cfg.with_NNLO:  # pybird.py:1447-1458
            IRn = self._ir_block(XpYp, self.extract_bao(st["CctNNLO"]))
            out["PctNNLOl"] = st["PctNNLOl"] + np.einsum("lpn,pnk,pi->lik", Q[1], IRn, self.lctNNLO)
        out.update(X=X, Y=Y, Q=Q)
        return out

    # ------------------------------------------------------------------ AP
    def _init_ap(self):
        """Fiducial distances and mu quadrature (reference pybird.py:1503-1552)."""
        cfg = self.cfg
        if cfg.DA_AP is not None and cfg.H_AP is not None:
            self.DA_fid, self.H_fid = cfg.DA_AP, cfg.H_AP
        elif cfg.Om_AP is not None and cfg.z_AP is not None:
            self.DA_fid, self.H_fid = da_func(cfg.Om_AP, cfg.z_AP), hubble(cfg.Om_AP, cfg.z_AP)
        else:
            raise ValueError("expect input params: Om_AP and z_AP, or DA and H")
        self.muacc = np.linspace(0, 1, cfg.nbinsmu)
        self.kgrid, self.mugrid = np.meshgrid(self.k, self.muacc, indexing="ij")
        self.leg_mu = np.array([(2 * l + 1) / 2.0 * legendre(l)(self.mugrid) for l in 2 * np.arange(self.Nl)])

    def _integr_ap(self, Pk, kp, leg_mup):
        """(reference pybird.py:1581-1596)"""
        Pkint = cubic_to(self.k, Pk, kp)
        Pkmu = np.einsum("lpkm,lkm->pkm", Pkint, leg_mup, optimize=True)
        integrand = np.einsum("pkm,lkm->lpkm", Pkmu, self.leg_mu, optimize=True)
        return 2 * _trapz(integrand, x=self.mugrid, axis=-1)

    def ap(self, DA, H, st):
        """Alcock-Paczynski distortion (reference pybird.py:1554-1562, 1598-1621)."""
        qperp, qpar = DA / self.DA_fid, self.H_fid / H
        F = qpar / qperp
        root = (1 + self.mugrid**2 * (F**-2 - 1)) ** 0.5
        kp = self.kgrid / qperp * root
        mup = self.mugrid / F / root
        leg_mup = np.array([legendre(2 * i)(mup) for i in range(self.Nl)])
        c = 1.0 / (qperp**2 * qpar)
        out = dict(st)
        for name in ("P11l", "Pctl", "Ploopl") + (("Pstl",) if self.cfg.APst else ()) + (("PctNNLOl",) if self.cfg.with_NNLO else ()):
            out[name] = c * self._integr_ap(st[name], kp, leg_mup)
        return out

    # ------------------------------------------------------------------ window
    def _init_window(self):
        """W_{a l}(k, p) from the config-space window (reference window.py:262-359)."""
        cfg, Nl = self.cfg, self.Nl
        Na, Nq = Nl, 3
        tab = np.load(cfg.window_file) if str(cfg.window_file).endswith(".npy") else np.loadtxt(cfg.window_file)
        while tab[0, 0] == 0.0:
            tab = tab[1:]
        tab = tab[:, : 1 + Nq]
        # (2a+1) * Wigner-3j(a, l, q; 0 0 0)^2 coupling coefficients (window.py:286-303)
        Calq = np.array(
            [
                [[1, 0, 0, 0], [0, 1 / 5, 0, 0], [0, 0, 1 / 9, 0], [0, 0, 0, 1 / 13]],
                [[0, 1, 0, 0], [1, 2 / 7, 2 / 7, 0], [0, 2 / 7, 100 / 693, 25 / 143], [0, 0, 25 / 143, 14 / 143]],
                [[0, 0, 1, 0], [0, 18 / 35, 20 / 77, 45 / 143], [1, 20 / 77, 162 / 1001, 20 / 143], [0, 45 / 143, 20 / 143, 252 / 2431]],
                [[0, 0, 0, 1], [0, 0, 5 / 11, 14 / 55], [0, 5 / 11, 20 / 99, 28 / 187], [1, 14 / 55, 28 / 187, 400 / 3553]],
            ]
        )[..., :Nq]
        sw, Qq = tab[:, 0], tab[:, 1:].T
        Qal = np.einsum("alq,qs->als", Calq, Qq)[:Na, :Nl]
        self.p = window_pgrid(float(self.k.max()), cfg.window_accboost)
        wfft = FFTLogGrid(cfg.window_Nmax, sw[0], sw[-1] * 100.0, -1.6)
        pPow = np.exp(np.einsum("n,p->np", -wfft.Pow - 3.0, np.log(self.p)))
        M = np.stack([4 * np.pi * T.mpc(2 * l, -0.5 * wfft.Pow) for l in range(Nl)])
        Wal = np.empty((Na, Nl, self.Nk, self.p.size))
        for a in range(Na):
            kern = lambda x, a=a: spherical_jn(2 * a, x[None, None, :] * self.k[None, :, None])
            coef = wfft.coef(sw, Qal[a][:, None, :] * np.ones(self.Nk)[None, :, None], window=1, extrap="padding", kernel=kern)
            phase = (-1j) ** (2 * a) * (1j) ** (2 * np.arange(Nl))
            coef = phase[:, None, None] * coef
            Wal[a] = self.p**2 * np.real(np.einsum("lkn,np,ln->lkp", coef, pPow, M))
        self.Wal = Wal
        pp, kk = np.meshgrid(self.p, self.k, indexing="ij")
        mask = (pp < kk + cfg.windowk) & (pp > kk - cfg.windowk)
        dp = np.concatenate([[0], np.diff(self.p)])
        self.Waldk = np.einsum("alkp,pk,p->alkp", Wal, mask, dp)

    def window(self, st):
        """Window convolution of every template (reference window.py:371-415)."""
        out = dict(st)
        for name in ("P11l", "Pctl", "Ploopl") + (("Pstl",) if self.cfg.window_st else ()) + (("PctNNLOl",) if self.cfg.with_NNLO else ()):
            Pp = cubic_to(self.k, st[name], self.p)
            out[name] = np.einsum("alkp,lsp->ask", self.Waldk, Pp, optimize=True)
        return out

    # ------------------------------------------------------------------ binning / chained / reduce
    def _init_binning(self):
        """Bin edges, volumes, effective k, quadrature points (reference binning.py:100-129)."""
        kout = np.asarray(self.cfg.kout, dtype=float)
        dk = np.round(kout[-1] - kout[-2], 2)
        kc = (kout[-1] - dk * np.arange(len(kout)))[::-1]
        self.binmin, self.binmax = kc - dk / 2, kc + dk / 2
        self.binvol = np.array([quad(lambda k: k**2, a, b)[0] for a, b in zip(self.binmin, self.binmax)])
        self.keff = np.array([quad(lambda k: k**3, a, b)[0] for a, b in zip(self.binmin, self.binmax)]) / self.binvol
        self.points = np.array([np.linspace(a, b, 100 * self.cfg.binning_accboost) for a, b in zip(self.binmin, self.binmax)])

    def binning(self, st):
        """k^2-weighted bin average (reference binning.py:131-162)."""
        out = dict(st)
        for name in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc") + (("PctNNLOl",) if "PctNNLOl" in st else ()):
            Pk = interp1d(self.k, st[name], axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")
            out[name] = _trapz(Pk(self.points) * self.points**2, x=self.points, axis=-1) / self.binvol
        return out

    @staticmethod
    def chain_coeff(l):
        """(reference chained.py:13-29)"""
        return ((2 * l + 1) * legendre(l)(0)) / ((2 * l + 5) * legendre(l + 2)(0))

    def chained(self, st):
        """Q_l = P_l - A_l P_{l+2} (reference chained.py:32-68)."""
        Nl = st["P11l"].shape[0]
        mat = np.zeros((Nl - 1, Nl))
        for a in range(Nl - 1):
            mat[a, a], mat[a, a + 1] = 1.0, -self.chain_coeff(2 * a)
        out = dict(st)
        for name in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc") + (("PctNNLOl",) if "PctNNLOl" in st else ()):
            out[name] = np.einsum("al,l...->a...", mat, st[name], optimize=True)
        return out

    def bias_vectors(self, f, bsA, bsB=None, es=(0.0, 0.0, 0.0), counterform="westcoast"):
        """Bias vectors (reference parambasis.py:69-126); east-coast counter-terms: :99-106."""
        b1A, b2A, b3A, b4A, cctA, cr1A, cr2A = bsA
        b1B, b2B, b3B, b4B, cctB, cr1B, cr2B = bsB or bsA
        kmA, krA, ndA, kmB, krB, ndB = self.kmA, self.krA, self.ndA, self.kmB, self.krB, self.ndB
        ce0, cemono, cequad = es
        b11 = np.array([b1A * b1B, (b1A + b1B) * f, f**2])
        bct = np.array(
            [
                b1A * cctB / kmB**2 + b1B * cctA / kmA**2,
                b1B * cr1A / krA**2 + b1A * cr1B / krB**2,
                b1B * cr2A / krA**2 + b1A * cr2B / krB**2,
                (cctA / kmA**2 + cctB / kmB**2) * f,
                (cr1A / krA**2 + cr1B / krB**2) * f,
                (cr2A / krA**2 + cr2B / krB**2) * f,
            ]
        )
        bloop = np.array(
            [
                1.0, 0.5 * (b1A + b1B), 0.5 * (b2A + b2B), 0.5 * (b3A + b3B), 0.5 * (b4A + b4B), b1A * b1B,
                0.5 * (b1A * b2B + b1B * b2A), 0.5 * (b1A * b3B + b1B * b3A), 0.5 * (b1A * b4B + b1B * b4A),
                b2A * b2B, 0.5 * (b2A * b4B + b2B * b4A), b4A * b4B,
            ]
        )
        x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
        x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
        bst = np.array([ce0 * x1, cemono * x2, cequad * x2])
        if counterform == "eastcoast":
            bct = np.array([-cctA - cctB, -(cr1A + cr1B) * f, -(cr2A + cr2B) * f**2, 0.0, 0.0, 0.0])
        return b11, bloop, bct, bst

    def nnlo_vector(self, f, bsA, cnnloA, counterform="westcoast"):
        """bctNNLOAB (parambasis.py:96-106): west coast (cr4, cr6), east coast (ctilde, _)"""
        b1A = bsA[0]
        if counterform == "westcoast":
            cr4, cr6 = cnnloA
            return np.array([1 / 4 * b1A**2 / self.krA**4 * cr4, 1 / 4 * b1A / self.krA**4 * cr6, 0.0])
        ctilde = cnnloA[0]
        return ctilde * np.array([-(b1A**2) * f**4, -2 * b1A * f**5, -(f**6)])

    def reduce_plk(self, f, st, bsA, bsB=None, es=(0.0, 0.0, 0.0), counterform="westcoast", cnnloA=None):
        """P_l(k) = b11.P11l + bloop.Ploopl + bct.Pctl (+ bctNNLO.PctNNLOl) + bst.Pstl + Picc (parambasis.py:128-136)."""
        b11, bloop, bct, bst = self.bias_vectors(f, bsA, bsB, es, counterform)
        No = min(self.No, st["P11l"].shape[0])
        extra = 0.0
        if cnnloA is not None:
            extra = np.einsum("b,lbx->lx", self.nnlo_vector(f, bsA, cnnloA, counterform), st["PctNNLOl"][:No])
        return extra + (
            np.einsum("b,lbx->lx", b11, st["P11l"][:No])
            + np.einsum("b,lbx->lx", bloop, st["Ploopl"][:No])
            + np.einsum("b,lbx->lx", bct, st["Pctl"][:No])
            + np.einsum("b,lbx->lx", bst, st["Pstl"][:No])
            + st["Picc"][:No]
        )

    # ------------------------------------------------------------------ whole path
    def evaluate(self, kin, Pin, f, DA=None, H=None, pairwise=False, taps=None):
        """The kernel of reference theory.py:557-609 up to (not including) binning/chained."""
        st = self.pscf(kin, Pin, pairwise=pairwise)
        if taps is not None:
            taps["pscf"] = dict(st)
        st.update(self.set_pscfl(f, st))
        if taps is not None:
            taps["setpscfl"] = dict(st)
        if self.cfg.with_resum:
            st = self.resum(f, kin, Pin, st)
            if taps is not None:
                taps["resum"] = dict(st)
        if self.cfg.with_ap:
            st = self.ap(DA, H, st)
            if taps is not None:
                taps["ap"] = dict(st)
        if self.cfg.window_file is not None:
            st = self.window(st)
            if taps is not None:
                taps["window"] = dict(st)
        return st


def plk_interpolate(ls, kgrid, Plk, l, k):
    """PlkInterpolator (reference theory.py:75-106), restated with the same scipy call."""
    from scipy.interpolate import interp1d

    kg = np.hstack(([0], kgrid))
    P = np.insert(Plk, 0, 0, axis=-1)
    tmp = interp1d(kg, kg * P, axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")
    ll = [l] if np.ndim(l) == 0 else list(l)
    idx = [list(ls).index(x) for x in ll]
    out = (tmp(k) / k)[idx]
    return out[0] if len(idx) == 1 else out


# ----------------------------------------------------------------------------- WindowMatrix (reference window.py:426-586)
def to_window_matrix(matrix, in_ells, in_kmin, in_kmax, in_nbins, out_ells, out_kmin, out_kmax, out_nbins, ells_in, kmax_in, ells_out, kmin_out,
                     kmax_out):
    """Stacked matrix -> [len(ells_out), len(ells_in), nk_out, nk_in] (window.py:426-467, same masking rules)."""
    kedges = np.linspace(in_kmin, in_kmax, in_nbins + 1)
    kin = (kedges[1:] + kedges[:-1]) / 2
    mask_in = np.zeros(in_nbins * len(in_ells), dtype=bool)
    ileft, iright = 0, np.searchsorted(kin, kmax_in)
    for ell in in_ells:
        if ell in ells_in:
            mask_in[ileft:iright] = True
        ileft, iright = ileft + in_nbins, iright + in_nbins
    kedges = np.linspace(out_kmin, out_kmax, out_nbins + 1)
    kout = (kedges[1:] + kedges[:-1]) / 2
    mask_out = np.zeros(out_nbins * len(out_ells), dtype=bool)
    ileft, iright = np.searchsorted(kout, kmin_out), np.searchsorted(kout, kmax_out)
    for ell in out_ells:
        if ell in ells_out:
            mask_out[ileft:iright] = True
        ileft, iright = ileft + out_nbins, iright + out_nbins
    sub = matrix[np.ix_(mask_out, mask_in)]
    nk_out, nk_in = sub.shape[0] // len(ells_out), sub.shape[1] // len(ells_in)
    res = np.zeros((len(ells_out), len(ells_in), nk_out, nk_in))
    for i in range(len(ells_out)):
        for j in range(len(ells_in)):
            res[i, j] = sub[i * nk_out : (i + 1) * nk_out, j * nk_in : (j + 1) * nk_in]
    return res


def window_matrix_convolve(k, matrix, Plk):
    """WindowMatrix.convolve (window.py:548-564) with its hard-coded kavg"""
    kavg = np.linspace(0, 0.4, 400)[:300]
    return np.einsum("alkp,l...p->a...k", matrix, cubic_to(k, Plk, kavg), optimize=True)
