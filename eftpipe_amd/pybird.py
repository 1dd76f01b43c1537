"""Drop-in surface for ``eftpipe.pybird.pybird``: same class names, constructor arguments, method
names and in-place mutation semantics, every stage computed on the MI355X through libeftbird.so.

``eftpipe.theory`` looks these classes up by attribute on the module object (reference
eftpipe/theory.py:44, 358, 366, 423, 439, 568), so substituting this module for
``eftpipe.pybird.pybird`` swaps the engine without touching the likelihood.

Scope: Common, Bird(.setPsCfl), NonLinear(.PsCf), Resum(.Ps), APeffect(.AP);
NNLO, IRcutoff, optiresum, east-coast counterterms and FiberCollision are not on the path and raise.
There is no CPU fallback: the first stage call needs libeftbird.so and a HIP device.
"""
from __future__ import annotations

import weakref

import numpy as np

from . import _lib as L
from . import loopmath as lm
from ._log import HasLogger
from .engine import ROWS, Engine
from .synth import _GL_W, _GL_X
from .tables import EngineConfig


# ----------------------------------------------------------------------------- helpers (pybird.py:34-42)
def Hubble(Om, z):
    """E(z) of flat LCDM (reference pybird.py:34-36)."""
    return ((Om) * (1 + z) ** 3.0 + (1 - Om)) ** 0.5


def DAfunc(Om, z):
    """int_0^z dz'/E / (1+z) (reference pybird.py:39-42; 64-point Gauss-Legendre instead of quad)."""
    x = 0.5 * z * (_GL_X + 1.0)
    return float(0.5 * z * np.sum(_GL_W / Hubble(Om, x)) / (1.0 + z))


def get_kbird(kmax=0.30):
    """Native k grid (reference pybird.py:472-479)."""
    if kmax > 0.30:
        head = lm.native_k()[:8]
        rest = np.arange(head[-1], kmax + 1e-3, 0.005)
        return np.concatenate([head, rest[1:]])
    return lm.native_k()


sbird = lm.native_s()


# ----------------------------------------------------------------------------- Common
class Common(object):
    """Shared grids, counts and mu-weight tables (reference pybird.py:486-582)."""

    def __init__(self, Nl=None, No=None, kmax=0.3, optiresum=False, kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None,
                 ndB=None, counterform="westcoast", with_NNLO=False, kIR=None, IRcutoff=False):
        if IRcutoff and kIR is None:
            raise ValueError("kIR must be specified when doing IRcutoff")
        if IRcutoff is True:
            IRcutoff = "all"
        if IRcutoff not in (False, "all", "loop", "resum"):
            raise ValueError(f"unexpected IRcutoff option: {IRcutoff}")
        if counterform not in ("westcoast", "eastcoast"):  # the templates are the same; only reduce_Plk reads it
            raise ValueError(f"unexpected counterform: {counterform}")
        self.optiresum, self.with_NNLO, self.IRcutoff, self.kIR = bool(optiresum), bool(with_NNLO), IRcutoff, kIR
        self.counterform = counterform
        self.kmA, self.krA, self.ndA = kmA, krA, ndA
        self.kmB = kmA if kmB is None else kmB
        self.krB = krA if krB is None else krB
        self.ndB = ndA if ndB is None else ndB
        if Nl is None and No is None:
            self.Nl = self.No = 2
        elif not (Nl is None or No is None):
            self.Nl, self.No = Nl, No
        else:
            self.Nl = self.No = Nl or No
        assert isinstance(self.Nl, int) and isinstance(self.No, int)
        if self.No > self.Nl:
            raise ValueError("No should always be smaller than Nl")
        self.N11, self.Nct, self.NctNNLO, self.N22, self.N13, self.Nloop = 3, 6, 3, 28, 10, 12
        self.k = get_kbird(kmax)
        self.Nk = self.k.shape[0]
        self.s = np.arange(70.0, 200.0, 2.5) if self.optiresum else sbird  # reference pybird.py:553-556
        self.Ns = self.s.shape[0]
        self.kr = self.k[0.02 <= self.k]
        self.Nkr = self.kr.shape[0]
        self.Nklow = self.Nk - self.Nkr
        w = lm.mu_weights(self.Nl)
        self.l11, self.lct, self.l22, self.l13 = w["l11"], w["lct"], w["l22"], w["l13"]
        self.lctNNLO = w["lctNNLO"]


common = Common()

# one device engine per Common object (constant tables depend on co.k and co.Nl only)
_ENGINES: "weakref.WeakKeyDictionary[Common, Engine]" = weakref.WeakKeyDictionary()


def engine_for(co, nbinsmu=200, loop_cache=None):
    """The engine serving `co`, created on first use with the resum and AP tables resident."""
    eng = _ENGINES.get(co)
    if eng is not None and (eng.Nk != co.Nk or not np.array_equal(eng.k, co.k) or eng.cfg.nbinsmu != nbinsmu
                            or eng.cfg.with_NNLO != bool(co.with_NNLO) or eng.cfg.IRcutoff != co.IRcutoff or eng.cfg.kIR != co.kIR
                            or eng.cfg.optiresum != bool(co.optiresum)):
        eng.close()
        eng = None
    if eng is None:
        cfg = EngineConfig(Nl=co.Nl, k=np.array(co.k, dtype=np.float64), with_resum=True, with_ap=True, DA_AP=1.0, H_AP=1.0,
                           nbinsmu=nbinsmu, with_NNLO=bool(co.with_NNLO), IRcutoff=co.IRcutoff, kIR=co.kIR, optiresum=bool(co.optiresum))
        eng = _ENGINES[co] = Engine(cfg, max_batch=1, loop_cache=loop_cache)
    return eng


NS_DEV = 80  # s slots on the device; Common(optiresum=True) uses the first 52 (include/eftbird.h eftb_config.optiresum)


def _s_pad(a):
    """[..., Ns] -> [..., 80] (zeros in the unused slots)"""
    a = np.asarray(a, dtype=np.float64)
    if a.shape[-1] == NS_DEV:
        return a
    out = np.zeros(a.shape[:-1] + (NS_DEV,))
    out[..., : a.shape[-1]] = a
    return out


def _s_get(eng, name, shape, Ns):
    """device [..., 80] -> host [..., Ns]"""
    return np.ascontiguousarray(eng.get(name, tuple(shape) + (NS_DEV,))[..., :Ns])


# ----------------------------------------------------------------------------- Bird
class BirdSnapshot:
    """(reference pybird.py:616-632)"""

    def __init__(self, bird):
        self.k = bird.co.k.copy()
        self.ls = [2 * i for i in range(bird.co.Nl)]
        self.co, self.f = bird.co, bird.f
        for n in ("P11l", "Ploopl", "Pctl", "Pstl", "Picc"):
            setattr(self, n, getattr(bird, n).copy())
        self.PctNNLOl = None if bird.PctNNLOl is None else bird.PctNNLOl.copy()


class Bird:
    """Container of one evaluation (reference pybird.py:635-866).  Arrays are host NumPy, C-contiguous
    float64, re-bound by each stage exactly as in the reference; ``P11`` is filled by ``NonLinear.PsCf``
    (the reference interpolates it in ``__init__``; here the spline runs on the device with the loops)."""

    def __init__(self, kin, Plin, f, DA=None, H=None, z=None, co=common, rdrag=None, h=None):
        self.co = co
        self.f = f
        self.DA, self.H, self.z, self.rdrag, self.h = DA, H, z, rdrag, h
        self.kin = np.asarray(kin, dtype=np.float64)
        self.Pin = np.asarray(Plin, dtype=np.float64)
        Nl, Nk, Ns = co.Nl, co.Nk, co.Ns
        self.P11 = None
        self.P22, self.P13 = None, None
        self.C11 = self.Cct = self.C22 = self.C13 = None
        self.P11l = self.Pctl = self.Ploopl = self.Cloopl = self.Pstl = None
        self.PctNNLOl = self.CctNNLO = None  # filled when co.with_NNLO (reference pybird.py:741-748, 1098-1101)
        self.Picc = np.zeros((Nl, Nk))
        self.snapshots = {}
        self._engine = None

    def create_snapshot(self, name):
        if name not in self.snapshots:
            self.snapshots[name] = BirdSnapshot(self)

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("call NonLinear.PsCf(bird) first (reference order: theory.py:568-570)")
        return self._engine

    def _templates_to_device(self, eng):
        T = np.empty((self.co.Nl, 24, self.co.Nk))
        for n, sl in ROWS.items():
            T[:, sl] = getattr(self, n)
        eng.put("TEMPL", T)
        if self.co.with_NNLO:  # second block: PctNNLOl in the Pctl slots, zero elsewhere (include/eftbird.h EFTB_B_TEMPLN)
            T[:] = 0.0
            T[:, 3:6] = self.PctNNLOl
            eng.put("TEMPLN", T)

    def _templates_from_device(self, eng, names=("P11l", "Pctl", "Ploopl", "Pstl")):
        T = eng.get("TEMPL", (self.co.Nl, 24, self.co.Nk))
        for n in names:
            setattr(self, n, np.ascontiguousarray(T[:, ROWS[n]]))
        if self.co.with_NNLO:
            self.PctNNLOl = np.ascontiguousarray(eng.get("TEMPLN", (self.co.Nl, 24, self.co.Nk))[:, 3:6])

    def setPsCfl(self):
        """Multipole weights, regrouping into the 12 bias groups, shot-noise subtraction, stochastic
        templates (reference pybird.py:737-866) -- regroup_kernel / regroup_cf_kernel."""
        eng, co = self._need_engine(), self.co
        eng.put("F", np.array([self.f]))
        eng.put("P11", self.P11)
        eng.put("P22", self.P22)
        eng.put("P13", self.P13)
        eng.put("CC", np.concatenate([_s_pad(self.C22).reshape(-1), _s_pad(self.C13).reshape(-1)]))
        eng.run(L.S_REGROUP)
        self._templates_from_device(eng)
        self.Cloopl = _s_get(eng, "CLOOPL", (co.Nl, co.Nloop), co.Ns)
        # the reference also leaves the mu-weighted pieces on the bird (pybird.py:749-753)
        self.P22l = co.l22[:, :, None] * self.P22[None]
        self.P13l = co.l13[:, :, None] * self.P13[None]
        self.C22 = self.C22 * co.l22[:, :, None]
        self.C13 = self.C13 * co.l13[:, :, None]


# ----------------------------------------------------------------------------- NonLinear
class NonLinear(HasLogger):
    """One-loop P(k) and xi(s) pieces (reference pybird.py:870-1171).  `load` / `save` / `path` follow the reference
    (pybird.py:917-981): the loop matrices are read from / written to `path/pyegg{NFFT}_Nl{Nl}.npz` in the reference's own
    layout (keys Pow, M22, M13, Mcf11, Mcf22, Mcf13, Mcfct, McfctNNLO), so existing caches drop in; a cache written for
    another FFTLog configuration, or an unreadable one, is ignored with a warning and the matrices are recomputed."""

    def __init__(self, load=True, save=True, path="./", NFFT=256, co=common, name="pybird.nonlinear"):
        from .tables import PYEGG_KEYS, loop_matrices, pyegg_path

        self.set_logger(name=name)
        if NFFT != 256:
            raise NotImplementedError("the HIP engine is specialised for NFFT=256 (reference default)")
        self.co = co
        self.fftsettings = dict(Nmax=NFFT, xmin=1.5e-5, xmax=1000.0, bias=-1.6)
        egg = pyegg_path(path, NFFT, co.Nl)
        cache = None
        if load is True:
            try:
                with np.load(egg) as z:
                    cache = {k: z[k] for k in ("Pow", "M22", "M13", "Mcf11", "Mcfct", "McfctNNLO")}
                save = False
            except Exception:  # same policy as the reference: warn and recompute
                self.mpi_warning("Can't load loop matrices at %s, computing new matrices.", path)
        if cache is not None:
            try:
                self.engine = engine_for(co, loop_cache=cache)
            except ValueError:
                self.mpi_warning("Loaded loop matrices do not correspond to asked FFTLog configuration, computing new matrices.")
                cache, save = None, save
        if cache is None:
            self.engine = engine_for(co)
        self.loaded = cache is not None
        if save is True and cache is None:
            try:
                mats = loop_matrices(co.Nl, NFFT)
                np.savez(egg, **{k: mats[k] for k in PYEGG_KEYS})
            except Exception:
                self.mpi_warning("Can't save loop matrices at %s.", path)

    def PsCf(self, bird, window=0.2):
        """FFTLog of P_lin + P22, P13, C11, Cct, C22, C13 (reference pybird.py:1143-1171) --
        prep_kernel, antidiag_kernel, build_rows_kernel, synth_kernel (FP64 MFMA), expand_kernel."""
        if window != self.engine.cfg.fft_window:
            raise NotImplementedError(f"engine built for FFTLog window={self.engine.cfg.fft_window}")
        if not np.array_equal(bird.kin, self.engine.kin):
            raise ValueError("bird.kin differs from the engine's input grid (reference theory.py:562 uses logspace(-5, 0, 200))")
        eng, co = self.engine, self.co
        bird._engine = eng
        eng.put("PIN", bird.Pin)
        eng.run(L.S_PREP | L.S_LOOPS | L.S_CF)
        bird.P11 = eng.get("P11", (co.Nk,))
        bird.P22 = eng.get("P22", (co.N22, co.Nk))
        bird.P13 = eng.get("P13", (co.N13, co.Nk))
        bird.C11 = _s_get(eng, "C11", (co.Nl,), co.Ns)
        bird.Cct = _s_get(eng, "CCT", (co.Nl,), co.Ns)
        if co.with_NNLO:
            bird.CctNNLO = _s_get(eng, "CCTN", (co.Nl,), co.Ns)
        cc = _s_get(eng, "CC", (co.Nl * 38,), co.Ns)
        bird.C22 = np.ascontiguousarray(cc[: co.Nl * 28].reshape(co.Nl, 28, co.Ns))
        bird.C13 = np.ascontiguousarray(cc[co.Nl * 28 :].reshape(co.Nl, 10, co.Ns))


# ----------------------------------------------------------------------------- Resum
class Resum(HasLogger):
    """IR-resummation (reference pybird.py:1174-1464) -- irfilter_kernel, resum_prep_kernel, resum_mfma_kernel."""

    def __init__(self, LambdaIR=0.2, NFFT=192, co=common, name="pybird.IRresum", snapshot=False):
        self.set_logger(name=name)
        if LambdaIR != 0.2 or NFFT != 192:
            raise NotImplementedError("engine tables are built for LambdaIR=0.2, NFFT=192 (reference defaults)")
        self.co, self.LambdaIR = co, LambdaIR
        self.NIR = 16 if co.Nl == 3 else 8
        self.Na = 3 if self.NIR == 16 else 2
        self.Nn = self.NIR * self.Na * 2
        self.snapshot = snapshot
        self.engine = engine_for(co)
        self.Q = None
        if co.optiresum:  # reference pybird.py:1235-1244
            self.sLow, self.sHigh = 70.0, 190.0
            self.idlow = int(np.where(co.s > self.sLow)[0][0])
            self.idhigh = int(np.where(co.s > self.sHigh)[0][0])
            self.sbao = co.s[self.idlow : self.idhigh]
            self.snobao = np.concatenate([co.s[: self.idlow], co.s[self.idhigh :]])
            self.sr = self.sbao
            self._sr = slice(self.idlow, self.idhigh)
        else:
            self.sr = co.s
            self._sr = slice(0, co.Ns)

    def IRFilters(self, bird):
        """X(s), Y(s) (reference pybird.py:1316-1353)."""
        eng = self.engine
        eng.put("PIN", bird.Pin)
        eng.put("F", np.array([bird.f]))
        eng.run(L.S_RESUM)  # cheap at B=1; filters are a by-product
        xy = eng.get("XY", (2, NS_DEV))[:, self._sr]
        return np.ascontiguousarray(xy[0]), np.ascontiguousarray(xy[1])

    def Ps(self, bird, window=None):
        """Adds the IR corrections to bird.P11l / Pctl / Ploopl in place (reference pybird.py:1413-1464)."""
        if window is not None:
            raise NotImplementedError("Resum.Ps(window=...) is not on the accelerated path")
        eng, co = self.engine, self.co
        eng.put("PIN", bird.Pin)
        eng.put("F", np.array([bird.f]))
        eng.put("C11", _s_pad(bird.C11))
        eng.put("CCT", _s_pad(bird.Cct))
        eng.put("CLOOPL", _s_pad(bird.Cloopl))
        if co.with_NNLO:
            eng.put("CCTN", _s_pad(bird.CctNNLO))
        bird._templates_to_device(eng)
        eng.run(L.S_RESUM)
        bird._templates_from_device(eng, ("P11l", "Pctl", "Ploopl"))
        self.Q = eng.get("Q", (2, co.Nl, co.Nl, self.Nn))
        if self.snapshot:
            bird.create_snapshot("IRresum")


# ----------------------------------------------------------------------------- APeffect
class APeffect(HasLogger):
    """Alcock-Paczynski distortion (reference pybird.py:1467-1629) -- spline_kernel, ap_prefix_kernel, ap_apply_kernel."""

    def __init__(self, Om_AP=None, z_AP=None, DA=None, H=None, rdrag_AP=None, h_AP=None, nbinsmu=200, accboost=1,
                 Nlmax=None, APst=False, co=common, name="pybird.apeffect", snapshot=False):
        self.set_logger(name=name)
        self.co, self.APst = co, APst
        if (DA is not None) and (H is not None):
            self.DA, self.H = DA, H
        elif (Om_AP is not None) and (z_AP is not None):
            self.mpi_warning("DA and H not given, compute using Om_AP and z_AP instead")
            self.DA, self.H = DAfunc(Om_AP, z_AP), float(Hubble(Om_AP, z_AP))
        else:
            raise ValueError("expect input params: Om_AP and z_AP, or DA and H")
        if Nlmax is not None and Nlmax != co.Nl:
            raise NotImplementedError("Nlmax != co.Nl is not supported (the reference raises on any integer Nlmax, pybird.py:1541)")
        self.Nlmax = co.Nl
        self.rdrag_AP, self.h_AP = rdrag_AP, h_AP
        self._rdrag_warned = False
        self.nbinsmu = accboost * nbinsmu
        self.snapshot = snapshot
        self.engine = engine_for(co, nbinsmu=self.nbinsmu)

    def get_AP_param(self, bird):
        """qperp, qpar (reference pybird.py:1554-1562)."""
        return bird.DA / self.DA, self.H / bird.H

    def get_alperp_alpara(self, bird):
        """(reference pybird.py:1564-1579)"""
        if any(x is None for x in (self.rdrag_AP, self.h_AP, bird.rdrag, bird.h)):
            if not self._rdrag_warned:
                self.mpi_warning("rdrag_AP or h_AP or bird.rdrag or bird.h not given, fallback to qperp and qpara")
                self._rdrag_warned = True
            return self.get_AP_param(bird)
        ratio = (self.rdrag_AP * self.h_AP) / (bird.rdrag * bird.h)
        return bird.DA / self.DA * ratio, self.H / bird.H * ratio

    def AP(self, bird, q=None):
        """Re-binds bird.P11l / Pctl / Ploopl (/ Pstl if APst) (reference pybird.py:1598-1621)."""
        eng = self.engine
        qperp, qpar = self.get_AP_param(bird) if q is None else q
        eng.set_ap_fiducial(self.DA, self.H)
        eng.set_ap_stochastic(self.APst)
        eng.put("DA", np.array([qperp * self.DA]))
        eng.put("H", np.array([self.H / qpar]))
        bird._templates_to_device(eng)
        eng.run(L.S_AP)
        bird._templates_from_device(eng)
        if self.snapshot:
            bird.create_snapshot("APeffect")


class FiberCollision(HasLogger):
    """Fibre-collision correction, effective-window method (same surface as reference pybird.py:1630-1810).
    The correlated correction is linear in the templates: F is built once on the host (tables.fiber_operator) and
    ``fibcolWindow`` is one dense operator on the FP64 matrix cores (P += F P; Pstl only if ``fiberst``)."""

    def __init__(self, fs, Dfc, ktrust=0.25, fiberst=False, co=None, name="pybird.fiber", snapshot=False):
        self.set_logger(name=name)
        self.ktrust, self.fiberst, self.fs, self.Dfc = ktrust, fiberst, fs, Dfc
        self.co = common if co is None else co
        self.snapshot = snapshot
        self._F = None
        self._op = None

    def dPuncorr(self, kout, fs=0.6, Dfc=0.43 / 0.6777):
        """Uncorrelated contribution (reference pybird.py:1680-1701)"""
        from scipy.special import legendre

        kout = np.asarray(kout, dtype=float)
        return np.stack([-fs * np.pi * Dfc**2.0 * (2.0 * np.pi / kout) * (2.0 * l + 1.0) / 2.0 * legendre(l)(0) * (1.0 - (kout * Dfc) ** 2 / 8.0)
                         for l in (0, 2, 4)])

    def operator(self):
        """F [Nl, Nl, Nk, Nk] of dPcorr on the engine grid"""
        if self._F is None:
            from .tables import fiber_operator

            self._F = fiber_operator(self.co.k, self.co.Nl, self.fs, self.Dfc, self.ktrust)
        return self._F

    def dPcorr(self, kout, kPS, PS, ktrust=0.25, fs=0.6, Dfc=0.43 / 0.6777):
        """Correlated contribution for PS [Nl, n, len(kPS)] -> [Nl, n, len(kout)] (host; reference pybird.py:1703-1757)"""
        from .tables import fiber_operator

        return np.einsum("alxk,lnk->anx", fiber_operator(kPS, self.co.Nl, fs, Dfc, ktrust, kout=kout), np.asarray(PS)[: self.co.Nl])

    def fibcolWindow(self, bird):
        """P11l, Pctl, Ploopl (and Pstl if fiberst) += dPcorr, in place (reference pybird.py:1760-1810)"""
        from .tables import compose_operator
        from .transformer import apply_operator_to_birdlike

        eng = engine_for(bird.co)
        if self._op is None or self._op[0] is not eng:
            self._op = (eng, eng.add_operator(compose_operator(self.co.Nl, self.co.Nk, fiber=self.operator())))
        keep = bird.Pstl
        out = apply_operator_to_birdlike(eng, self._op[1], bird)
        bird.P11l, bird.Pctl, bird.Ploopl = out["P11l"], out["Pctl"], out["Ploopl"]
        bird.Pstl = out["Pstl"] if self.fiberst else keep
        if "PctNNLOl" in out:
            bird.PctNNLOl = out["PctNNLOl"]
        if self.snapshot:
            bird.create_snapshot("fiber")
