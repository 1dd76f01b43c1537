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


_ENGINE_DEFAULTS = dict(nbinsmu=200, LambdaIR=0.2, NFFT_resum=192, resum_window=None, fft_window=0.2, kin=None,
                        irf_soffset=1.0, irf_rescale=1.0, irf_window=None)
_DEFAULT_KIN = np.logspace(-5, 0, 200)  # reference theory.py:562


def engine_for(co, loop_cache=None, **opts):
    """The engine serving `co`, created on first use with the resum and AP tables resident.  The plugin constructors record the
    options that shape the device tables (APeffect: mu nodes; Resum: LambdaIR, NFFT, coefficient window) on `co` itself, so every
    plugin of one tracer finds the same engine whatever the construction order; a changed option rebuilds the tables."""
    store = co.__dict__.setdefault("_engine_opts", dict(_ENGINE_DEFAULTS))
    for k in opts:
        if k not in _ENGINE_DEFAULTS:
            raise TypeError(f"unknown engine option {k!r}")
    store.update(opts)
    if store["kin"] is not None and np.array_equal(store["kin"], _DEFAULT_KIN):
        store["kin"] = None
    eng = _ENGINES.get(co)
    kin_now = _DEFAULT_KIN if store["kin"] is None else store["kin"]
    if eng is not None and (eng.Nk != co.Nk or not np.array_equal(eng.k, co.k) or eng.cfg.nbinsmu != store["nbinsmu"]
                            or eng.cfg.fft_window != store["fft_window"] or not np.array_equal(eng.kin, kin_now)
                            or (eng.cfg.irf_soffset, eng.cfg.irf_rescale, eng.cfg.irf_window) != (store["irf_soffset"], store["irf_rescale"], store["irf_window"])
                            or eng.cfg.LambdaIR != store["LambdaIR"] or eng.cfg.NFFT_resum != store["NFFT_resum"]
                            or eng.cfg.resum_window != store["resum_window"]
                            or eng.cfg.with_NNLO != bool(co.with_NNLO) or eng.cfg.IRcutoff != co.IRcutoff or eng.cfg.kIR != co.kIR
                            or eng.cfg.optiresum != bool(co.optiresum)):
        release(eng)
        eng.close()
        eng = None
    if eng is None:
        cfg = EngineConfig(Nl=co.Nl, k=np.array(co.k, dtype=np.float64), with_resum=True, with_ap=True, DA_AP=1.0, H_AP=1.0,
                           nbinsmu=store["nbinsmu"], LambdaIR=store["LambdaIR"], NFFT_resum=store["NFFT_resum"], resum_window=store["resum_window"],
                           with_NNLO=bool(co.with_NNLO), IRcutoff=co.IRcutoff, kIR=co.kIR, optiresum=bool(co.optiresum),
                           fft_window=store["fft_window"], kin=None if store["kin"] is None else np.array(store["kin"], dtype=np.float64),
                           irf_soffset=store["irf_soffset"], irf_rescale=store["irf_rescale"], irf_window=store["irf_window"])
        eng = _ENGINES[co] = Engine(cfg, max_batch=1, loop_cache=loop_cache)
    return eng


def release(eng):
    """Bring the device-resident results of the bird that currently owns `eng` to the host (they are about to be overwritten)."""
    ref = getattr(eng, "_owner", None)
    owner = ref() if ref is not None else None
    if owner is not None:
        owner._flush()
    eng._owner = None


def claim(eng, bird):
    """`bird` is about to run a stage on `eng`: one engine serves one Common, so another bird's pending results are fetched first."""
    ref = getattr(eng, "_owner", None)
    owner = ref() if ref is not None else None
    if owner is not bird:
        if owner is not None:
            owner._flush()
        if getattr(bird, "_dev", None) is not None:
            bird._dev.clear()
        eng._owner = weakref.ref(bird)


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


# host attribute -> (device buffer, fetch group).  One group = one transfer that fills every pending attribute of the group.
_LAZY = {
    "P11": "P11", "P22": "P22", "P13": "P13", "C11": "C11", "Cct": "CCT", "CctNNLO": "CCTN", "C22": "CC", "C13": "CC",
    "Cloopl": "CLOOPL", "P11l": "TEMPL", "Pctl": "TEMPL", "Ploopl": "TEMPL", "Pstl": "TEMPL", "PctNNLOl": "TEMPLN",
    "P22l": "P22", "P13l": "P13",
}
_INPUTS = {"Pin": "PIN", "f": "F"}  # re-binding these invalidates the device copy too


class Bird:
    """Container of one evaluation (reference pybird.py:635-866).  Arrays are host NumPy, C-contiguous float64, re-bound by each stage
    exactly as in the reference -- but LAZILY: a stage leaves its outputs on the device and the attribute is fetched on first access
    (``bird.P11l`` etc. behave as ordinary arrays from then on).  The next stage uses the device copy as long as the attribute has
    neither been read nor re-bound on the host (a read hands out an array the caller may modify in place, so it ends the device
    copy's validity); the call sequence of reference theory.py:568-604 therefore moves P_lin, f, DA, H in and the final templates out.
    ``P11`` is filled by ``NonLinear.PsCf`` (the reference interpolates it in ``__init__``; here the spline runs on the device with the loops)."""

    def __init__(self, kin, Plin, f, DA=None, H=None, z=None, co=common, rdrag=None, h=None):
        d = self.__dict__
        d["_pending"] = {}      # attribute -> fetch group (device buffer) whose content is this bird's current value
        d["_dev"] = set()       # device buffers that hold this bird's current values (valid while the engine's owner is this bird)
        d["_shape"] = None      # (nl, nx) of the template block on the device
        d["_engine"] = None
        self.co = co
        self.f = f
        self.DA, self.H, self.z, self.rdrag, self.h = DA, H, z, rdrag, h
        self.kin = np.asarray(kin, dtype=np.float64)
        self.Pin = np.asarray(Plin, dtype=np.float64)
        Nl, Nk = co.Nl, co.Nk
        self.P11 = None
        self.P22, self.P13 = None, None
        self.C11 = self.Cct = self.C22 = self.C13 = None
        self.P11l = self.Pctl = self.Ploopl = self.Cloopl = self.Pstl = None
        self.PctNNLOl = self.CctNNLO = None  # filled when co.with_NNLO (reference pybird.py:741-748, 1098-1101)
        self.Picc = np.zeros((Nl, Nk))
        self.snapshots = {}

    # ---- lazy attributes
    def __getattr__(self, name):  # reached only when `name` is not in __dict__: a pending device result, or a genuine miss
        pend = self.__dict__.get("_pending")
        if pend and name in pend:
            self._materialize(pend[name])
            return self.__dict__[name]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def __setattr__(self, name, value):
        d = self.__dict__
        if "_pending" in d:
            d["_pending"].pop(name, None)
            buf = _LAZY.get(name) or _INPUTS.get(name)
            if buf is not None:
                d["_dev"].discard(buf)  # the host value is the truth from now on
        d[name] = value

    def _set_pending(self, names, group):
        for n in names:
            self.__dict__.pop(n, None)
            self._pending[n] = group
        self._dev.add(group)

    def _owns(self, eng):
        ref = getattr(eng, "_owner", None)
        return ref is not None and ref() is self

    def _materialize(self, group):
        """One transfer for every pending attribute of the group; a fetched array may be modified by the caller, so the device copy
        stops counting as this bird's value (the next stage uploads the host arrays again)."""
        eng, co, d = self._engine, self.co, self.__dict__
        want = [n for n, g in self._pending.items() if g == group]
        if not want:
            return
        if not self._owns(eng):
            raise RuntimeError("device results of this Bird were overwritten (one engine serves one Common; evaluate birds one at a time)")
        if group in ("TEMPL", "TEMPLN"):
            nl, nx = self._shape
            T = eng.get(group, (nl, 24, nx))
            for n in want:
                d[n] = np.ascontiguousarray(T[:, 3:6] if n == "PctNNLOl" else T[:, ROWS[n]])
        elif group == "CC":
            cc = _s_get(eng, "CC", (co.Nl * 38,), co.Ns)
            raw = {"C22": cc[: co.Nl * 28].reshape(co.Nl, 28, co.Ns), "C13": cc[co.Nl * 28 :].reshape(co.Nl, 10, co.Ns)}
            wt = {"C22": co.l22, "C13": co.l13}
            for n in want:  # after setPsCfl the reference leaves the mu-weighted pieces on the bird (pybird.py:752-753)
                d[n] = np.ascontiguousarray(raw[n] * wt[n][:, :, None] if self.__dict__.get("_cf_weighted") else raw[n])
        elif group in ("P22", "P13"):
            n0 = group
            full = eng.get(group, (co.N22 if group == "P22" else co.N13, co.Nk))
            for n in want:
                d[n] = full if n == n0 else (co.l22 if group == "P22" else co.l13)[:, :, None] * full[None]  # P22l / P13l (pybird.py:749-750)
        elif group == "P11":
            d["P11"] = eng.get("P11", (co.Nk,))
        elif group in ("C11", "CCT", "CCTN"):
            for n in want:
                d[n] = _s_get(eng, group, (co.Nl,), co.Ns)
        elif group == "CLOOPL":
            d["Cloopl"] = _s_get(eng, "CLOOPL", (co.Nl, co.Nloop), co.Ns)
        else:  # pragma: no cover
            raise KeyError(group)
        for n in want:
            del self._pending[n]
        self._dev.discard(group)

    def _flush(self):
        """Fetch everything still pending (another bird is about to use the engine)."""
        for group in list(dict.fromkeys(self._pending.values())):
            self._materialize(group)

    def _on_device(self, eng, buf):
        return self._owns(eng) and buf in self._dev

    def create_snapshot(self, name):
        if name not in self.snapshots:
            self.snapshots[name] = BirdSnapshot(self)

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("call NonLinear.PsCf(bird) first (reference order: theory.py:568-570)")
        return self._engine

    def _templates_to_device(self, eng):
        """The template block (and the NNLO block) on the device: nothing to do while the stage outputs are still there."""
        if not self._on_device(eng, "TEMPL"):
            T = np.empty((self.P11l.shape[0], 24, self.P11l.shape[-1]))
            for n, sl in ROWS.items():
                T[:, sl] = getattr(self, n)
            eng.set_template_dims(T.shape[0], T.shape[2])
            eng.put("TEMPL", T)
            self._dev.add("TEMPL")
            self._shape = (T.shape[0], T.shape[2])
        if self.co.with_NNLO and not self._on_device(eng, "TEMPLN"):  # second block: PctNNLOl in the Pctl slots, zero elsewhere (include/eftbird.h EFTB_B_TEMPLN)
            T = np.zeros((self.PctNNLOl.shape[0], 24, self.PctNNLOl.shape[-1]))
            T[:, 3:6] = self.PctNNLOl
            eng.put("TEMPLN", T)
            self._dev.add("TEMPLN")

    def _templates_pending(self, names=("P11l", "Pctl", "Ploopl", "Pstl"), shape=None):
        """The stage that just ran re-binds these attributes: they now live on the device."""
        if shape is not None:
            self._shape = shape
        # attributes of the block that are NOT re-bound keep their host value if they have one; since the whole block is valid on the
        # device, they may as well stay pending if they were
        self._set_pending(names, "TEMPL")
        if self.co.with_NNLO:
            self._set_pending(("PctNNLOl",), "TEMPLN")

    def setPsCfl(self):
        """Multipole weights, regrouping into the 12 bias groups, shot-noise subtraction, stochastic
        templates (reference pybird.py:737-866) -- regroup_kernel / regroup_cf_kernel."""
        self._need_engine()
        eng, co = engine_for(self.co), self.co
        claim(eng, self)
        self.__dict__["_engine"] = eng
        if not self._on_device(eng, "F"):
            eng.put("F", np.array([self.f]))
            self._dev.add("F")
        for n in ("P11", "P22", "P13"):
            if not self._on_device(eng, n):
                eng.put(n, getattr(self, n))
                self._dev.add(n)
        if not self._on_device(eng, "CC"):
            eng.put("CC", np.concatenate([_s_pad(self.C22).reshape(-1), _s_pad(self.C13).reshape(-1)]))
            self._dev.add("CC")
        eng.run(L.S_REGROUP, sync=False)
        self._templates_pending(shape=(co.Nl, co.Nk))
        self._set_pending(("Cloopl",), "CLOOPL")
        # the reference also leaves the mu-weighted pieces on the bird (pybird.py:749-753): derived on access
        self._set_pending(("P22l",), "P22")
        self._set_pending(("P13l",), "P13")
        if "C22" in self._pending or "C13" in self._pending:
            self.__dict__["_cf_weighted"] = True
        for n, wt in (("C22", co.l22), ("C13", co.l13)):
            if n not in self._pending:
                self.__dict__[n] = self.__dict__[n] * wt[:, :, None]


    # ---- host forms of the pieces of setPsCfl (reference pybird.py:758-866), for callers that use them on their own; setPsCfl itself
    #      runs regroup_kernel / regroup_cf_kernel
    def reducePsCfl(self):
        """Ploopl / Cloopl from the mu-weighted P22l, P13l / C22, C13 (reference pybird.py:758-846), then subtractShotNoise."""
        co, f = self.co, self.f
        Ploopl, Cloopl = np.zeros((co.Nl, co.Nloop, co.Nk)), np.zeros((co.Nl, co.Nloop, co.Ns))
        for b in range(co.N22):
            grp, pw = lm.GROUP_22[b]
            Ploopl[:, grp] += f**pw * self.P22l[:, b]
            Cloopl[:, grp] += f**pw * self.C22[:, b]
        for b in range(co.N13):
            grp, pw = lm.GROUP_13[b]
            Ploopl[:, grp] += f**pw * self.P13l[:, b]
            Cloopl[:, grp] += f**pw * self.C13[:, b]
        self.Ploopl, self.Cloopl = Ploopl, Cloopl
        self.subtractShotNoise()

    def setPstl(self):
        """(reference pybird.py:848-857)"""
        co = self.co
        Pstl = np.zeros((co.Nl, 3, co.Nk))
        Pstl[0, 0], Pstl[0, 1] = 1.0, co.k**2
        if co.Nl >= 2:
            Pstl[1, 2] = co.k**2
        self.Pstl = Pstl

    def subtractShotNoise(self):
        """(reference pybird.py:859-866)"""
        P = np.array(self.Ploopl, dtype=np.float64, copy=True)
        P -= P[:, :, :1]
        self.Ploopl = P


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
                engine_for(co, loop_cache=cache)
            except ValueError:
                self.mpi_warning("Loaded loop matrices do not correspond to asked FFTLog configuration, computing new matrices.")
                cache, save = None, save
        if cache is None:
            engine_for(co)
        self.loaded = cache is not None
        if save is True and cache is None:
            try:
                mats = loop_matrices(co.Nl, NFFT)
                np.savez(egg, **{k: mats[k] for k in PYEGG_KEYS})
            except Exception:
                self.mpi_warning("Can't save loop matrices at %s.", path)

    @property
    def engine(self):
        return engine_for(self.co)

    def Coef(self, bird, window=None, IRcut=False):
        """FFTLog coefficients of the linear spectrum [NFFT + 1] complex (reference pybird.py:1127-1141), host form for callers of the
        helper (the device keeps its own real-reduced copy, EFTB_B_COEF)."""
        from .tables import FFTLogOperator

        kin, Pin = bird.kin, bird.Pin
        if IRcut:
            idx = int(np.searchsorted(kin, self.co.kIR))
            op = FFTLogOperator(256, 1.5e-5, 1000.0, -1.6, kin[idx:], window, extrap=("padding", "extrap"))
            Pin = Pin[idx:]
            kin = kin[idx:]
        else:
            op = FFTLogOperator(256, 1.5e-5, 1000.0, -1.6, kin, window)
        c = op.G @ Pin
        if op.high_active:
            slope = (np.log(Pin[-1]) - np.log(Pin[-2])) / (np.log(kin[-1]) - np.log(kin[-2]))
            c = c + op.E_hi @ (Pin[-1] / kin[-1] ** slope * np.exp(slope * op.lnx_hi))
        if op.low_active:
            slope = (np.log(Pin[1]) - np.log(Pin[0])) / (np.log(kin[1]) - np.log(kin[0]))
            c = c + op.E_lo @ (Pin[0] / kin[0] ** slope * np.exp(slope * op.lnx_lo))
        return c

    def PsCf(self, bird, window=0.2):
        """FFTLog of P_lin + P22, P13, C11, Cct, C22, C13 (reference pybird.py:1143-1171) --
        prep_kernel, antidiag_kernel, build_rows_kernel, synth_kernel (FP64 MFMA), expand_kernel."""
        # the FFTLog coefficient window and the input grid shape the first-stage operator tables: another value rebuilds them (as
        # Resum(LambdaIR, NFFT) does for its own); the shipped callers pass window=0.2 and kin = logspace(-5, 0, 200) (theory.py:562)
        eng, co = engine_for(self.co, fft_window=window, kin=bird.kin), self.co
        claim(eng, bird)
        bird.__dict__["_engine"] = eng
        bird.__dict__["_cf_weighted"] = False
        eng.put("PIN", bird.Pin)
        eng.put("F", np.array([bird.f]))   # the growth rate is known from the start: setPsCfl and Resum.Ps find it in place
        bird._dev.update(("PIN", "F"))
        eng.run(L.S_PREP | L.S_LOOPS | L.S_CF, sync=False)
        for n in ("P11", "P22", "P13", "C11"):
            bird._set_pending((n,), _LAZY[n])
        bird._set_pending(("Cct",), "CCT")
        if co.with_NNLO:
            bird._set_pending(("CctNNLO",), "CCTN")
        bird._set_pending(("C22", "C13"), "CC")


# ----------------------------------------------------------------------------- Resum
class Resum(HasLogger):
    """IR-resummation (reference pybird.py:1174-1464) -- irfilter_kernel, resum_prep_kernel, resum_mfma_kernel."""

    def __init__(self, LambdaIR=0.2, NFFT=192, co=common, name="pybird.IRresum", snapshot=False):
        self.set_logger(name=name)
        if NFFT % 2 or NFFT < 8:
            raise ValueError(f"expected even Nmax, instead of Nmax={NFFT}")  # reference fftlog.py:62-63
        self.co, self.LambdaIR, self.NFFT = co, LambdaIR, NFFT
        self.NIR = 16 if co.Nl == 3 else 8
        self.Na = 3 if self.NIR == 16 else 2
        self.Nn = self.NIR * self.Na * 2
        self.snapshot = snapshot
        self._window = None
        engine_for(co, LambdaIR=LambdaIR, NFFT_resum=NFFT, resum_window=None)
        self._Q = None
        self._Qf = None
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

    @property
    def Q(self):
        """Resum.Q of the last ``Ps`` / ``makeQ`` call [2, Nl, Nl, Nn] (reference pybird.py:1367-1380) -- the polynomials in f, evaluated
        on the host when asked for (the device evaluates its own copy inside the stage)."""
        if self._Q is None and self._Qf is not None:
            self.makeQ(self._Qf)
        return self._Q

    @Q.setter
    def Q(self, value):
        self._Q = value

    def makeQ(self, f):
        """(reference pybird.py:1367-1380)  Q[a] = table[1 - a](f)"""
        P = lm.q_polynomials(self.co.Nl)                      # [2, Nl, Nl, Nn, 15] coefficients of f^0..f^14
        val = np.zeros(P.shape[:-1])
        for p in range(P.shape[-1] - 1, -1, -1):
            val = val * f + P[..., p]
        self._Q = np.ascontiguousarray(val[::-1])
        self._Qf = f

    @property
    def engine(self):
        return engine_for(self.co, LambdaIR=self.LambdaIR, NFFT_resum=self.NFFT, resum_window=self._window)

    def _inputs(self, bird, eng=None):
        eng = self.engine if eng is None else eng
        claim(eng, bird)
        bird.__dict__["_engine"] = eng
        if not bird._on_device(eng, "PIN"):
            eng.put("PIN", bird.Pin)
            bird._dev.add("PIN")
        if not bird._on_device(eng, "F"):
            eng.put("F", np.array([bird.f]))
            bird._dev.add("F")
        return eng

    def IRFilters(self, bird, soffset=1.0, LambdaIR=None, RescaleIR=1.0, window=None):
        """X(s), Y(s) (reference pybird.py:1316-1353); only the arguments the reference's own callers use are supported."""
        default = soffset == 1.0 and RescaleIR == 1.0 and window is None and (LambdaIR is None or LambdaIR == self.LambdaIR)
        special = None
        if not default:
            # other arguments select other X / Y operator tables, the way Resum(LambdaIR, NFFT) selects its own: the engine's tables are rebuilt
            # for this call and put back afterwards (Resum.Ps always runs with the defaults, reference pybird.py:1424)
            special = engine_for(self.co, LambdaIR=self.LambdaIR if LambdaIR is None else LambdaIR, NFFT_resum=self.NFFT, resum_window=self._window,
                                 irf_soffset=float(soffset), irf_rescale=float(RescaleIR), irf_window=window)
        try:
            eng = self._inputs(bird, special)
            eng.run(L.K_IRFILTER)  # irfilter_kernel alone: X, Y, Q(f) from the inputs; nothing else on the device is touched
            xy = eng.get("XY", (2, NS_DEV))[:, self._sr]
        finally:
            if not default:
                engine_for(self.co, LambdaIR=self.LambdaIR, irf_soffset=1.0, irf_rescale=1.0, irf_window=None)
        return np.ascontiguousarray(xy[0]), np.ascontiguousarray(xy[1])

    def setXpYp(self, bird):
        """[X^(p+1), p < NIR] then [Y X^p, p < NIR] on the resummed s range (reference pybird.py:1402-1407): [2 NIR, |sr|]"""
        X, Y = self.IRFilters(bird)
        return np.concatenate((np.array([X ** (p + 1) for p in range(self.NIR)]), np.array([Y * X**p for p in range(self.NIR)])))

    def extractBAO(self, cf):
        """(reference pybird.py:1382-1400) host form, for callers of the helper; the stage itself uses extract_bao_kernel"""
        if not self.co.optiresum:
            return cf
        from scipy.interpolate import interp1d

        cfnobao = np.concatenate([cf[..., : self.idlow], cf[..., self.idhigh :]], axis=-1)
        nobao = interp1d(self.snobao, self.snobao**2 * cfnobao, kind="linear", axis=-1)(self.sbao) * self.sbao**-2
        return cf[..., self.idlow : self.idhigh] - nobao

    def Ps(self, bird, window=None):
        """Adds the IR corrections to bird.P11l / Pctl / Ploopl in place (reference pybird.py:1413-1464)."""
        self._window = window  # coefficient taper of the resummation FFTLog: part of the H table (rebuilt when it changes)
        co = self.co
        eng = self._inputs(bird)
        for n, buf in (("C11", "C11"), ("Cct", "CCT"), ("Cloopl", "CLOOPL")) + ((("CctNNLO", "CCTN"),) if co.with_NNLO else ()):
            if not bird._on_device(eng, buf):
                eng.put(buf, _s_pad(getattr(bird, n)))
                bird._dev.add(buf)
        bird._templates_to_device(eng)
        eng.run(L.S_RESUM, sync=False)
        bird._templates_pending(("P11l", "Pctl", "Ploopl"))
        self._Q, self._Qf = None, bird.f
        if self.snapshot:
            bird.create_snapshot("IRresum")


# ----------------------------------------------------------------------------- APeffect
class APeffect(HasLogger):
    """Alcock-Paczynski distortion (reference pybird.py:1467-1629) -- spline_kernel, ap_prefix_kernel, ap_weights_kernel, ap_rows_kernel (ap_direct_kernel for extreme distortions)."""

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
        engine_for(co, nbinsmu=self.nbinsmu)

    @property
    def engine(self):
        return engine_for(self.co, nbinsmu=self.nbinsmu)

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
        claim(eng, bird)
        bird.__dict__["_engine"] = eng
        qperp, qpar = self.get_AP_param(bird) if q is None else q
        if getattr(eng, "_ap_state", None) != (self.DA, self.H, bool(self.APst)):  # tables of the stage: only when another plugin set them last
            eng.set_ap_fiducial(self.DA, self.H)
            eng.set_ap_stochastic(self.APst)
            eng._ap_state = (self.DA, self.H, bool(self.APst))
        eng.put("DA", np.array([qperp * self.DA]))
        eng.put("H", np.array([self.H / qpar]))
        bird._templates_to_device(eng)
        eng.run(L.S_AP, sync=False)
        bird._templates_pending(("P11l", "Pctl", "Ploopl", "Pstl") if self.APst else ("P11l", "Pctl", "Ploopl"))
        if self.snapshot:
            bird.create_snapshot("APeffect")

    def integrAP(self, k, Pk, kp, arrayLegendremup, many=False):
        """Host form of the AP integral for one set of rows (reference pybird.py:1581-1596): cubic interpolation onto kp(k, mu),
        Legendre weights, trapezoid rule on the mu grid.  Not on the device path (``AP`` is); kept for callers of the helper."""
        from scipy.interpolate import interp1d

        mugrid = np.linspace(0.0, 1.0, self.nbinsmu)
        _trapz = getattr(np, "trapezoid", None) or np.trapz
        Pkint = interp1d(k, Pk, axis=-1, kind="cubic", bounds_error=False, fill_value="extrapolate")
        if many:
            return 2.0 * _trapz(np.einsum("lpkm,lkm->lpkm", Pkint(kp), arrayLegendremup), x=mugrid, axis=-1)
        return 2.0 * _trapz(Pkint(kp) * arrayLegendremup, x=mugrid, axis=-1)


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
        from .transformer import apply_operator_in_place

        eng = engine_for(bird.co)
        if self._op is None or self._op[0] is not eng:
            full = compose_operator(self.co.Nl, self.co.Nk, fiber=self.operator())
            keep = None if self.fiberst else compose_operator(self.co.Nl, self.co.Nk)  # Pstl untouched unless fiberst (pybird.py:1788-1797)
            self._op = (eng, eng.add_operator(full, stochastic=keep))
        apply_operator_in_place(eng, self._op[1], bird)
        if self.snapshot:
            bird.create_snapshot("fiber")
