"""Bias contraction onto P_l(k): host mirror of reference eftpipe/parambasis.py:42-136 and of its two parameter
bases (WestCoastBasis :166-316, EastCoastBasis :320-454).

``bias_vectors`` builds the 3 + 6 + 12 + 3 coefficient vectors; the
contraction itself runs on the device (``reduce_kernel``) when driven through ``Engine`` and is a
24-term dot product per (l, k).  ``reduce_Plk`` keeps the reference signature for BirdLike objects.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class BirdComponent:
    """Same record as reference parambasis.py:30-39."""

    Plin: np.ndarray
    Ploop: np.ndarray
    Pct: np.ndarray
    Pst: np.ndarray
    Picc: np.ndarray

    def sum(self):
        return self.Plin + self.Ploop + self.Pct + self.Pst + self.Picc


def bias_vectors(f, bsA, bsB=None, es=(0.0, 0.0, 0.0), kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None, ndB=None,
                 counterform="westcoast"):
    """-> b11[3], bct[6], bloop[12], bst[3] (reference parambasis.py:69-126; counterform 'eastcoast' reads the last
    three entries of bsA/bsB as ctilde_0, ctilde_2, ctilde_4, :93-102)."""
    kmB = kmA if kmB is None else kmB
    krB = krA if krB is None else krB
    ndB = ndA if ndB is None else ndB
    b1A, b2A, b3A, b4A, cctA, cr1A, cr2A = bsA
    b1B, b2B, b3B, b4B, cctB, cr1B, cr2B = bsB if bsB is not None else bsA
    ce0, cemono, cequad = es
    b11 = np.array([b1A * b1B, (b1A + b1B) * f, f * f])
    if counterform not in ("westcoast", "eastcoast"):
        raise ValueError(f"unexpected counterform: {counterform}")
    bct = np.array([-cctA - cctB, -(cr1A + cr1B) * f, -(cr2A + cr2B) * f**2, 0.0, 0.0, 0.0]) if counterform == "eastcoast" else np.array([
        b1A * cctB / kmB**2 + b1B * cctA / kmA**2,
        b1B * cr1A / krA**2 + b1A * cr1B / krB**2,
        b1B * cr2A / krA**2 + b1A * cr2B / krB**2,
        (cctA / kmA**2 + cctB / kmB**2) * f,
        (cr1A / krA**2 + cr1B / krB**2) * f,
        (cr2A / krA**2 + cr2B / krB**2) * f,
    ])
    bloop = np.array([
        1.0, 0.5 * (b1A + b1B), 0.5 * (b2A + b2B), 0.5 * (b3A + b3B), 0.5 * (b4A + b4B), b1A * b1B,
        0.5 * (b1A * b2B + b1B * b2A), 0.5 * (b1A * b3B + b1B * b3A), 0.5 * (b1A * b4B + b1B * b4A),
        b2A * b2B, 0.5 * (b2A * b4B + b2B * b4A), b4A * b4B,
    ])
    x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
    x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
    bst = np.array([ce0 * x1, cemono * x2, cequad * x2])
    return b11, bct, bloop, bst


def bias_row(f, bsA, bsB=None, es=(0.0, 0.0, 0.0), **scales):
    """The 24 coefficients in template-row order (P11l, Pctl, Ploopl, Pstl) for the device reduce."""
    return np.concatenate(bias_vectors(f, bsA, bsB, es, **scales))


def nnlo_vector(f, b1A, cnnloA, krA=0.25, counterform="westcoast"):
    """bctNNLOAB[3] (reference parambasis.py:96-106): west coast cnnloA = (cr4, cr6); east coast (ctilde, _).
    These are the EFTB_B_BIASN coefficients of the device reduce."""
    if counterform == "westcoast":
        cr4, cr6 = cnnloA
        return np.array([1 / 4 * b1A**2 / krA**4 * cr4, 1 / 4 * b1A / krA**4 * cr6, 0.0])
    ctilde = cnnloA[0]
    return ctilde * np.array([-(b1A**2) * f**4, -2 * b1A * f**5, -(f**6)])


def reduce_Plk(bird, bsA, bsB=None, es=(0.0, 0.0, 0.0), cnnloA=(0.0, 0.0), cnnloB=None):
    """BirdLike -> BirdComponent (reference parambasis.py:42-136; the counter-term form is bird.co.counterform; with
    bird.co.with_NNLO the k^4 counter-terms bctNNLO . PctNNLOl are added to Pct)."""
    co = bird.co
    b11, bct, bloop, bst = bias_vectors(bird.f, list(bsA), None if bsB is None else list(bsB), tuple(es),
                                        kmA=co.kmA, krA=co.krA, ndA=co.ndA, kmB=co.kmB, krB=co.krB, ndB=co.ndB,
                                        counterform=getattr(co, "counterform", "westcoast"))
    No = co.No
    return BirdComponent(
        Plin=np.einsum("b,lbx->lx", b11, bird.P11l[:No]),
        Ploop=np.einsum("b,lbx->lx", bloop, bird.Ploopl[:No]),
        Pct=np.einsum("b,lbx->lx", bct, bird.Pctl[:No])
        + (np.einsum("b,lbx->lx", nnlo_vector(bird.f, list(bsA)[0], tuple(cnnloA), co.krA, getattr(co, "counterform", "westcoast")),
                     bird.PctNNLOl[:No]) if getattr(co, "with_NNLO", False) else 0.0),
        Pst=np.einsum("b,lbx->lx", bst, bird.Pstl[:No]),
        Picc=bird.Picc[:No],
    )


# ----------------------------------------------------------------------------- Gaussian table
GAUSSIAN = ("b3", "cct", "cr1", "cr2")
STOCHASTIC = ("ce0", "cemono", "cequad")


def gaussian_params(prefix="", cross_prefix=()):
    """Names of the analytically marginalisable parameters, in table order (reference parambasis.py:209-223)."""
    if cross_prefix:
        return [x + p for x in cross_prefix for p in GAUSSIAN] + [prefix + p for p in STOCHASTIC]
    return [prefix + p for p in GAUSSIAN + STOCHASTIC]


def gaussian_rows(f, ngA, ngB=None, kmA=0.7, krA=0.25, ndA=3e-4, kmB=None, krB=None, ndB=None, basis="westcoast"):
    """Coefficient rows over the 24 template rows (P11l[3], Pctl[6], Ploopl[12], Pstl[3]) of

        row 0        P_NG: the model with every Gaussian parameter at 0 (reference likelihood.py:524-549 via reduce_Plk)
        rows 1..nG   dP/d(gaussian parameter) in the order of ``gaussian_params`` (reference parambasis.py:249-316)

    for the non-Gaussian parameters ngA = (b1, b2, b4) [and ngB for a cross spectrum].  Everything downstream of the
    templates is linear in these rows, so the device builds P_NG and P_G with one small contraction per walker.
    basis="eastcoast": ngA = (b1, b2, bG2), rows in the order of EastCoastBasis.gaussian_params (reference :413-444)."""
    if basis == "eastcoast":
        if ngB is not None:
            raise NotImplementedError("EastCoastBasis does not support cross yet")
        return _eastcoast_rows(f, ngA, kmA, krA, ndA)
    if basis != "westcoast":
        raise ValueError(f"unexpected basis: {basis}")
    cross = ngB is not None
    kmB = kmA if kmB is None else kmB
    krB = krA if krB is None else krB
    ndB = ndA if ndB is None else ndB
    b1A, b2A, b4A = ngA
    b1B, b2B, b4B = ngB if cross else ngA
    bsA = [b1A, b2A, 0.0, b4A, 0.0, 0.0, 0.0]
    bsB = [b1B, b2B, 0.0, b4B, 0.0, 0.0, 0.0] if cross else None
    rows = [bias_row(f, bsA, bsB, (0.0, 0.0, 0.0), kmA=kmA, krA=krA, ndA=ndA, kmB=kmB, krB=krB, ndB=ndB)]

    def row(pairs):
        r = np.zeros(24)
        for i, c in pairs:
            r[i] = c
        return r

    if cross:
        for b1o, km, kr in ((b1B, kmA, krA), (b1A, kmB, krB)):
            rows.append(row([(LOOP + 3, 0.5), (LOOP + 7, 0.5 * b1o)]))
            rows.append(row([(CT + 0, b1o / km**2), (CT + 3, f / km**2)]))
            rows.append(row([(CT + 1, b1o / kr**2), (CT + 4, f / kr**2)]))
            rows.append(row([(CT + 2, b1o / kr**2), (CT + 5, f / kr**2)]))
    else:
        rows.append(row([(LOOP + 3, 1.0), (LOOP + 7, b1A)]))
        rows.append(row([(CT + 0, 2.0 * b1A / kmA**2), (CT + 3, 2.0 * f / kmA**2)]))
        rows.append(row([(CT + 1, 2.0 * b1A / krA**2), (CT + 4, 2.0 * f / krA**2)]))
        rows.append(row([(CT + 2, 2.0 * b1A / krA**2), (CT + 5, 2.0 * f / krA**2)]))
    x1 = 0.5 * (1.0 / ndA + 1.0 / ndB)
    x2 = 0.5 * (1.0 / ndA / kmA**2 + 1.0 / ndB / kmB**2)
    rows += [row([(ST + 0, x1)]), row([(ST + 1, x2)]), row([(ST + 2, x2)])]
    return np.stack(rows)


# ----------------------------------------------------------------------------- parameter bases
CT, LOOP, ST = 3, 9, 21  # first template row of Pctl, Ploopl, Pstl in the 24-row order


def eastcoast_to_bs(f, b1, b2, bG2, bGamma3, c0, c2, c4, Pshot=0.0, a0=0.0, a2=0.0):
    """East-coast parameters -> (bsA, es) of reduce_Plk with counterform='eastcoast' (reference parambasis.py:378-397)."""
    bsA = [b1, b1 + 7 / 2 * bG2, b1 + 15 * bG2 + 6 * bGamma3, 1 / 2 * b2 - 7 / 2 * bG2,
           c0 - f / 3 * c2 + 3 / 35 * f**2 * c4, c2 - 6 / 7 * f * c4, c4]
    es = [Pshot, a0 + 1 / 3 * a2, 2 / 3 * a2]
    return bsA, es


def eastcoast_bias_row(f, b1, b2, bG2, bGamma3=0.0, c0=0.0, c2=0.0, c4=0.0, Pshot=0.0, a0=0.0, a2=0.0, **scales):
    """The 24 device-reduce coefficients for east-coast parameters (arxiv 2106.12580 convention of the reference)."""
    bsA, es = eastcoast_to_bs(f, b1, b2, bG2, bGamma3, c0, c2, c4, Pshot, a0, a2)
    return bias_row(f, bsA, None, es, counterform="eastcoast", **scales)


def _eastcoast_rows(f, ng, kmA, krA, ndA):
    b1, b2, bG2 = ng
    rows = [eastcoast_bias_row(f, b1, b2, bG2, kmA=kmA, krA=krA, ndA=ndA)]

    def row(pairs):
        r = np.zeros(24)
        for i, c in pairs:
            r[i] = c
        return r

    rows.append(row([(LOOP + 3, 6.0), (LOOP + 7, 6.0 * b1)]))                                      # bGamma3
    rows.append(row([(CT + 0, -2.0)]))                                                             # c0
    rows.append(row([(CT + 0, 2 / 3 * f), (CT + 1, -2.0 * f)]))                                    # c2
    rows.append(row([(CT + 0, -6 / 35 * f**2), (CT + 1, 12 / 7 * f**2), (CT + 2, -2.0 * f**2)]))   # c4
    x1 = 1.0 / ndA
    x2 = 1.0 / ndA / kmA**2
    rows.append(row([(ST + 0, x1)]))                                                               # Pshot
    rows.append(row([(ST + 1, x2)]))                                                               # a0
    rows.append(row([(ST + 1, x2 / 3), (ST + 2, 2 * x2 / 3)]))                                     # a2
    return np.stack(rows)


def _table_from_rows(bird, rows, names, requires):
    No = bird.co.No
    T = np.concatenate([bird.P11l[:No], bird.Pctl[:No], bird.Ploopl[:No], bird.Pstl[:No]], axis=1)  # [No, 24, nx]
    return {p: np.einsum("b,lbx->lx", r, T) for p, r in zip(names, rows) if requires is None or p in requires}


@dataclass(frozen=True)
class WestCoastBasis:
    """Same surface as reference parambasis.py:166-316 (b1 b2 b3 b4 cct cr1 cr2 | ce0 cemono cequad)."""

    prefix: str = ""
    cross_prefix: list = field(default_factory=list)

    def default(self):
        return {p: 0.0 for p in self.gaussian_params()}

    def bsA(self):
        prefix = self.cross_prefix[0] if self.is_cross() else self.prefix
        return [prefix + p for p in ("b1", "b2", "b3", "b4", "cct", "cr1", "cr2")]

    def bsB(self):
        return [self.cross_prefix[1] + p for p in ("b1", "b2", "b3", "b4", "cct", "cr1", "cr2")] if self.is_cross() else []

    def es(self):
        return [self.prefix + p for p in STOCHASTIC]

    def cnnloA(self):
        return [self.prefix + p for p in ("cr4", "cr6")]

    def is_cross(self):
        return bool(self.cross_prefix)

    @classmethod
    def get_name(cls):
        return "westcoast"

    @classmethod
    def counterform(cls):
        return "westcoast"

    def non_gaussian_params(self):
        names = ("b1", "b2", "b4")
        if self.is_cross():
            return [x + p for x in self.cross_prefix for p in names]
        return [self.prefix + p for p in names]

    def gaussian_params(self):
        if self.is_cross():
            return gaussian_params(self.prefix, self.cross_prefix)
        return gaussian_params(self.prefix) + self.cnnloA()  # the NNLO names never enter the table (as the reference)

    def reduce_Plk(self, bird, params_values_dict):
        v = self.default()
        v.update(params_values_dict)
        cnnloA = [v[p] for p in self.cnnloA()] if getattr(bird.co, "with_NNLO", False) else (0.0, 0.0)
        return reduce_Plk(bird, [v[p] for p in self.bsA()], [v[p] for p in self.bsB()] or None, [v[p] for p in self.es()], cnnloA)

    def nnlo_row(self, f, params_values_dict, krA=0.25):
        """EFTB_B_BIASN coefficients (cr4, cr6) for one walker"""
        v = self.default()
        v.update(params_values_dict)
        return nnlo_vector(f, v[self.bsA()[0]], [v[p] for p in self.cnnloA()], krA, "westcoast")

    def bias_row(self, f, params_values_dict, **scales):
        """Device-reduce coefficients for one walker (the batched counterpart of reduce_Plk)."""
        v = self.default()
        v.update(params_values_dict)
        return bias_row(f, [v[p] for p in self.bsA()], [v[p] for p in self.bsB()] or None, [v[p] for p in self.es()], **scales)

    def gaussian_rows(self, f, params_values_dict, **scales):
        ng = [[params_values_dict[x + p] for p in ("b1", "b2", "b4")] for x in (self.cross_prefix or [self.prefix])]
        return gaussian_rows(f, ng[0], ng[1] if self.is_cross() else None, **scales)

    def reduce_Plk_gaussian_table(self, bird, params_values_dict, requires=None):
        co = bird.co
        rows = self.gaussian_rows(bird.f, params_values_dict, kmA=co.kmA, krA=co.krA, ndA=co.ndA, kmB=co.kmB, krB=co.krB, ndB=co.ndB)
        PG = _table_from_rows(bird, rows[1:], gaussian_params(self.prefix, self.cross_prefix), requires)
        if getattr(co, "with_NNLO", False) and not self.is_cross():  # reference parambasis.py:303-307
            b1, No, st = params_values_dict[self.prefix + "b1"], co.No, PG.pop(self.prefix + "ce0", None)
            extra = {self.prefix + "cr4": 1 / 4 * b1**2 / co.krA**4 * bird.PctNNLOl[:No, 0], self.prefix + "cr6": 1 / 4 * b1 / co.krA**4 * bird.PctNNLOl[:No, 1]}
            PG.update({p: v for p, v in extra.items() if requires is None or p in requires})
            if st is not None:  # keep the reference's insertion order: ..., cr4, cr6, ce0, cemono, cequad
                rest = {p: PG.pop(p) for p in (self.prefix + "cemono", self.prefix + "cequad") if p in PG}
                PG[self.prefix + "ce0"] = st
                PG.update(rest)
        return PG


@dataclass(frozen=True)
class EastCoastBasis:
    """Same surface as reference parambasis.py:320-454 (b1 b2 bG2 bGamma3 c0 c2 c4 | Pshot a0 a2; auto spectra only)."""

    prefix: str = ""
    cross_prefix: list = field(default_factory=list)

    def __post_init__(self):
        if self.cross_prefix:
            raise NotImplementedError("EastCoastBasis does not support cross yet")

    def default(self):
        return {p: 0.0 for p in self.gaussian_params()}

    def bsA(self):
        return [self.prefix + p for p in ("b1", "b2", "bG2", "bGamma3", "c0", "c2", "c4")]

    def es(self):
        return [self.prefix + p for p in ("Pshot", "a0", "a2")]

    def cnnloA(self):
        return [self.prefix + "ctilde"]

    def is_cross(self):
        return False

    @classmethod
    def get_name(cls):
        return "eastcoast"

    @classmethod
    def counterform(cls):
        return "eastcoast"

    def non_gaussian_params(self):
        return [self.prefix + p for p in ("b1", "b2", "bG2")]

    def gaussian_params(self):
        return [self.prefix + p for p in ("bGamma3", "c0", "c2", "c4", "Pshot", "a0", "a2")] + self.cnnloA()

    def _values(self, params_values_dict):
        v = self.default()
        v.update(params_values_dict)
        return [v[p] for p in self.bsA() + self.es()]

    def reduce_Plk(self, bird, params_values_dict):
        bsA, es = eastcoast_to_bs(bird.f, *self._values(params_values_dict))
        v = self.default()
        v.update(params_values_dict)
        cnnloA = [v[self.prefix + "ctilde"], 0.0] if getattr(bird.co, "with_NNLO", False) else (0.0, 0.0)
        return reduce_Plk(bird, bsA, None, es, cnnloA)  # bird.co.counterform must be 'eastcoast', as in the reference

    def nnlo_row(self, f, params_values_dict, krA=0.25):
        """EFTB_B_BIASN coefficients (ctilde) for one walker"""
        v = self.default()
        v.update(params_values_dict)
        return nnlo_vector(f, v[self.prefix + "b1"], [v[self.prefix + "ctilde"], 0.0], krA, "eastcoast")

    def bias_row(self, f, params_values_dict, **scales):
        return eastcoast_bias_row(f, *self._values(params_values_dict), **scales)

    def gaussian_rows(self, f, params_values_dict, **scales):
        return gaussian_rows(f, [params_values_dict[self.prefix + p] for p in ("b1", "b2", "bG2")], basis="eastcoast", **scales)

    def reduce_Plk_gaussian_table(self, bird, params_values_dict, requires=None):
        co = bird.co
        rows = self.gaussian_rows(bird.f, params_values_dict, kmA=co.kmA, krA=co.krA, ndA=co.ndA)
        PG = _table_from_rows(bird, rows[1:], self.gaussian_params()[:7], requires)
        p = self.prefix + "ctilde"
        if getattr(co, "with_NNLO", False) and (requires is None or p in requires):  # reference parambasis.py:429-435
            b1, f, Pn = params_values_dict[self.prefix + "b1"], bird.f, bird.PctNNLOl[: co.No]
            ct = -(b1**2) * f**4 * Pn[:, 0] - 2.0 * b1 * f**5 * Pn[:, 1] - f**6 * Pn[:, 2]
            keys = list(PG)
            st = {q: PG.pop(q) for q in keys if q in (self.prefix + "Pshot", self.prefix + "a0", self.prefix + "a2")}
            PG[p] = ct  # reference order: bGamma3, c0, c2, c4, ctilde, Pshot, a0, a2
            PG.update(st)
        return PG


def find_param_basis(name):
    """reference parambasis.py:457-466"""
    if name == "westcoast":
        return WestCoastBasis
    if name == "eastcoast":
        return EastCoastBasis
    import importlib

    module_name, class_name = name.rsplit(".", 1)
    return getattr(importlib.import_module(module_name), class_name)
