"""Extract the reference's formula DATA into ``eftpipe_amd/data/pt_tables.json`` / ``q_tables.npz``.

Build-container tool (needs /root/reference + sympy); never runs on the GPU box.

What is extracted (data, not code):
  * M22b[0..27](n1, n2)  -- reference eftpipe/pybird/pybird.py:119-148 -- as factored rational
    functions: rational prefactor, list of irreducible numerator factors (each a sparse
    polynomial {(i, j): integer coeff}), and the exponents of the three universal denominator
    factors n(1+n)(2n-1) (symmetric in n1, n2).
  * M13b[0..9](n1)       -- pybird.py:98-109 -- same, one variable.
  * IR-resummation Q tables -- pybird.py:179-469 (Nl=2, ``Qa``) and
    eftpipe/pybird/resumfactor.py:4632-4643 (Nl=3, ``Qawithhex``) -- every entry is a polynomial
    in the growth rate f; stored as float64 coefficient arrays [2, Nl, Nl, Nn, 15] (ascending powers).
  * mu-power -> Legendre projection weights (pybird.py:89-95) incl. the 48/148 entry as written.

Every table is verified here against the reference lambdas at random points before it is written.
"""
from __future__ import annotations

import json
import os
import sys
from fractions import Fraction

import numpy as np
import sympy as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from refimport import load_reference  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "eftpipe_amd", "data")


def _poly_terms(expr, syms):
    poly = sp.Poly(sp.expand(expr), *syms)
    out = []
    for mon, c in poly.terms():
        c = sp.Rational(c)
        assert c.q == 1, f"non-integer coefficient {c}"
        out.append([list(map(int, mon)), int(c.p)])
    return out


def _factored(expr, syms):
    """-> dict(scale=[p, q], num=[[terms, mult], ...], den=[[terms, mult], ...])"""
    expr = sp.nsimplify(sp.together(expr), rational=True)
    num, den = sp.fraction(sp.together(expr))
    cn, fn = sp.factor_list(sp.expand(num), *syms)
    cd, fd = sp.factor_list(sp.expand(den), *syms)
    scale = sp.Rational(cn) / sp.Rational(cd)
    return {
        "scale": [int(scale.p), int(scale.q)],
        "num": [[_poly_terms(f, syms), int(m)] for f, m in fn],
        "den": [[_poly_terms(f, syms), int(m)] for f, m in fd],
    }


def _eval_terms(terms, vals):
    tot = 0.0
    for mon, c in terms:
        t = complex(c)
        for v, p in zip(vals, mon):
            t = t * v**p
        tot = tot + t
    return tot


def eval_factored(entry, vals):
    r = entry["scale"][0] / entry["scale"][1]
    for terms, m in entry["num"]:
        r = r * _eval_terms(terms, vals) ** m
    for terms, m in entry["den"]:
        r = r / _eval_terms(terms, vals) ** m
    return r


def main():
    ref = load_reference()
    pb = ref.pybird
    rng = np.random.default_rng(7)
    n1, n2, f = sp.symbols("n1 n2 f")

    tables = {}
    # ---- M22b
    m22 = []
    for b in range(28):
        e = pb.M22b[b](n1, n2)
        e = sp.Integer(e) if isinstance(e, int) else e
        entry = _factored(e, (n1, n2))
        for _ in range(20):
            a = complex(0.8, rng.uniform(-25, 25))
            c = complex(0.8, rng.uniform(-25, 25))
            got, want = eval_factored(entry, (a, c)), complex(pb.M22b[b](a, c))
            assert abs(got - want) <= 1e-12 * abs(want), (b, got, want)
        m22.append(entry)
    tables["M22b"] = m22
    # ---- M13b
    m13 = []
    for b in range(10):
        e = pb.M13b[b](n1)
        e = sp.nsimplify(e, rational=True) if not isinstance(e, sp.Basic) else e
        entry = _factored(e, (n1,))
        for _ in range(20):
            a = complex(0.8, rng.uniform(-25, 25))
            got, want = eval_factored(entry, (a,)), complex(pb.M13b[b](a))
            assert abs(got - want) <= 1e-13 * abs(want), (b, got, want)
        m13.append(entry)
    tables["M13b"] = m13
    # ---- mu -> Legendre weights, exactly as written in the reference (floats)
    tables["mu"] = {str(p): [pb.mu[p][l] for l in (0, 2, 4)] for p in (0, 2, 4, 6, 8)}
    # ---- native grids (pybird.py:472-482)
    tables["kbird"] = pb.get_kbird(0.3).tolist()
    tables["sbird"] = pb.sbird.tolist()

    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, "pt_tables.json"), "w") as fh:
        json.dump(tables, fh, separators=(",", ":"))

    # ---- Q tables -> polynomial coefficients in f
    def qcoef(table, Nl, Nn):
        out = np.zeros((2, Nl, Nl, Nn, 15))
        for a in range(2):
            for l in range(Nl):
                for lp in range(Nl):
                    for u in range(Nn):
                        e = table[a][2 * l][2 * lp][u](f)
                        if isinstance(e, (int, float)):
                            out[a, l, lp, u, 0] = float(e)
                            continue
                        poly = sp.Poly(sp.expand(sp.nsimplify(e, rational=True)), f)
                        co = poly.all_coeffs()[::-1]
                        assert len(co) <= 15, len(co)
                        for i, c in enumerate(co):
                            out[a, l, lp, u, i] = float(Fraction(int(sp.Rational(c).p), int(sp.Rational(c).q)))
        # verify
        worst = 0.0
        for fv in (0.0, 0.3, 0.55, 0.8, 1.0):
            for a in range(2):
                for l in range(Nl):
                    for lp in range(Nl):
                        for u in range(Nn):
                            want = float(table[a][2 * l][2 * lp][u](fv))
                            got = float(np.polynomial.polynomial.polyval(fv, out[a, l, lp, u]))
                            if want != 0:
                                worst = max(worst, abs(got - want) / abs(want))
                            else:
                                assert got == 0.0
        return out, worst

    q2, w2 = qcoef(pb.Qa, 2, 32)
    q3, w3 = qcoef(ref.resumfactor.Qawithhex, 3, 96)
    print("Q-table polynomial round-trip: Nl=2 %.2e, Nl=3 %.2e" % (w2, w3))
    assert w2 < 1e-13 and w3 < 1e-13
    np.savez_compressed(os.path.join(OUT_DIR, "q_tables.npz"), Qa_Nl2=q2, Qa_Nl3=q3)
    print("wrote", OUT_DIR)


if __name__ == "__main__":
    main()
