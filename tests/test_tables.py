"""Host logic: the real-reduced / operator-folded tables reproduce the oracle (CPU only)."""
import numpy as np
import pytest

import emulate as E
from conftest import relerr
from eftpipe_amd import synth
from eftpipe_amd.tables import EngineConfig, build_tables
from oracle_util import oracle_engine


def _tables(g, resum, ap, **kw):
    from oracle.engine import da_func, hubble

    native = g["k"].size == 50
    cfg = EngineConfig(Nl=int(g["Nl"]), k=None if native else g["k"], with_resum=resum, with_ap=ap,
                       DA_AP=da_func(synth.OM_AP, float(g["z"])), H_AP=hubble(synth.OM_AP, float(g["z"])), **kw)
    return build_tables(cfg)


@pytest.mark.parametrize("name,resum,ap", [("caseA", False, False), ("caseE", True, True), ("caseC", True, True)])
def test_device_algebra_matches_oracle(golden, name, resum, ap):
    g = golden(name)
    t = _tables(g, resum, ap)
    f = float(g["f"])
    eng = oracle_engine(g, name, window_file=None, kout=None)
    taps = {}
    eng.evaluate(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), taps=taps)
    st = E.pscf(t, g["Pin"], with_cf=resum)
    for n in ("P11", "P22", "P13") + (("C11", "Cct", "C22", "C13") if resum else ()):
        assert relerr(st[n], taps["pscf"][n]) < 1e-9, n
    st.update(E.setpscfl(t, f, st, with_cf=resum))
    for n in ("P11l", "Pctl", "Ploopl", "Pstl") + (("Cloopl",) if resum else ()):
        assert relerr(st[n], taps["setpscfl"][n]) < 1e-9, n
    if resum:
        st = E.resum(t, f, g["Pin"], st)
        assert relerr(st["X"], taps["resum"]["X"]) < 1e-10
        assert relerr(st["Y"], taps["resum"]["Y"]) < 1e-10
        for n in ("P11l", "Pctl", "Ploopl"):
            assert relerr(st[n], taps["resum"][n]) < 1e-9, n
        if "rs_rows" in t:  # matrix-core form of the stage
            pre = {n: taps["setpscfl"][n] for n in ("P11l", "Pctl", "Ploopl", "Cloopl")}
            pre.update(C11=taps["pscf"]["C11"], Cct=taps["pscf"]["Cct"])
            alt = E.resum_mfma(t, f, g["Pin"], pre)
            for n in ("P11l", "Pctl", "Ploopl"):
                assert relerr(alt[n], taps["resum"][n]) < 1e-9, n
            z = t["k"][:, None] ** 2 * alt["X"][None, :]
            assert z.max() < 2 * 8.0  # the scaled variable of the polynomial basis stays O(1)
            if t["l11"].shape[0] == 3:
                # direct-P_l runs: the bias contraction commutes with the stage -- the contracted correction (nine scalar-coefficient polynomials per s)
                # against the reference's template corrections contracted afterwards
                rng = np.random.default_rng(3)
                bias = rng.normal(size=24)
                dref = sum(np.einsum("i,lik->lk", bias[sl], taps["resum"][n] - taps["setpscfl"][n])
                           for n, sl in (("P11l", slice(0, 3)), ("Pctl", slice(3, 9)), ("Ploopl", slice(9, 21))))
                assert relerr(E.resum_plk(t, f, g["Pin"], pre, bias), dref) < 1e-9
    if ap:
        names = ("P11l", "Pctl", "Ploopl") + (("Pstl",) if name == "caseC" else ())
        st = E.ap(t, float(g["DA"]), float(g["H"]), st, names)
        for n in names:
            assert relerr(st[n], taps["ap"][n]) < 1e-9, n
            assert relerr(st[n], g["ap_" + n]) < 1e-9, n


def test_banded_spline_operator_matches_scipy():
    from scipy.interpolate import CubicSpline

    from eftpipe_amd.tables import spline_derivative_band

    for k in (synth.survey_kgrid(256), synth.survey_kgrid(2048), np.array(__import__("eftpipe_amd.loopmath", fromlist=["x"]).native_k())):
        rng = np.random.default_rng(0)
        y = rng.normal(size=(5, k.size)).cumsum(axis=-1)
        t = dict(k=k, sp_band=spline_derivative_band(k))
        sd, slope = E.spline_derivs(t, y)
        cs = CubicSpline(k, y, axis=-1)
        assert np.max(np.abs(sd - cs(k, 1))) < 1e-10 * np.max(np.abs(cs(k, 1)))
        xe = np.concatenate([[0.0005], rng.uniform(0.001, 0.3, 400), [0.31]])
        assert relerr(E.spline_eval(t, y, sd, slope, xe), cs(xe)) < 1e-11


def test_projection_operators_match_reference(golden):
    """Host-folded window / binning / chained operators applied with NumPy == the reference's own outputs."""
    import os

    from eftpipe_amd import tables as TB

    g = golden("caseC")
    k = g["k"]
    tab = np.load(os.path.join(os.path.dirname(__file__), "golden", "win_NGC_LRG_sQ024.npy"))
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    assert np.array_equal(p, g["window_p"])
    Wfold, Waldk = TB.window_fold(k, Wal, p)
    assert relerr(Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
    assert relerr(Waldk.sum(axis=-1), g["window_Waldk_sum_p"]) < 1e-9
    names = ("P11l", "Pctl", "Ploopl", "Pstl")
    win = {n: np.einsum("alxk,lrk->arx", Wfold, g["ap_" + n]) for n in names}
    for n in names:
        assert relerr(win[n], g["window_" + n]) < 1e-9, n
    B, keff, _, _ = TB.binning_operator(k, g["kout"])
    assert np.allclose(keff, g["keff"], rtol=1e-13)
    for n in names:
        assert relerr(np.einsum("bk,lrk->lrb", B, win[n]), g["binned_" + n]) < 1e-9, n
    full = TB.compose_operator(3, k.size, Wfold=Wfold, binning=B, chained=True)
    assert full.shape == (2, 3, len(g["kout"]), k.size)
    for n in names:
        assert relerr(np.einsum("alxk,lrk->arx", full, g["ap_" + n]), g["chained_" + n]) < 1e-9, n


def test_surface_options_select_operator_tables(golden):
    """VERDICT r02 item 8 on the CPU: PsCf(window=0.3), a non-default input grid and IRFilters(soffset, LambdaIR, RescaleIR, window) are other
    operator TABLES of the same kernels -- the NumPy emulation of the device contractions on those tables against the REAL reference
    (tests/golden/surface.npz, tools/make_fixtures.py surface; reference pybird.py:682-695, 1143-1171, 1316-1353)."""
    g = golden("surface")
    f = float(g["f"])
    # (1) coefficient window 0.3
    t = build_tables(EngineConfig(Nl=3, with_resum=True, fft_window=0.3))
    st = E.pscf(t, g["Pin"], with_cf=True)
    for n in ("P11", "P22", "P13", "C11", "Cct", "C22", "C13"):
        assert relerr(st[n], g["w03_" + n]) < 1e-9, n
    # (2) IR filters with every argument off its default
    so, lam, resc, win = (float(x) for x in g["irf"])
    t = build_tables(EngineConfig(Nl=3, with_resum=True, LambdaIR=lam, irf_soffset=so, irf_rescale=resc, irf_window=win))
    X, Y = E.ir_filters(t, g["Pin"])
    assert relerr(X[None], g["irf_X"][None]) < 1e-12 and relerr(Y[None], g["irf_Y"][None]) < 1e-12
    t0 = build_tables(EngineConfig(Nl=3, with_resum=True))
    X, Y = E.ir_filters(t0, g["Pin"])
    assert relerr(X[None], g["irf_X_default"][None]) < 1e-12 and relerr(Y[None], g["irf_Y_default"][None]) < 1e-12
    # (3) another input grid: 240 samples up to k = 10^0.2
    t = build_tables(EngineConfig(Nl=3, with_resum=True, kin=g["kin2"]))
    st = E.pscf(t, g["Pin2"], with_cf=True)
    for n in ("P11", "P22", "P13", "C11"):
        assert relerr(st[n], g["kin2_" + n]) < 1e-9, n
    st.update(E.setpscfl(t, f, st, with_cf=True))
    st = E.resum(t, f, g["Pin2"], st)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(st[n], g["kin2_resum_" + n]) < 1e-9, n


def test_bspline_tables_reproduce_the_not_a_knot_spline():
    """The AP stage's spline data in B-spline form (round 3; tables.bspline_tables): the banded coefficient operator (what spline_kernel applies
    on the fast path) and the per-interval pieces (what ap_weights_kernel / ap_direct_kernel combine the interval moments / basis values with)
    together reproduce scipy's not-a-knot CubicSpline, inside the grid and on the extrapolated end pieces (reference pybird.py:1581-1621 uses
    interp1d(kind="cubic", fill_value="extrapolate"))."""
    from scipy.interpolate import CubicSpline

    from eftpipe_amd.loopmath import native_k
    from eftpipe_amd.tables import SPL_HB, bspline_tables

    for k in (synth.survey_kgrid(512), np.array(native_k()), synth.survey_kgrid(2048)):
        cband, local, J = bspline_tables(k)
        n = k.size
        assert cband.shape == (2 * SPL_HB + 1, n) and local.shape == (n, 4, 4) and np.array_equal(J, np.clip(np.arange(n) - 1, 0, n - 4))
        rng = np.random.default_rng(3)
        y = rng.normal(size=n).cumsum()
        c = np.zeros(n)
        for d in range(2 * SPL_HB + 1):
            j = np.arange(n) + d - SPL_HB
            ok = (j >= 0) & (j < n)
            c[ok] += cband[d, ok] * y[j[ok]]
        cs = CubicSpline(k, y)
        xe = np.concatenate([[0.5 * k[0]], rng.uniform(k[0], k[-1], 3000), k, [1.03 * k[-1]]])
        i = np.clip(np.searchsorted(k, xe, "right") - 1, 0, n - 2)
        t = xe - k[i]
        val = sum(c[J[i] + e] * (local[i, e, 0] + t * (local[i, e, 1] + t * (local[i, e, 2] + t * local[i, e, 3]))) for e in range(4))
        assert np.max(np.abs(val - cs(xe))) < 1e-12 * np.max(np.abs(cs(xe))), n
