"""Pin the CPU oracle against outputs of the REAL reference (tests/golden, tools/make_fixtures.py)
and against the reference's own known answers.  CPU only."""
import numpy as np
import pytest

from conftest import relerr
from oracle import tables as T
from oracle.engine import OracleEngine, da_func, hubble
from oracle.fftlog import FFTLogGrid
from oracle_util import oracle_engine

TOL = 1e-10  # oracle vs reference: same algorithm, same libraries -> rounding-level agreement


def test_known_answers(golden):
    """reference tests/test_pybird.py:5-11"""
    g = golden("tables")
    assert np.isclose(hubble(0.2, 1.0), 1.549193338482967, atol=0.0)
    assert np.isclose(da_func(0.2, 1.0), 0.4117451980802465, atol=0.0)
    assert np.isclose(hubble(0.2, 1.0), g["known_Hubble_0.2_1.0"][1], rtol=1e-15)
    assert np.isclose(da_func(0.2, 1.0), g["known_DAfunc_0.2_1.0"][1], rtol=1e-14)
    assert np.allclose([OracleEngine.chain_coeff(l) for l in (0, 2, 4)], g["chain_coeff"], rtol=1e-15)


def test_fftlog_reference_vector(golden):
    """reference tests/compare/test_fftlog.py:5-23 (vectorised == per-row, and == reference output)"""
    g = golden("tables")
    fft = FFTLogGrid(256, 10**-5, 10, -0.3)
    c = fft.coef(g["fftlog_gauss_k"], g["fftlog_gauss_p"], extrap="padding", window=0.3)
    assert relerr(c, g["fftlog_gauss_coef"]) < 1e-13
    cb = fft.coef(g["fftlog_gauss_k"], np.vstack([g["fftlog_gauss_p"], 2 * g["fftlog_gauss_p"]]), extrap="padding", window=0.3)
    assert relerr(cb, g["fftlog_gauss_coef_batch"]) < 1e-13
    np.testing.assert_allclose(cb[0], c, rtol=1e-6, atol=0.0)


def test_constant_tables(golden):
    g = golden("tables")
    fft = FFTLogGrid(256, 1.5e-5, 1000.0, -1.6)
    assert np.max(np.abs(fft.Pow - g["Pow"])) == 0.0
    nu = -0.5 * fft.Pow
    idx = g["M22_idx"]
    got = np.array([T.m22a(nu[n], nu[m]) * T.m22b(b, nu[n], nu[m]) for b, n, m in idx])
    assert np.max(np.abs(got - g["M22_val"]) / np.abs(g["M22_val"])) < 1e-12
    m13 = np.stack([T.m13a(nu) * T.m13b(b, nu) for b in range(10)])
    assert np.max(np.abs(m13 - g["M13"]) / np.abs(g["M13"])) < 1e-12
    lidx = g["Ml_idx"]
    ml = np.array([T.mpc(2 * l, nu[n] + nu[m] - 1.5) for l, n, m in lidx])
    assert np.max(np.abs(ml - g["Ml_val"]) / np.abs(g["Ml_val"])) < 1e-12
    for Nl in (2, 3):
        w = T.mu_weights(Nl)
        for n in ("l11", "lct", "l22", "l13"):
            assert np.array_equal(w[n], g[f"{n}_Nl{Nl}"])
        for i, f in enumerate(g["Q_fvals"]):
            q = T.q_matrix(f, Nl)
            assert q.shape == g[f"Q_f_Nl{Nl}"][i].shape
            assert relerr(q.reshape(-1, q.shape[-1]), g[f"Q_f_Nl{Nl}"][i].reshape(-1, q.shape[-1])) < 1e-13


@pytest.mark.parametrize("name", ["caseA", "caseB", "caseC", "caseD", "caseE"])
def test_stages(golden, name):
    g = golden(name)
    eng = oracle_engine(g, name)
    assert np.array_equal(eng.k, g["k"])
    taps = {}
    f = float(g["f"])
    st = eng.evaluate(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), taps=taps)
    assert relerr(taps["pscf"]["coef"], g["coef"]) < 1e-13
    for n in ("P11", "P22", "P13", "C11", "Cct", "C22", "C13"):
        assert relerr(taps["pscf"][n], g["pscf_" + n]) < TOL, n
    for n in ("P11l", "Pctl", "Ploopl", "Cloopl", "Pstl"):
        assert relerr(taps["setpscfl"][n], g["setpscfl_" + n]) < TOL, n
    if "resum_X" in g:
        assert relerr(taps["resum"]["X"], g["resum_X"]) < TOL
        assert relerr(taps["resum"]["Y"], g["resum_Y"]) < TOL
        for n in ("P11l", "Pctl", "Ploopl"):
            assert relerr(taps["resum"][n], g["resum_" + n]) < TOL, n
    if "ap_P11l" in g:
        assert np.isclose(eng.DA_fid, g["DA_AP"], rtol=1e-14) and np.isclose(eng.H_fid, g["H_AP"], rtol=1e-15)
        for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
            assert relerr(taps["ap"][n], g["ap_" + n]) < TOL, n
    if "window_P11l" in g:
        assert np.array_equal(eng.p, g["window_p"])
        assert relerr(eng.Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
        assert relerr(eng.Waldk.sum(axis=-1), g["window_Waldk_sum_p"]) < 1e-9
        for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
            assert relerr(taps["window"][n], g["window_" + n]) < 1e-9, n
    bsA, bsB, es = list(g["bsA"]), list(g["bsB"]), tuple(g["es"])
    assert relerr(eng.reduce_plk(f, st, bsA, es=es), g["plk_auto"]) < TOL
    last = st
    if "binned_P11l" in g:
        assert np.allclose(eng.keff, g["keff"], rtol=1e-14)
        last = eng.binning(st)
        for n in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc"):
            assert relerr(last[n], g["binned_" + n]) < TOL, n
        ch = eng.chained(last)
        for n in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc"):
            assert relerr(ch[n], g["chained_" + n]) < TOL, n
        assert relerr(eng.reduce_plk(f, last, bsA, es=es), g["plk_binned_auto"]) < TOL
    eng.kmB, eng.krB, eng.ndB = 0.6, 0.3, 2.3e-4
    assert relerr(eng.reduce_plk(f, last, bsA, bsB, es=es), g["plk_cross"]) < TOL


def test_final_nk2048(golden):
    g = golden("caseF")
    eng = oracle_engine(g, "caseF")
    f = float(g["f"])
    st = eng.evaluate(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), pairwise=True)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(st[n], g["ap_" + n]) < 1e-9, n
    assert relerr(eng.reduce_plk(f, st, list(g["bsA"]), es=tuple(g["es"])), g["plk_auto"]) < 1e-9


def test_cfg3_shape_window_chained_cross(golden):
    """Oracle == reference for the cfg-3 shape (Nk=512, window, chained, cross-tracer biases; final stages)."""
    g = golden("caseG")
    eng = oracle_engine(g, "caseG")
    f = float(g["f"])
    st = eng.evaluate(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), pairwise=True)
    assert relerr(eng.Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(st[n], g["window_" + n]) < 1e-9, n
    ch = eng.chained(st)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(ch[n], g["chained_" + n]) < 1e-9, n
    eng.kmB, eng.krB, eng.ndB, eng.No = 0.6, 0.3, 2.3e-4, 2
    assert relerr(eng.reduce_plk(f, ch, list(g["bsA"]), list(g["bsB"]), tuple(g["es"])), g["plk_chained_cross"]) < 1e-9


@pytest.mark.parametrize("tag", ["auto", "cross"])
def test_gaussian_table_and_marginalised_logp(golden, tag):
    """oracle == reference (parambasis.derivative_table, likelihood.flatten, marginal.marginalized_logp)."""
    from oracle import marginal as M

    g = golden("marg")
    st = {n: g["binned_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl")}
    f = float(g["f"])
    kmA, krA, ndA, kmB, krB, ndB = g[tag + "_co"]
    ng = dict(zip(g[tag + "_ng_names"], g[tag + "_ng_values"]))
    names = M.gaussian_names("", ()) if tag == "auto" else M.gaussian_names("X_", ("A_", "B_"))
    assert names == list(g[tag + "_names"])
    if tag == "auto":
        table = M.derivative_table(st, f, ng["b1"], None, kmA, krA, ndA)
    else:
        table = M.derivative_table(st, f, ng["A_b1"], ng["B_b1"], kmA, krA, ndA, kmB, krB, ndB)
    assert relerr(np.stack(table).reshape(len(names), -1), g[tag + "_table"].reshape(len(names), -1)) < 1e-14
    ls = list(g["ls"])
    masks = {l: slice(a, b) for l, (a, b) in zip(ls, g["masks"])}
    PG = np.stack([M.flatten(ls, t, masks) for t in table])
    assert np.array_equal(M.flatten(ls, g[tag + "_PNGl"], masks), g[tag + "_PNG"])
    assert relerr(PG, g[tag + "_PG"]) < 1e-14
    args = (g[tag + "_PG"], g[tag + "_PNG"], g[tag + "_D"], g[tag + "_invcov"], g[tag + "_loc"], g[tag + "_scale"])
    logp, fullchi2, best, F = M.marginalized_logp(*args, return_best=True)
    assert relerr(F["F2"], g[tag + "_F2"]) < 1e-13 and relerr(F["F1"][None], g[tag + "_F1"][None]) < 1e-13
    assert np.isclose(F["F0"], g[tag + "_F0"], rtol=1e-13)
    assert np.isclose(logp, g[tag + "_logp"], rtol=1e-12) and np.isclose(fullchi2, g[tag + "_fullchi2"], rtol=1e-10)
    assert relerr(best[None], g[tag + "_best"][None]) < 1e-10
    assert np.isclose(M.marginalized_logp(*args, jeffreys=True), g[tag + "_logp_jeffreys"], rtol=1e-12)
    if tag == "auto":  # cross + flat prior is singular (cond F2 = 2e18): the sign of det F2 is rounding noise there
        flat = M.marginalized_logp(*args[:4], np.zeros(len(names)), np.full(len(names), np.inf))
        assert np.isclose(flat, g[tag + "_logp_flat"], rtol=1e-12)


def test_eastcoast_basis(golden):
    """oracle == reference EastCoastBasis (reduce_Plk with counterform='eastcoast', Gaussian table)."""
    from oracle import OracleConfig
    from oracle import marginal as M

    g, c = golden("east"), golden("caseC")
    st = {n: c["binned_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc")}
    f = float(g["f"])
    kmA, krA, ndA = g["co"]
    eng = OracleEngine(OracleConfig(Nl=3, kmA=kmA, krA=krA, ndA=ndA))
    full = dict(zip(g["full_names"], g["full_values"]))
    bsA, es = M.eastcoast_bs(f, **full)
    assert relerr(eng.reduce_plk(f, st, bsA, None, es, counterform="eastcoast"), g["plk"]) < 1e-14
    b11, bloop, bct, bst = eng.bias_vectors(f, bsA, None, es, counterform="eastcoast")
    assert relerr(np.einsum("b,lbx->lx", bct, st["Pctl"]), g["Pct"]) < 1e-14
    assert relerr(np.einsum("b,lbx->lx", bst, st["Pstl"]), g["Pst"]) < 1e-14
    assert list(g["names"]) == list(M.EAST_GAUSSIAN)
    table = M.eastcoast_derivative_table(st, f, full["b1"], kmA, ndA)
    assert relerr(np.stack(table).reshape(7, -1), g["table"].reshape(7, -1)) < 1e-14
    bs0, es0 = M.eastcoast_bs(f, full["b1"], full["b2"], full["bG2"])
    assert relerr(eng.reduce_plk(f, st, bs0, None, es0, counterform="eastcoast"), g["PNGl"]) < 1e-14
    logp, fullchi2, best, _ = M.marginalized_logp(g["PG"], g["PNG"], g["D"], g["invcov"], g["loc"], g["scale"], return_best=True)
    assert np.isclose(logp, g["logp"], rtol=1e-12) and np.isclose(fullchi2, g["fullchi2"], rtol=1e-10)
    assert relerr(best[None], g["best"][None]) < 1e-10


def test_fiber_collision(golden):
    """oracle == reference FiberCollision.fibcolWindow on the post-window templates of caseC."""
    from oracle import fiber as F

    g, c = golden("fiber"), golden("caseC")
    st = {n: c["window_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl")}
    for fiberst in (False, True):
        out = F.fibcol_window(st, c["k"], 3, float(g["fs"]), float(g["Dfc"]), float(g["ktrust"]), fiberst=fiberst)
        tag = "st_" if fiberst else ""
        for n in st:
            assert relerr(out[n], g["fiber_" + tag + n]) < 1e-13, (fiberst, n)
    assert np.array_equal(g["fiber_Pstl"], c["window_Pstl"])


def test_nnlo_counterterms(golden):
    """oracle == reference with Common(with_NNLO=True), stage by stage, both counter-term forms."""
    from oracle import fiber as F
    from oracle import marginal as M

    g = golden("nnlo")
    eng = oracle_engine(g, "nnlo")
    f = float(g["f"])
    taps = {}
    st = eng.evaluate(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), taps=taps)
    assert relerr(taps["pscf"]["CctNNLO"], g["pscf_CctNNLO"]) < TOL
    assert relerr(taps["setpscfl"]["PctNNLOl"], g["setpscfl_PctNNLOl"]) < TOL
    assert relerr(taps["resum"]["PctNNLOl"], g["resum_PctNNLOl"]) < TOL
    for n in ("P11l", "Pctl", "Ploopl", "Pstl", "PctNNLOl"):
        assert relerr(taps["ap"][n], g["ap_" + n]) < TOL, n
        assert relerr(st[n], g["window_" + n]) < 1e-9, n
    bsA, es, cn = list(g["bsA"]), tuple(g["es"]), list(g["cnnlo"])
    assert relerr(eng.reduce_plk(f, taps["ap"], bsA, es=es, cnnloA=cn), g["plk_ap_west"]) < TOL
    fib = F.fibcol_window(st, g["k"], 3, float(g["fs"]), float(g["Dfc"]), float(g["ktrust"]))
    fib["PctNNLOl"] = st["PctNNLOl"] + F.dpcorr(g["k"], g["k"], st["PctNNLOl"], 3, ktrust=float(g["ktrust"]), fs=float(g["fs"]), Dfc=float(g["Dfc"]))
    for n in ("P11l", "Pctl", "Ploopl", "Pstl", "PctNNLOl"):
        assert relerr(fib[n], g["fiber_" + n]) < 1e-9, n
    binned = eng.binning(fib)
    ch = eng.chained(binned)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl", "PctNNLOl"):
        assert relerr(binned[n], g["binned_" + n]) < 1e-9, n
        assert relerr(ch[n], g["chained_" + n]) < 1e-9, n
    assert relerr(eng.reduce_plk(f, binned, bsA, es=es, cnnloA=cn), g["plk_binned_west"]) < 1e-9
    full = dict(zip(g["east_names"], g["east_values"]))
    ctilde = full.pop("ctilde")
    bsE, esE = M.eastcoast_bs(f, **full)
    assert relerr(eng.reduce_plk(f, binned, bsE, None, esE, counterform="eastcoast", cnnloA=[ctilde, 0.0]), g["plk_binned_east"]) < 1e-9
    # Gaussian-table entries of the NNLO parameters (parambasis.py:303-307, 429-435)
    Pn = binned["PctNNLOl"]
    b1 = full["b1"]
    assert relerr(-(b1**2) * f**4 * Pn[:, 0] - 2.0 * b1 * f**5 * Pn[:, 1] - f**6 * Pn[:, 2], g["east_table_ctilde"]) < 1e-9
    assert relerr(0.25 * 2.1**2 / eng.krA**4 * Pn[:, 0], g["west_table_cr4"]) < 1e-9
    assert relerr(0.25 * 2.1 / eng.krA**4 * Pn[:, 1], g["west_table_cr6"]) < 1e-9


@pytest.mark.parametrize("mode", ["all", "loop", "resum"])
def test_ircutoff(golden, mode):
    """oracle == reference with Common(IRcutoff=mode, kIR)."""
    from oracle import OracleConfig

    g = golden("ircut")
    eng = OracleEngine(OracleConfig(Nl=3, kmA=0.7, krA=0.25, ndA=4.5e-5, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]),
                                    H_AP=float(g["H_AP"]), IRcutoff=mode, kIR=float(g["kIR"])))
    taps = {}
    st = eng.evaluate(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), taps=taps)
    for n in ("P22", "P13", "C11", "Cct"):
        assert relerr(taps["pscf"][n], g[f"{mode}_pscf_{n}"]) < TOL, n
    assert relerr(taps["pscf"]["C22"][0], g[f"{mode}_pscf_C22_l0"]) < TOL and relerr(taps["pscf"]["C13"][1], g[f"{mode}_pscf_C13_l2"]) < TOL
    assert relerr(taps["resum"]["X"][None], g[f"{mode}_X"][None]) < TOL and relerr(taps["resum"]["Y"][None], g[f"{mode}_Y"][None]) < TOL
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(st[n], g[f"{mode}_ap_{n}"]) < TOL, n
    with pytest.raises(ValueError):
        OracleEngine(OracleConfig(Nl=2, IRcutoff="all")).pscf(g["kin"], g["Pin"])


def test_optiresum(golden):
    """oracle == reference with Common(optiresum=True)."""
    from oracle import OracleConfig

    g = golden("opti")
    eng = OracleEngine(OracleConfig(Nl=3, kmA=0.7, krA=0.25, ndA=4.5e-5, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]),
                                    H_AP=float(g["H_AP"]), optiresum=True))
    assert np.array_equal(eng.s, g["s"]) and np.array_equal(eng.sr, g["sr"])
    taps = {}
    st = eng.evaluate(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), taps=taps)
    assert relerr(taps["pscf"]["C11"], g["pscf_C11"]) < TOL and relerr(taps["pscf"]["Cct"], g["pscf_Cct"]) < TOL
    assert relerr(taps["pscf"]["C22"][1], g["pscf_C22_l2"]) < TOL
    assert relerr(taps["setpscfl"]["Cloopl"].reshape(36, -1), g["setpscfl_Cloopl"].reshape(36, -1)) < TOL
    assert relerr(eng.extract_bao(taps["pscf"]["Cct"]), g["bao_Cct"]) < 1e-9
    assert relerr(taps["resum"]["X"][None], g["X"][None]) < TOL and relerr(taps["resum"]["Y"][None], g["Y"][None]) < TOL
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(taps["resum"][n], g["resum_" + n]) < TOL, n
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(st[n], g["ap_" + n]) < TOL, n


def test_window_matrix(golden):
    """oracle == reference WindowMatrix (to_window_matrix + convolve) on the AP-stage templates of caseC"""
    from oracle import engine as OE

    g, c = golden("wmat"), golden("caseC")
    st = g["stacked"].astype(np.float64)
    m = OE.to_window_matrix(st, (0, 2, 4), 0, 0.4, 400, (0, 1, 2, 3, 4), 0, 0.4, 40, (0, 2, 4), c["k"].max(), tuple(g["ells"]), float(g["kmin"]),
                            float(g["kmax"]))
    assert m.shape == tuple(g["matrix_shape"]) and np.array_equal(m[:, :, ::5, ::37], g["matrix_spot"])
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(OE.window_matrix_convolve(c["k"], m, c["ap_" + n]), g["wm_" + n]) < 1e-12, n
    assert relerr(OE.window_matrix_convolve(c["k"], m, c["ap_Pstl"]), g["st_wm_Pstl"]) < 1e-12
    assert np.array_equal(g["wm_Pstl"], c["ap_Pstl"])


def test_cfg3_production_window_and_joint_likelihood(golden):
    """BASELINE cfg 3 with the parameters the reference ships (yaml: accboost 4, windowk 0.1, per-tracer DR16 windows, ELG chained,
    window_st off as a variant): oracle window / binning / chained on the reference's AP-stage templates, then EFTLike.PNG / PG and the
    Jeffreys, flat-prior marginalised log-posterior on the reference's own data vector and covariance -- full and `_xnost` sets."""
    import os

    import cfg3_util as U
    from oracle import OracleConfig
    from oracle import marginal as M

    g = golden("cfg3")
    p = U.params(g)
    # oracle window at production settings: the cross spectrum's window (one FFTLog(4096) precompute ~ 20 s; the other two tracers go
    # through the host table builder in test_tables.py and through the device precompute in the GPU tests)
    t = "X_NGC"
    eng = OracleEngine(OracleConfig(Nl=3, ndA=4.5e-5, window_file=os.path.join(U.GOLD, "win_NGC_X_sQ024.npy"), window_accboost=int(g["accboost"]),
                                    windowk=float(g["windowk"]), kout=g[t + "_kout"]))
    assert np.array_equal(eng.p, g["window_p"]) and eng.p.size == 1540
    assert relerr(eng.Waldk[:, :, 10, :], g[t + "_Waldk_k10"]) < 1e-12 and relerr(eng.Waldk[:, :, 37, :], g[t + "_Waldk_k37"]) < 1e-12
    assert relerr(eng.Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-12
    st = {n: g[f"{t}_ap_{n}"] for n in U.NAMES}
    st["Picc"] = np.zeros((3, 50))
    w = eng.window(st)
    for n in U.NAMES:
        assert relerr(w[n], g[f"{t}_window_{n}"]) < TOL, n
    b = eng.binning(w)
    assert relerr(eng.keff[None], g[t + "_keff"][None]) < 1e-14
    for n in U.NAMES:
        assert relerr(b[n], g[f"{t}_binned_{n}"]) < TOL, n
    # chained ELG from its binned templates
    ch = eng.chained({n: g["ELG_NGC_binned_" + n] for n in U.NAMES} | {"Picc": np.zeros((3, 17))})
    for n in U.NAMES:
        assert relerr(ch[n], g["ELG_NGC_chained_" + n]) < 1e-13, n
    # joint data vector: P_NG and P_G of every tracer in EFTLike's order
    PNG, PGrows = [], {}
    for t, sc in zip(U.TRACERS, U.scales(g)):
        st = U.final_templates(g, t)
        No = st["P11l"].shape[0]
        cross = t in U.CROSS
        A, B = (U.CROSS[t] if cross else (t, t))
        ngA = [p[A + "_b1"], p[A + "_b2"], 0.0, p[A + "_b4"], 0.0, 0.0, 0.0]
        ngB = [p[B + "_b1"], p[B + "_b2"], 0.0, p[B + "_b4"], 0.0, 0.0, 0.0] if cross else None
        oe = OracleEngine(OracleConfig(Nl=3, No=No, **sc))
        plk = oe.reduce_plk(float(g[t + "_f"]), dict(st, Picc=np.zeros((No, st["P11l"].shape[-1]))), ngA, ngB, (0.0, 0.0, 0.0))
        assert relerr(plk, g[t + "_plk"][:No]) < 1e-13, t
        mk = U.masks(g, t)
        ls = [int(l) for l in g[t + "_ls"]]
        PNG.append(M.flatten(ls, plk, mk))
        tab = M.derivative_table(st, float(g[t + "_f"]), p[A + "_b1"], p[B + "_b1"] if cross else None, **sc)
        for name, arr in zip(M.gaussian_names(t + "_", tuple(x + "_" for x in U.CROSS[t]) if cross else ()), tab):
            PGrows.setdefault(name, []).append((t, M.flatten(ls, arr, mk)))
    PNG = np.hstack(PNG)
    assert relerr(PNG[None], g["PNG"][None]) < 1e-13
    sizes = {t: int(sum(b - a for a, b in g[t + "_mask"])) for t in U.TRACERS}
    starts = dict(zip(U.TRACERS, np.concatenate([[0], np.cumsum([sizes[t] for t in U.TRACERS])])))
    for tag in ("full", "xnost"):
        names = [str(n) for n in g[tag + "_names"]]
        PG = np.zeros((len(names), PNG.size))
        for i, n in enumerate(names):
            for t, v in PGrows[n]:
                PG[i, starts[t] : starts[t] + sizes[t]] = v
        assert relerr(PG, g[tag + "_PG"]) < 1e-13, tag
        flat = (np.zeros(len(names)), np.full(len(names), np.inf))
        logp, fullchi2, best, _ = M.marginalized_logp(PG, PNG, g["data_vector"], g["invcov"], *flat, jeffreys=True, return_best=True)
        assert np.isclose(logp, g[tag + "_logp"], rtol=1e-11) and np.isclose(fullchi2, g[tag + "_fullchi2"], rtol=1e-9), tag
        assert relerr(best[None], g[tag + "_best"][None]) < 1e-8, tag
        assert np.isclose(M.marginalized_logp(PG, PNG, g["data_vector"], g["invcov"], *flat), g[tag + "_logp_nojeffreys"], rtol=1e-11)


def test_resum_nondefault_options(golden):
    """Resum(LambdaIR=0.25, NFFT=128) and Resum.Ps(window=0.3) (reference pybird.py:1230-1300, 1409-1464): oracle == reference."""
    from oracle import OracleConfig

    g = golden("resumopt")
    eng = OracleEngine(OracleConfig(Nl=3, kmA=0.7, krA=0.25, ndA=4.5e-5, with_resum=True, LambdaIR=float(g["LambdaIR"]), NFFT_resum=int(g["NFFT"]),
                                    resum_window=float(g["window"])))
    taps = {}
    st = eng.evaluate(g["kin"], g["Pin"], float(g["f"]), taps=taps)
    assert relerr(st["X"][None], g["X"][None]) < TOL and relerr(st["Y"][None], g["Y"][None]) < TOL
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(st[n], g["resum_" + n]) < TOL, n
    assert relerr(taps["setpscfl"]["Ploopl"], g["setpscfl_Ploopl"]) < TOL and relerr(taps["setpscfl"]["Cloopl"], g["setpscfl_Cloopl"]) < TOL


def test_cfg3_nk512_oracle_ap_stage_other_redshift(golden):
    """cfg 3 on the BASELINE grid (tests/golden/cfg3_nk512.npz): the oracle through resummation + AP (APst) for the ELG kernel (z = 0.849, its
    own P_lin and AP fiducial) at Nk = 512 against the reference -- the Nk = 512 pin of caseD at a second redshift; the production window
    at this grid is pinned through the host table builder (test_host_logic.py) and the device precompute (test_gpu_cfg3.py), the
    oracle's own 4096-point precompute at Np = 1540 x Nk = 512 takes minutes."""
    import cfg3_util as U
    from oracle import OracleConfig

    g = golden("cfg3_nk512")
    t = "ELG_NGC"
    sc = U.scales(g)[U.TRACERS.index(t)]
    eng = OracleEngine(OracleConfig(Nl=3, k=g["k"], with_resum=True, with_ap=True, APst=True, DA_AP=float(g[t + "_DA_AP"]), H_AP=float(g[t + "_H_AP"]), **sc))
    st = eng.evaluate(g["kin"], g[t + "_Pin"], float(g[t + "_f"]), float(g[t + "_DA"]), float(g[t + "_H"]), pairwise=True)
    for n in U.NAMES:
        assert relerr(st[n], g[f"{t}_ap_{n}"]) < TOL, n
