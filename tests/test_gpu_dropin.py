"""GPU: the PyBird-compatible classes driven exactly as reference eftpipe/theory.py:557-585 drives
`eftpipe.pybird.pybird` (mutate-in-place semantics), checked against the reference's own outputs."""
import numpy as np
import pytest

from conftest import relerr
from eftpipe_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.mark.parametrize("name", ["caseC", "caseD", "caseE"])
def test_theory_sequence(golden, name):
    from eftpipe_amd import pybird
    from eftpipe_amd.parambasis import reduce_Plk

    g = golden(name)
    Nl, z = int(g["Nl"]), float(g["z"])
    co = pybird.Common(Nl=Nl, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    if g["k"].size != 50:  # non-native grid: overwrite before building the plugins
        co.k, co.Nk = g["k"], g["k"].size
        co.kr = co.k[0.02 <= co.k]
        co.Nkr = co.kr.size
        co.Nklow = co.Nk - co.Nkr
    nonlinear = pybird.NonLinear(load=False, save=False, co=co)
    resum = pybird.Resum(co=co)
    ap = pybird.APeffect(Om_AP=synth.OM_AP, z_AP=z, co=co, APst=(name == "caseC"))
    assert np.isclose(ap.DA, g["DA_AP"], rtol=1e-13) and np.isclose(ap.H, g["H_AP"], rtol=1e-15)

    bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), z, co=co)
    nonlinear.PsCf(bird)
    for n in ("P11", "P22", "P13", "C11", "Cct", "C22", "C13"):
        a = getattr(bird, n)
        assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
        assert relerr(a, g["pscf_" + n]) < TOL, n
    bird.setPsCfl()
    for n in ("P11l", "Pctl", "Ploopl", "Cloopl", "Pstl"):
        assert getattr(bird, n).shape == g["setpscfl_" + n].shape
        assert relerr(getattr(bird, n), g["setpscfl_" + n]) < TOL, n
    resum.Ps(bird)
    assert relerr(resum.Q.reshape(-1, resum.Nn), g["resum_Q"].reshape(-1, resum.Nn)) < 1e-12
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(getattr(bird, n), g["resum_" + n]) < TOL, n
    ap.AP(bird)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(getattr(bird, n), g["ap_" + n]) < TOL, n
    qperp, qpar = ap.get_AP_param(bird)
    assert ap.get_alperp_alpara(bird) == (qperp, qpar)
    if name != "caseC":
        plk = reduce_Plk(bird, list(g["bsA"]), es=tuple(g["es"])).sum()
        assert relerr(plk, g["plk_auto"]) < TOL
        nz = np.abs(g["plk_auto"]) > 1e-3 * np.max(np.abs(g["plk_auto"]), axis=-1, keepdims=True)
        assert np.max(np.abs(plk / g["plk_auto"] - 1.0)[nz]) < 1e-6  # the north-star bar, pointwise


def test_q_override_and_snapshots(golden):
    from eftpipe_amd import pybird

    g = golden("caseE")
    co = pybird.Common(Nl=2, kmA=0.7, krA=0.25, ndA=4.5e-5)
    nl = pybird.NonLinear(co=co)
    ap = pybird.APeffect(Om_AP=synth.OM_AP, z_AP=0.7, co=co, snapshot=True)
    bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    nl.PsCf(bird)
    bird.setPsCfl()
    before = bird.Ploopl.copy()
    ap.AP(bird, q=(1.0, 1.0))  # identity distortion: the mu quadrature of the spline reproduces the input
    # trapezoid rule on 200 mu nodes: identity up to the quadrature error of the Legendre products
    assert np.max(np.abs(bird.P11l - np.einsum("x,ln->lnx", bird.P11, co.l11))) < 5e-4 * np.max(bird.P11)
    assert "APeffect" in bird.snapshots and np.array_equal(bird.snapshots["APeffect"].Ploopl, bird.Ploopl)
    assert np.max(np.abs(bird.Ploopl - before)) < 5e-3 * np.max(np.abs(before))


def test_nonlinear_pyegg_cache_round_trip(golden, tmp_path):
    """NonLinear(load, save, path) as in the reference (pybird.py:917-981): the first construction writes
    pyegg256_Nl2.npz in the reference's layout, the second one loads it and gives bit-identical loop pieces."""
    from eftpipe_amd import pybird
    from eftpipe_amd.tables import PYEGG_KEYS

    g = golden("caseA")
    path = str(tmp_path)

    def run(load, save):
        co = pybird.Common(Nl=2, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
        nl = pybird.NonLinear(load=load, save=save, path=path, co=co)
        bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), co=co)
        nl.PsCf(bird)
        return nl, bird

    nl1, b1 = run(load=True, save=True)  # nothing to load yet: computed and saved
    assert not nl1.loaded
    egg = tmp_path / "pyegg256_Nl2.npz"
    assert egg.exists()
    with np.load(egg) as z:
        assert list(z.files) == list(PYEGG_KEYS) and z["Mcf22"].shape == (28, 2, 257, 257)
    nl2, b2 = run(load=True, save=True)
    assert nl2.loaded
    for n in ("P22", "P13", "C11", "Cct", "C22", "C13"):
        assert np.array_equal(getattr(b1, n), getattr(b2, n)), n
        assert relerr(getattr(b2, n), g["pscf_" + n]) < TOL, n


def test_with_nnlo_sequence_and_batched(golden):
    """Common(with_NNLO=True).  The k^4 counter-terms PctNNLOl go through Resum / AP / window / fibre / binning /
    chained as a second template block; drop-in call sequence (reference theory.py:557-604) and the batched pipeline with the device
    reduce, against the reference-generated fixture, in both counter-term forms."""
    import os

    from eftpipe_amd import pybird
    from eftpipe_amd import tables as TB
    from eftpipe_amd.binning import Binning
    from eftpipe_amd.chained import Chained
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import EastCoastBasis, WestCoastBasis, bias_row, nnlo_vector, reduce_Plk
    from eftpipe_amd.tables import EngineConfig
    from eftpipe_amd.window import Window

    WIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "win_NGC_LRG_sQ024.npy")
    g = golden("nnlo")
    names = ("P11l", "Pctl", "Ploopl", "Pstl", "PctNNLOl")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5, with_NNLO=True)
    f = float(g["f"])
    bird = pybird.Bird(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), 0.7, co=co)
    pybird.NonLinear(load=False, save=False, co=co).PsCf(bird)
    assert relerr(bird.CctNNLO, g["pscf_CctNNLO"]) < TOL
    bird.setPsCfl()
    assert relerr(bird.PctNNLOl, g["setpscfl_PctNNLOl"]) < TOL
    pybird.Resum(co=co).Ps(bird)
    assert relerr(bird.PctNNLOl, g["resum_PctNNLOl"]) < TOL
    pybird.APeffect(DA=float(g["DA_AP"]), H=float(g["H_AP"]), co=co, APst=True).AP(bird)
    for n in names:
        assert relerr(getattr(bird, n), g["ap_" + n]) < TOL, n
    bsA, es, cn = list(g["bsA"]), tuple(g["es"]), list(g["cnnlo"])
    assert relerr(reduce_Plk(bird, bsA, es=es, cnnloA=cn).sum(), g["plk_ap_west"]) < TOL
    Window(window_configspace_file=WIN, co=co, load=False, save=False).Window(bird)
    for n in names:
        assert relerr(getattr(bird, n), g["window_" + n]) < TOL, n
    fs, Dfc, kt = float(g["fs"]), float(g["Dfc"]), float(g["ktrust"])
    pybird.FiberCollision(fs=fs, Dfc=Dfc, ktrust=kt, co=co).fibcolWindow(bird)
    for n in names:
        assert relerr(getattr(bird, n), g["fiber_" + n]) < TOL, n
    binned = Binning(kout=g["kout"], co=co).transform(bird)
    ch = Chained().transform(binned)
    for n in names:
        assert relerr(getattr(binned, n), g["binned_" + n]) < TOL, n
        assert relerr(getattr(ch, n), g["chained_" + n]) < TOL, n
    assert relerr(reduce_Plk(binned, bsA, es=es, cnnloA=cn).sum(), g["plk_binned_west"]) < TOL
    full = dict(zip(g["east_names"], g["east_values"]))
    coe = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5, with_NNLO=True, counterform="eastcoast")
    binned.co = coe
    east = EastCoastBasis(prefix="")
    assert relerr(east.reduce_Plk(binned, full).sum(), g["plk_binned_east"]) < TOL
    tab = east.reduce_Plk_gaussian_table(binned, {p: full[p] for p in ("b1", "b2", "bG2")})
    assert list(tab) == ["bGamma3", "c0", "c2", "c4", "ctilde", "Pshot", "a0", "a2"] and relerr(tab["ctilde"], g["east_table_ctilde"]) < TOL
    binned.co = co
    tabw = WestCoastBasis(prefix="").reduce_Plk_gaussian_table(binned, {"b1": 2.1, "b2": 0.5, "b4": 0.1})
    assert list(tabw) == ["b3", "cct", "cr1", "cr2", "cr4", "cr6", "ce0", "cemono", "cequad"]
    assert relerr(tabw["cr4"], g["west_table_cr4"]) < TOL and relerr(tabw["cr6"], g["west_table_cr6"]) < TOL

    # batched: one folded operator (window -> fibre -> binning; Pstl without the fibre matrix) and the device reduce with BIASN
    k = g["k"]
    wt = np.load(WIN)
    Wal, p = TB.window_matrix(k, wt[:, 0], wt[:, 1:].T, 3, 3)
    Wfold, _ = TB.window_fold(k, Wal, p)
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    Fm = TB.fiber_operator(k, 3, fs, Dfc, kt)
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, with_NNLO=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=3)
    eng.set_pipeline_operator(eng.add_operator(TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, fiber=Fm),
                                               stochastic=TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm)))
    Pin = np.stack([g["Pin"], 1.07 * g["Pin"], g["Pin"]])
    scales = dict(kmA=0.7, krA=0.25, ndA=4.5e-5)
    bias = np.stack([bias_row(f, bsA, None, es, **scales), bias_row(f, bsA, None, es, **scales), east.bias_row(f, full, **scales)])
    bn = np.stack([nnlo_vector(f, bsA[0], cn, 0.25), nnlo_vector(f, bsA[0], cn, 0.25), east.nnlo_row(f, full)])
    templ, plk = eng.eval_batch(Pin, f, float(g["DA"]), float(g["H"]), bias=bias, bias_nnlo=bn)
    nb = len(g["kout"])
    tn = eng.get("TEMPLN", (3, 3, 24, nb))
    for i in (0, 2):
        assert relerr(tn[i][:, 3:6], g["binned_PctNNLOl"]) < TOL
        assert relerr(templ[i][:, 3:9], g["binned_Pctl"]) < TOL and relerr(templ[i][:, 9:21], g["binned_Ploopl"]) < TOL
        assert relerr(templ[i][:, 21:24], g["binned_Pstl"]) < TOL
        assert np.all(tn[i][:, :3] == 0.0) and np.all(tn[i][:, 6:21] == 0.0)
    assert relerr(plk[0], g["plk_binned_west"]) < TOL and relerr(plk[2], g["plk_binned_east"]) < TOL
    assert np.max(np.abs(plk[1] - plk[0])) > 0
    # same engine, stage by stage without the tail, then REDUCE alone
    eng.run(eng.full_mask(reduce=False), 3)
    from eftpipe_amd import _lib as L

    eng.run(L.S_REDUCE, 3)
    assert np.array_equal(eng.get("PLK", (3, 3, nb)), plk)
    # LOGP with the NNLO parameters cr4, cr6 marginalised as well (reference parambasis.py:303-307): rows over both blocks
    from eftpipe_amd.marginal import MarginalLikelihood, data_index
    from eftpipe_amd.parambasis import gaussian_rows
    from oracle import marginal as M

    b1, b2, b4 = bsA[0], bsA[1], bsA[3]
    rows = np.zeros((3, 10, 24))
    rows[:, :8] = gaussian_rows(f, (b1, b2, b4), None, 0.7, 0.25, 4.5e-5)
    rn = np.zeros((3, 10, 3))
    rn[:, 8, 0], rn[:, 9, 1] = 0.25 * b1**2 / 0.25**4, 0.25 * b1 / 0.25**4
    index = data_index([0, 2, 4], {0: slice(0, nb), 2: slice(1, nb - 2), 4: slice(2, 10)}, nb)
    rng = np.random.default_rng(11)
    V = [(np.einsum("gr,lrx->glx", rows[i], templ[i]) + np.einsum("gj,ljx->glx", rn[i], tn[i][:, 3:6])).reshape(10, -1)[:, index] for i in range(3)]
    sig = 0.05 * np.abs(V[0][0]) + 15.0
    D = V[0][0] + sig * rng.normal(size=index.size)
    C = np.diag(1.0 / sig**2)
    loc, scale = np.zeros(9), np.array([2.0, 2.0, 4.0, 4.0, 2.0, 2.0, 2.0, 3.0, 3.0])
    like = MarginalLikelihood(eng, index, D, C, loc, scale)
    lp, full, best = like.logp(rows, return_best=True, rows_nnlo=rn)
    for i in range(3):
        w = M.marginalized_logp(V[i][1:], V[i][0], D, C, loc, scale, return_best=True)
        assert np.isclose(lp[i], w[0], rtol=1e-10) and np.isclose(full[i], w[1], rtol=1e-9) and relerr(best[i][None], w[2][None]) < 1e-8, i
    assert abs(lp[0] - like.logp(rows)[0]) > 1e-3  # the NNLO rows matter
    eng.close()



@pytest.mark.parametrize("mode", ["all", "loop", "resum"])
def test_ircutoff_modes(golden, mode):
    """Common(IRcutoff=mode, kIR) (reference pybird.py:1127-1160, 1316-1335): the cut FFTLog operator is a table;
    "loop" / "resum" use two coefficient sets on the device.  Drop-in sequence and batched engine against reference outputs."""
    from eftpipe_amd import pybird
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    g = golden("ircut")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5, IRcutoff=mode, kIR=float(g["kIR"]))
    f = float(g["f"])
    bird = pybird.Bird(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), 0.7, co=co)
    pybird.NonLinear(load=False, save=False, co=co).PsCf(bird)
    for n in ("P22", "P13", "C11", "Cct"):
        assert relerr(getattr(bird, n), g[f"{mode}_pscf_{n}"]) < TOL, n
    assert relerr(bird.C22[0], g[f"{mode}_pscf_C22_l0"]) < TOL and relerr(bird.C13[1], g[f"{mode}_pscf_C13_l2"]) < TOL
    bird.setPsCfl()
    rs = pybird.Resum(co=co)
    X, Y = rs.IRFilters(bird)
    assert relerr(X[None], g[f"{mode}_X"][None]) < TOL and relerr(Y[None], g[f"{mode}_Y"][None]) < TOL
    rs.Ps(bird)
    pybird.APeffect(DA=float(g["DA_AP"]), H=float(g["H_AP"]), co=co).AP(bird)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(getattr(bird, n), g[f"{mode}_ap_{n}"]) < TOL, n
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]), IRcutoff=mode, kIR=float(g["kIR"])),
                 max_batch=2)
    templ = eng.eval_batch(np.stack([g["Pin"], 1.02 * g["Pin"]]), f, float(g["DA"]), float(g["H"]))
    assert relerr(templ[0][:, 9:21], g[f"{mode}_ap_Ploopl"]) < TOL and relerr(templ[0][:, 0:3], g[f"{mode}_ap_P11l"]) < TOL
    assert relerr(templ[1][:, 9:21] / 1.02**2, g[f"{mode}_ap_Ploopl"]) > 1e-6  # (X, Y enter the resummation non-linearly)
    eng.close()
    with pytest.raises(ValueError):
        pybird.Common(Nl=3, IRcutoff="all")
    with pytest.raises(ValueError):
        pybird.Common(Nl=3, IRcutoff="some", kIR=0.01)


def test_optiresum(golden):
    """Common(optiresum=True) (reference pybird.py:553-556, 1235-1244, 1382-1400): xi pieces on the 52-point s grid,
    BAO-peak extraction on the device, resummation over the 48 BAO points -- drop-in sequence and batched engine vs reference outputs."""
    from eftpipe_amd import pybird
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    g = golden("opti")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5, optiresum=True)
    assert np.array_equal(co.s, g["s"]) and co.Ns == 52
    f = float(g["f"])
    bird = pybird.Bird(g["kin"], g["Pin"], f, float(g["DA"]), float(g["H"]), 0.7, co=co)
    pybird.NonLinear(load=False, save=False, co=co).PsCf(bird)
    assert bird.C11.shape == (3, 52) and bird.C22.shape == (3, 28, 52)
    assert relerr(bird.C11, g["pscf_C11"]) < TOL and relerr(bird.Cct, g["pscf_Cct"]) < TOL
    assert relerr(bird.C22[1], g["pscf_C22_l2"]) < TOL
    bird.setPsCfl()
    assert relerr(bird.Cloopl.reshape(36, -1), g["setpscfl_Cloopl"].reshape(36, -1)) < TOL
    rs = pybird.Resum(co=co)
    assert np.array_equal(rs.sr, g["sr"])
    X, Y = rs.IRFilters(bird)
    assert relerr(X[None], g["X"][None]) < TOL and relerr(Y[None], g["Y"][None]) < TOL
    rs.Ps(bird)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(getattr(bird, n), g["resum_" + n]) < TOL, n
    pybird.APeffect(DA=float(g["DA_AP"]), H=float(g["H_AP"]), co=co).AP(bird)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        assert relerr(getattr(bird, n), g["ap_" + n]) < TOL, n
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]), optiresum=True), max_batch=2)
    templ = eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), f, float(g["DA"]), float(g["H"]))
    for i in range(2):
        assert relerr(templ[i][:, 9:21], g["ap_Ploopl"]) < TOL and relerr(templ[i][:, 3:9], g["ap_Pctl"]) < TOL
    eng.close()


def test_resum_nondefault_options_and_helper_methods(golden):
    """Surface the reference offers beyond what theory.py calls: Resum(LambdaIR, NFFT), Resum.Ps(window=...), Resum.IRFilters / setXpYp /
    makeQ, NonLinear.Coef, Bird.reducePsCfl / setPstl / subtractShotNoise and the mu-weighted by-products of setPsCfl -- against the
    reference (tests/golden/resumopt.npz)."""
    from eftpipe_amd import pybird

    g = golden("resumopt")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    nl = pybird.NonLinear(load=False, save=False, co=co)
    bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    c = nl.Coef(bird, window=None)
    assert np.max(np.abs(c - g["coef_window_none"])) < 1e-12 * np.max(np.abs(g["coef_window_none"]))
    nl.PsCf(bird)
    bird.setPsCfl()
    rs = pybird.Resum(LambdaIR=float(g["LambdaIR"]), NFFT=int(g["NFFT"]), co=co)
    X, Y = rs.IRFilters(bird)
    assert relerr(X[None], g["X"][None]) < 1e-10 and relerr(Y[None], g["Y"][None]) < 1e-10
    assert relerr(rs.setXpYp(bird), g["XpYp"]) < 1e-10
    rs.Ps(bird, window=float(g["window"]))
    assert relerr(rs.Q.reshape(-1, rs.Nn), g["Q"].reshape(-1, rs.Nn)) < 1e-12
    rs.makeQ(0.5)
    assert rs.Q.shape == (2, 3, 3, 96)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(getattr(bird, n), g["resum_" + n]) < TOL, n
    Xs, Ys = rs.IRFilters(bird, soffset=2.0)   # other arguments select other tables (test_surface_arguments_off_their_defaults pins them) ...
    assert Xs.shape == X.shape and not np.array_equal(Xs, X) and np.array_equal(Ys, Y)   # soffset moves X's offset term only
    X2, Y2 = rs.IRFilters(bird)                 # ... and the constructor's tables are back afterwards
    assert np.array_equal(X2, X) and np.array_equal(Y2, Y)
    # a second bird on the default-option engine of another Common is not disturbed
    b2 = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), co=co)
    nl.PsCf(b2)
    b2.setPsCfl()
    assert relerr(b2.P22l.reshape(-1, 50), g["P22l"].reshape(-1, 50)) < TOL and relerr(b2.P13l.reshape(-1, 50), g["P13l"].reshape(-1, 50)) < TOL
    assert relerr(b2.C22.reshape(-1, 80), g["C22w"].reshape(-1, 80)) < TOL and relerr(b2.C13.reshape(-1, 80), g["C13w"].reshape(-1, 80)) < TOL
    assert relerr(b2.Cloopl, g["setpscfl_Cloopl"]) < TOL
    # host helper forms
    b2.reducePsCfl()
    assert relerr(b2.Ploopl, g["setpscfl_Ploopl"]) < TOL and relerr(b2.Cloopl, g["setpscfl_Cloopl"]) < TOL
    b2.setPstl()
    assert np.array_equal(b2.Pstl, g["setpscfl_Pstl"])


def test_lazy_bird_attributes_follow_host_mutation(golden):
    """The stage outputs stay on the device until read; a host read, an in-place change of the array that was read, or a re-binding must all
    be honoured by the next stage (the reference's plugins mutate the bird in place, theory.py:568-604)."""
    from eftpipe_amd import pybird

    g = golden("caseC")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    nl, rs = pybird.NonLinear(load=False, save=False, co=co), pybird.Resum(co=co)
    ap = pybird.APeffect(Om_AP=synth.OM_AP, z_AP=0.7, co=co, APst=True)

    def run(mutate):
        bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
        nl.PsCf(bird)
        bird.setPsCfl()
        rs.Ps(bird)
        mutate(bird)
        ap.AP(bird)
        return {n: getattr(bird, n).copy() for n in ("P11l", "Pctl", "Ploopl", "Pstl")}

    base = run(lambda b: None)
    for n in base:
        assert relerr(base[n], g["ap_" + n]) < TOL, n
    read_only = run(lambda b: b.P11l)                              # a read alone changes nothing
    assert all(np.array_equal(read_only[n], base[n]) for n in base)

    def in_place(b):
        b.Pctl[...] *= 2.0                                         # array handed out by the read, modified in place

    doubled = run(in_place)
    assert relerr(doubled["Pctl"], 2.0 * base["Pctl"]) < 1e-13 and np.array_equal(doubled["P11l"], base["P11l"])

    def rebind(b):
        b.Ploopl = 3.0 * b.Ploopl

    tripled = run(rebind)
    assert relerr(tripled["Ploopl"], 3.0 * base["Ploopl"]) < 1e-13 and np.array_equal(tripled["Pstl"], base["Pstl"])
    # two birds in flight on one engine: the first one's pending results are fetched before the second one runs
    b1 = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    nl.PsCf(b1)
    b1.setPsCfl()
    b2 = pybird.Bird(g["kin"], 1.1 * g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    nl.PsCf(b2)
    b2.setPsCfl()
    assert relerr(b1.Ploopl, g["setpscfl_Ploopl"]) < TOL and relerr(b2.P11l, 1.1 * g["setpscfl_P11l"]) < TOL
    rs.Ps(b1)
    assert relerr(b1.Ploopl, g["resum_Ploopl"]) < TOL


def test_surface_arguments_off_their_defaults(golden):
    """NonLinear.PsCf(bird, window=0.3), a Bird on another input grid (240 samples up to k = 10^0.2) through PsCf /
    setPsCfl / Resum.Ps, and Resum.IRFilters(bird, soffset, LambdaIR, RescaleIR, window) -- each selects other operator tables of the same
    device kernels (the engine rebuilds them, as Resum(LambdaIR, NFFT) always did) -- against the REAL reference (tests/golden/surface.npz;
    reference pybird.py:682-695, 1143-1171, 1316-1353); and back to the defaults afterwards."""
    from eftpipe_amd import pybird

    g, c = golden("surface"), golden("caseC")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    f, DA, H, z = float(g["f"]), float(g["DA"]), float(g["H"]), float(g["z"])
    nl, rs = pybird.NonLinear(load=False, save=False, co=co), pybird.Resum(co=co)
    # (1) another coefficient window
    bird = pybird.Bird(g["kin"], g["Pin"], f, DA, H, z, co=co)
    nl.PsCf(bird, window=0.3)
    for n in ("P11", "P22", "P13", "C11", "Cct", "C22", "C13"):
        assert relerr(getattr(bird, n), g["w03_" + n]) < TOL, n
    # (2) IR filters off their defaults, then with them (the tables are put back)
    b2 = pybird.Bird(g["kin"], g["Pin"], f, DA, H, z, co=co)
    so, lam, resc, win = (float(x) for x in g["irf"])
    X, Y = rs.IRFilters(b2, soffset=so, LambdaIR=lam, RescaleIR=resc, window=win)
    assert relerr(X[None], g["irf_X"][None]) < 1e-10 and relerr(Y[None], g["irf_Y"][None]) < 1e-10
    X, Y = rs.IRFilters(b2)
    assert relerr(X[None], g["irf_X_default"][None]) < 1e-10 and relerr(Y[None], g["irf_Y_default"][None]) < 1e-10
    # (3) another input grid
    b3 = pybird.Bird(g["kin2"], g["Pin2"], f, DA, H, z, co=co)
    nl.PsCf(b3)
    for n in ("P11", "P22", "P13", "C11"):
        assert relerr(getattr(b3, n), g["kin2_" + n]) < TOL, n
    b3.setPsCfl()
    rs.Ps(b3)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(getattr(b3, n), g["kin2_resum_" + n]) < TOL, n
    # (4) and the shipped call sequence still gives the shipped answer on the same Common
    b4 = pybird.Bird(c["kin"], c["Pin"], float(c["f"]), float(c["DA"]), float(c["H"]), 0.7, co=co)
    nl.PsCf(b4)
    b4.setPsCfl()
    rs.Ps(b4)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(getattr(b4, n), c["resum_" + n]) < TOL, n
    # NonLinear(NFFT != 256) stays a loud refusal: the anti-diagonal tables and the synthesis kernels are sized for 257 powers
    with pytest.raises(NotImplementedError):
        pybird.NonLinear(load=False, save=False, NFFT=128, co=co)
