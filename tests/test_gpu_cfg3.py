"""GPU: BASELINE cfg 3 as the reference ships it -- LRG z=0.696, ELG z=0.849 (chained), cross spectrum z=0.763, DR16 windows at
accboost=4 / windowk=0.1, the reference's own data vector, covariance and marginalised-parameter sets (full and `_xnost`).
Everything is compared with outputs of the REAL reference (tests/golden/cfg3.npz, tools/make_fixtures.py cfg3), through the C ABI."""
import numpy as np
import pytest

import cfg3_util as U
from conftest import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def windows(golden):
    """Window plugin objects of the three tracers at production settings (device precompute, eftb_window_precompute)"""
    from eftpipe_amd import pybird
    from eftpipe_amd.window import Window

    g = golden("cfg3")
    out = {}
    for t, sc in zip(U.TRACERS, U.scales(g)):
        co = pybird.Common(Nl=3, No=3, kmax=0.3, **sc)
        import os

        out[t] = (co, Window(window_configspace_file=os.path.join(U.GOLD, "win_NGC_%s_sQ024.npy" % t.split("_")[0]), co=co, load=False, save=False,
                             accboost=int(g["accboost"]), windowk=float(g["windowk"])))
    return out


@pytest.mark.parametrize("t", U.TRACERS)
def test_window_plugin_at_production_settings(golden, windows, t):
    """eftpipe_amd.window.Window(accboost=4, windowk=0.1): the p grid (1 540 points), the masked dp-weighted matrix against the
    reference's Waldk rows, and Window.Window(bird) against the reference's convolved templates (window.py:27-33, 262-415)."""
    from types import SimpleNamespace

    g = golden("cfg3")
    co, win = windows[t]
    assert np.array_equal(win.p, g["window_p"]) and win.p.size == 1540
    assert relerr(win.Waldk[:, :, 10, :], g[t + "_Waldk_k10"]) < 1e-9 and relerr(win.Waldk[:, :, 37, :], g[t + "_Waldk_k37"]) < 1e-9
    assert relerr(win.Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
    bird = SimpleNamespace(co=co, f=float(g[t + "_f"]), Picc=np.zeros((3, 50)), PctNNLOl=None, **{n: g[f"{t}_ap_{n}"].copy() for n in U.NAMES})
    win.Window(bird)
    for n in U.NAMES:
        assert relerr(getattr(bird, n), g[f"{t}_window_{n}"]) < TOL, n


def test_window_st_off_keeps_the_stochastic_rows(golden, windows):
    """Window(window_st=False) (reference window.py:412-415) at production settings"""
    import os
    from types import SimpleNamespace

    from eftpipe_amd.window import Window

    g = golden("cfg3")
    t = "LRG_NGC"
    co, _ = windows[t]
    win = Window(window_configspace_file=os.path.join(U.GOLD, "win_NGC_LRG_sQ024.npy"), co=co, load=False, save=False, accboost=int(g["accboost"]),
                 windowk=float(g["windowk"]), window_st=False)
    bird = SimpleNamespace(co=co, f=float(g[t + "_f"]), Picc=np.zeros((3, 50)), PctNNLOl=None, **{n: g[f"{t}_ap_{n}"].copy() for n in U.NAMES})
    win.Window(bird)
    for n in U.NAMES:
        assert relerr(getattr(bird, n), g[f"{t}_windownost_{n}"]) < TOL, n
    assert np.array_equal(bird.Pstl, g[t + "_ap_Pstl"])


def test_dropin_sequence_per_tracer(golden, windows):
    """The three kernels of a likelihood point through the PyBird-compatible classes, exactly as reference theory.py:557-609 drives them:
    Bird -> PsCf -> setPsCfl -> Resum.Ps -> APeffect.AP(APst) -> Window -> Binning(kout of the data) [-> Chained]."""
    from eftpipe_amd import pybird
    from eftpipe_amd.binning import Binning
    from eftpipe_amd.chained import Chained
    from eftpipe_amd.parambasis import reduce_Plk

    g = golden("cfg3")
    p = U.params(g)
    for t in U.TRACERS:
        co, win = windows[t]
        z = float(g[t + "_z"])
        nl = pybird.NonLinear(load=False, save=False, co=co)
        rs = pybird.Resum(co=co)
        ap = pybird.APeffect(Om_AP=float(g["Om_AP"]), z_AP=z, rdrag_AP=147.66, h_AP=0.6777, APst=True, co=co)
        assert np.isclose(ap.DA, g[t + "_DA_AP"], rtol=1e-12) and np.isclose(ap.H, g[t + "_H_AP"], rtol=1e-14)
        bird = pybird.Bird(g["kin"], g[t + "_Pin"], float(g[t + "_f"]), float(g[t + "_DA"]), float(g[t + "_H"]), z, co=co)
        nl.PsCf(bird)
        bird.setPsCfl()
        rs.Ps(bird)
        ap.AP(bird)
        for n in U.NAMES:
            assert relerr(getattr(bird, n), g[f"{t}_ap_{n}"]) < TOL, (t, n)
        win.Window(bird)
        for n in U.NAMES:
            assert relerr(getattr(bird, n), g[f"{t}_window_{n}"]) < TOL, (t, n)
        bn = Binning(kout=g[t + "_kout"], co=co)
        assert relerr(bn.keff[None], g[t + "_keff"][None]) < 1e-13
        like = bn.transform(bird)
        for n in U.NAMES:
            assert relerr(getattr(like, n), g[f"{t}_binned_{n}"]) < TOL, (t, n)
        if U.CHAINED[t]:
            like = Chained().transform(like)
            for n in U.NAMES:
                assert relerr(getattr(like, n), g[f"{t}_chained_{n}"]) < TOL, (t, n)
            like.co = pybird.Common(Nl=3, No=2, kmax=0.3, **U.scales(g)[U.TRACERS.index(t)])
        A, B = U.CROSS.get(t, (t, t))
        bsA = [p[A + "_b1"], p[A + "_b2"], 0.0, p[A + "_b4"], 0.0, 0.0, 0.0]
        bsB = [p[B + "_b1"], p[B + "_b2"], 0.0, p[B + "_b4"], 0.0, 0.0, 0.0] if t in U.CROSS else None
        plk = reduce_Plk(like, bsA, bsB).sum()
        assert relerr(plk, g[t + "_plk"][: plk.shape[0]]) < TOL, t
        nz = np.abs(g[t + "_plk"][: plk.shape[0]]) > 1e-3 * np.max(np.abs(g[t + "_plk"][: plk.shape[0]]), axis=-1, keepdims=True)
        assert np.max(np.abs(plk / g[t + "_plk"][: plk.shape[0]] - 1.0)[nz]) < 1e-6, t  # the north-star bar, pointwise


def _engine_with_tracer_operators(g, windows, max_batch):
    """One engine for the three kernels of a likelihood point: folded operators window -> binning [-> chained], padded to one shape
    ([3][24][18]: chained ELG has 2 multipoles and 17 bins)."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    k = g["k"]
    nb = max(g[t + "_kout"].size for t in U.TRACERS)
    ops_host = []
    for t in U.TRACERS:
        _, win = windows[t]
        Bm, keff, _, _ = TB.binning_operator(k, g[t + "_kout"])
        op = TB.compose_operator(3, k.size, Wfold=win.Wfold, binning=Bm, chained=U.CHAINED[t])
        full = np.zeros((3, 3, nb, k.size))
        full[: op.shape[0], :, : op.shape[2]] = op
        ops_host.append(full)
    t0 = U.TRACERS[0]
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(g[t0 + "_DA_AP"]), H_AP=float(g[t0 + "_H_AP"])), max_batch=max_batch)
    ops = [eng.add_operator(o) for o in ops_host]
    return eng, ops, nb


def _inputs(g, scale_walker=None):
    """Pin, f, DA, H of the three entries of one likelihood point; the AP fiducial of the engine is the first tracer's, the others'
    (z_AP = their own z) enter through rescaled DA, H (include/eftbird.h eftb_set_tracers)"""
    t0 = U.TRACERS[0]
    Pin = np.stack([g[t + "_Pin"] for t in U.TRACERS])
    f = np.array([float(g[t + "_f"]) for t in U.TRACERS])
    DA = np.array([float(g[t + "_DA"]) * float(g[t0 + "_DA_AP"]) / float(g[t + "_DA_AP"]) for t in U.TRACERS])
    H = np.array([float(g[t + "_H"]) * float(g[t0 + "_H_AP"]) / float(g[t + "_H_AP"]) for t in U.TRACERS])
    return Pin, f, DA, H


@pytest.mark.parametrize("tag", ["full", "xnost"])
def test_joint_likelihood_on_the_reference_data(golden, windows, tag):
    """EFTLike(tracers=[LRG, ELG, X], chained=[F, T, F], with_binning, jeffreys, marg=...) on DR16 NGC (reference likelihood.py:281-307,
    340-372, 483-549; marginal.py:79-140): batched theory + projection + marginalised log-posterior on the device against the
    reference's ln P (1e-9), full chi2 and best-fit Gaussian parameters (1e-7)."""
    from eftpipe_amd.marginal import MarginalLikelihood, data_index, joint_gaussian_rows

    g = golden("cfg3")
    nW = 2
    eng, ops, nb = _engine_with_tracer_operators(g, windows, max_batch=3 * nW)
    eng.set_tracers(3, ops)
    Pin, f, DA, H = _inputs(g)
    # walker 0: the fixture's point; walker 1: a perturbed one (must not leak into walker 0)
    Pin2, f2, DA2, H2 = np.concatenate([Pin, 1.03 * Pin]), np.concatenate([f, 0.99 * f]), np.concatenate([DA, 1.01 * DA]), np.concatenate([H, H])
    templ = eng.eval_batch(Pin2, f2, DA2, H2)
    assert templ.shape == (6, 3, 24, nb)
    rows24 = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))
    for i, t in enumerate(U.TRACERS):
        want = U.final_templates(g, t)
        no, nx = want["P11l"].shape[0], want["P11l"].shape[-1]
        for n, sl in rows24.items():
            assert relerr(templ[i][:no, sl, :nx], want[n]) < TOL, (t, n)
        assert np.all(templ[i][no:] == 0.0) and np.all(templ[i][..., nx:] == 0.0)  # padding of the chained / shorter tracer
    index = np.concatenate([data_index([int(l) for l in g[t + "_ls"]], U.masks(g, t), nb, tracer=i, nl=3) for i, t in enumerate(U.TRACERS)])
    names = [str(n) for n in g[tag + "_names"]]
    nG = len(names)
    like = MarginalLikelihood(eng, index, g["data_vector"], g["invcov"], np.zeros(nG), np.full(nG, np.inf), jeffreys=True)
    p = U.params(g)
    rows = np.concatenate([joint_gaussian_rows(U.bases(), fw, p, names, U.scales(g)) for fw in (f, 0.99 * f)])
    logp, full, best = like.logp(rows, return_best=True)
    assert np.isclose(logp[0], g[tag + "_logp"], rtol=1e-9), (logp[0], g[tag + "_logp"])
    assert np.isclose(full[0], g[tag + "_fullchi2"], rtol=1e-8)
    assert relerr(best[0][None], g[tag + "_best"][None]) < 1e-7
    assert logp[1] != logp[0]
    # theory + likelihood in one call
    lp2, full2, best2 = like.eval_logp(Pin2, f2, DA2, H2, rows, return_best=True)
    assert np.array_equal(lp2, logp) and np.array_equal(best2, best)
    # without the Jeffreys option: + ln det(F2 / 2 pi)
    like_nj = MarginalLikelihood(eng, index, g["data_vector"], g["invcov"], np.zeros(nG), np.full(nG, np.inf), jeffreys=False)
    assert np.isclose(like_nj.logp(rows)[0], g[tag + "_logp_nojeffreys"], rtol=1e-9)
    eng.close()


# ------------------------------------------------------------------------------------------------------------------------------------
# cfg 3 on the BASELINE grid: Nk = 512 with the production windows (tests/golden/cfg3_nk512.npz, tools/make_fixtures.py cfg3_nk512)
# ------------------------------------------------------------------------------------------------------------------------------------
def _co512(g, sc, No=3):
    from eftpipe_amd import pybird

    co = pybird.Common(Nl=3, No=No, kmax=0.3, **sc)
    co.k, co.Nk = g["k"], g["k"].size
    co.kr = co.k[0.02 <= co.k]
    co.Nkr = co.kr.size
    co.Nklow = co.Nk - co.Nkr
    return co


@pytest.fixture(scope="module")
def windows512(golden):
    """Window plugin objects of the three tracers on the 512-point survey grid at accboost 4 / windowk 0.1: eftb_window_precompute at
    Np = 1540 x Nk = 512 (window_bessel_kernel, three K = 2473 GEMMs per tracer, window_maskdp_kernel, the fold GEMM)"""
    import os

    from eftpipe_amd.window import Window

    g = golden("cfg3_nk512")
    out = {}
    for t, sc in zip(U.TRACERS, U.scales(g)):
        co = _co512(g, sc)
        out[t] = (co, Window(window_configspace_file=os.path.join(U.GOLD, "win_NGC_%s_sQ024.npy" % t.split("_")[0]), co=co, load=False, save=False,
                             accboost=int(g["accboost"]), windowk=float(g["windowk"])))
    return out


@pytest.mark.parametrize("t", U.TRACERS)
def test_nk512_window_precompute_at_production_settings(golden, windows512, t):
    """the device precompute at the shipped accuracy on the BASELINE grid against the reference's own Waldk (two k rows and the p sums,
    1e-9) and its convolved templates (reference window.py:27-33, 262-415)"""
    from types import SimpleNamespace

    g = golden("cfg3_nk512")
    co, win = windows512[t]
    assert np.array_equal(win.p, g["window_p"]) and win.p.size == 1540 and win.Waldk.shape == (3, 3, 512, 1540)
    assert relerr(win.Waldk[:, :, 100, :], g[t + "_Waldk_k100"]) < 1e-9 and relerr(win.Waldk[:, :, 411, :], g[t + "_Waldk_k411"]) < 1e-9
    assert relerr(win.Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
    bird = SimpleNamespace(co=co, f=float(g[t + "_f"]), Picc=np.zeros((3, 512)), PctNNLOl=None, **{n: g[f"{t}_ap_{n}"].copy() for n in U.NAMES})
    win.Window(bird)
    for n in U.NAMES:
        assert relerr(getattr(bird, n), g[f"{t}_window_{n}"]) < TOL, n


def test_nk512_three_tracer_batch_against_reference(golden, windows512):
    """BASELINE cfg 3 as one batched run: the three kernels of a likelihood point (own P_lin, own AP fiducial, own production window,
    binning onto the data k, ELG chained) at Nl = 3, Nk = 512 through eftb_eval_batch with per-tracer folded operators; templates
    against the reference at 1e-8, P_l pointwise at 1e-6 (yaml :6-27, 63-70; theory.py:557-609)."""
    from eftpipe_amd import pybird
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("cfg3_nk512")
    k = g["k"]
    nb = max(g[t + "_kout"].size for t in U.TRACERS)
    t0 = U.TRACERS[0]
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, APst=True, DA_AP=float(g[t0 + "_DA_AP"]), H_AP=float(g[t0 + "_H_AP"])), max_batch=6)
    ops = []
    for t in U.TRACERS:
        _, win = windows512[t]
        Bm, keff, _, _ = TB.binning_operator(k, g[t + "_kout"])
        assert relerr(keff[None], g[t + "_keff"][None]) < 1e-13
        op = TB.compose_operator(3, k.size, Wfold=win.Wfold, binning=Bm, chained=U.CHAINED[t])
        full = np.zeros((3, 3, nb, k.size))
        full[: op.shape[0], :, : op.shape[2]] = op
        ops.append(eng.add_operator(full))
    # (a) the AP-stage templates of every tracer, before any operator
    Pin, f, DA, H = _inputs(g)
    templ = eng.eval_batch(Pin, f, DA, H)
    rows24 = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))
    for i, t in enumerate(U.TRACERS):
        for n, sl in rows24.items():
            assert relerr(templ[i][:, sl], g[f"{t}_ap_{n}"]) < TOL, (t, n)
    # (b) the whole chain per tracer in one batched run of two walkers, with the bias contraction
    eng.set_tracers(3, ops)
    p = U.params(g)
    sc = U.scales(g)
    bias = []
    for i, t in enumerate(U.TRACERS):
        A, B = U.CROSS.get(t, (t, t))
        bsA = [p[A + "_b1"], p[A + "_b2"], 0.0, p[A + "_b4"], 0.0, 0.0, 0.0]
        bsB = [p[B + "_b1"], p[B + "_b2"], 0.0, p[B + "_b4"], 0.0, 0.0, 0.0] if t in U.CROSS else None
        bias.append(bias_row(float(f[i]), bsA, bsB, (0.0, 0.0, 0.0), **sc[i]))
    bias = np.stack(bias)
    Pin2, f2, DA2, H2 = np.concatenate([Pin, 1.02 * Pin]), np.concatenate([f, f]), np.concatenate([DA, 0.99 * DA]), np.concatenate([H, 1.01 * H])
    templ, plk = eng.eval_batch(Pin2, f2, DA2, H2, bias=np.concatenate([bias, bias]))
    assert templ.shape == (6, 3, 24, nb) and plk.shape == (6, 3, nb)
    for i, t in enumerate(U.TRACERS):
        want = U.final_templates(g, t)
        no, nx = want["P11l"].shape[0], want["P11l"].shape[-1]
        for n, sl in rows24.items():
            assert relerr(templ[i][:no, sl, :nx], want[n]) < TOL, (t, n)
        ref = g[t + "_plk"][:no]
        assert relerr(plk[i][:no, :nx], ref) < TOL, t
        nz = np.abs(ref) > 1e-3 * np.max(np.abs(ref), axis=-1, keepdims=True)
        assert np.max(np.abs(plk[i][:no, :nx] / ref - 1.0)[nz]) < 1e-6, t       # the north-star bar, pointwise
    assert not np.array_equal(plk[3], plk[0])
    # (c) the same batched run as a direct-P_l run (round 4: the bias contraction first, ONE row per multipole through resummation, AP and the
    # per-tracer window -> binning -> chained operators): the reference's P_l again, and the template path's to summation order
    eng.set_plk_direct(True)
    plk_d = eng.eval_batch(Pin2, f2, DA2, H2, bias=np.concatenate([bias, bias]), templates=False)
    eng.set_plk_direct(False)
    assert plk_d.shape == plk.shape and not np.array_equal(plk_d, plk)   # (the option took effect)
    for i, t in enumerate(U.TRACERS):
        ref = g[t + "_plk"]
        no, nx = ref.shape[0] if ref.shape[0] < 3 else U.final_templates(g, t)["P11l"].shape[0], U.final_templates(g, t)["P11l"].shape[-1]
        ref = ref[:no]
        assert relerr(plk_d[i][:no, :nx], ref) < TOL, t
        nz = np.abs(ref) > 1e-3 * np.max(np.abs(ref), axis=-1, keepdims=True)
        assert np.max(np.abs(plk_d[i][:no, :nx] / ref - 1.0)[nz]) < 1e-6, t
    assert relerr(plk_d, plk) < 1e-9
    eng.close()
