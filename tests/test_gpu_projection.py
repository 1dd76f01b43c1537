"""GPU: window / binning / chained as device operators -- the mirror classes one by one (reference
theory.py:584-604 order) and the single folded pipeline operator of the batched path."""
import os

import numpy as np
import pytest

from conftest import relerr
from eftpipe_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-8
WIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "win_NGC_LRG_sQ024.npy")
NAMES = ("P11l", "Pctl", "Ploopl", "Pstl")


def test_window_binning_chained_mirrors(golden, tmp_path):
    from eftpipe_amd import pybird
    from eftpipe_amd.binning import Binning
    from eftpipe_amd.chained import Chained
    from eftpipe_amd.parambasis import reduce_Plk
    from eftpipe_amd.transformer import BirdCopier
    from eftpipe_amd.window import Window

    g = golden("caseC")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    nl, rs = pybird.NonLinear(co=co), pybird.Resum(co=co)
    ap = pybird.APeffect(Om_AP=synth.OM_AP, z_AP=0.7, co=co, APst=True)
    cache = tmp_path / "win_cache.npy"
    win = Window(window_fourier_file=cache, window_configspace_file=WIN, co=co, load=True, save=True)
    assert cache.exists() and cache.with_suffix(".json").exists()      # reference cache format written
    win2 = Window(window_fourier_file=cache, co=co, load=True)          # ... and read back
    assert np.array_equal(win2.Wal, win.Wal)
    binning = Binning(kout=g["kout"], co=co)
    assert np.allclose(binning.keff, g["keff"], rtol=1e-13)

    bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    nl.PsCf(bird)
    bird.setPsCfl()
    rs.Ps(bird)
    snap = BirdCopier().transform(bird)
    ap.AP(bird)
    assert relerr(snap.Ploopl, g["resum_Ploopl"]) < TOL
    win.Window(bird)
    for n in NAMES:
        assert relerr(getattr(bird, n), g["window_" + n]) < TOL, n
    assert relerr(reduce_Plk(bird, list(g["bsA"]), es=tuple(g["es"])).sum(), g["plk_auto"]) < TOL
    binned = binning.transform(bird)
    for n in NAMES + ("Picc",):
        assert relerr(getattr(binned, n), g["binned_" + n]) < TOL, n
    assert relerr(reduce_Plk(binned, list(g["bsA"]), es=tuple(g["es"])).sum(), g["plk_binned_auto"]) < TOL
    ch = Chained().transform(binned)
    for n in NAMES + ("Picc",):
        assert getattr(ch, n).shape == g["chained_" + n].shape
        assert relerr(getattr(ch, n), g["chained_" + n]) < TOL, n
    # cross-spectrum style contraction A x B (reference parambasis.py:69-126 with kmB/krB/ndB)
    cox = pybird.Common(Nl=3, kmA=0.7, krA=0.25, ndA=4.5e-5, kmB=0.6, krB=0.3, ndB=2.3e-4)
    binned.co = cox
    assert relerr(reduce_Plk(binned, list(g["bsA"]), list(g["bsB"]), tuple(g["es"])).sum(), g["plk_cross"]) < TOL


def test_folded_pipeline_operator_batched(golden):
    """One operator = chained o binning o window, applied inside eval_batch for several cosmologies."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    k = g["k"]
    tab = np.load(WIN)
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wfold, _ = TB.window_fold(k, Wal, p)
    Bm, keff, _, _ = TB.binning_operator(k, g["kout"])
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=4)
    op_full = eng.add_operator(TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, chained=True))
    op_bin = eng.add_operator(TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm))
    f = float(g["f"])
    Pin = np.stack([g["Pin"], 1.05 * g["Pin"], g["Pin"]])
    bias = np.tile(bias_row(f, list(g["bsA"]), None, tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5), (3, 1))
    eng.set_pipeline_operator(op_full)
    templ, plk = eng.eval_batch(Pin, f, float(g["DA"]), float(g["H"]), bias=bias)
    assert templ.shape == (3, 2, 24, len(g["kout"])) and plk.shape == (3, 2, len(g["kout"]))
    for i in (0, 2):
        assert relerr(templ[i][:, 9:21], g["chained_Ploopl"]) < TOL
        assert relerr(templ[i][:, 0:3], g["chained_P11l"]) < TOL
    assert np.max(np.abs(templ[1] - templ[0])) > 0
    eng.set_pipeline_operator(op_bin)
    templ, plk = eng.eval_batch(Pin, f, float(g["DA"]), float(g["H"]), bias=bias)
    assert relerr(plk[0], g["plk_binned_auto"]) < TOL and relerr(plk[2], g["plk_binned_auto"]) < TOL
    eng.set_pipeline_operator(-1)
    templ = eng.eval_batch(Pin, f, float(g["DA"]), float(g["H"]))
    assert templ.shape == (3, 3, 24, k.size) and relerr(templ[0][:, 9:21], g["ap_Ploopl"]) < TOL
    eng.close()


def test_interpolation_operator_then_marginalised_logp(golden):
    """The `with_interp` data path on the device: PlkInterpolator as the pipeline operator, P_l at the data k,
    then the marginalised log-posterior of the interpolated templates (rank 1) -- against the oracle on the host."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.marginal import MarginalLikelihood, data_index, gaussian_rows
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig
    from oracle import marginal as M
    from oracle.engine import plk_interpolate

    g = golden("caseD")
    k = g["k"]
    kdata = np.arange(0.03, 0.2, 0.01)
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=2)
    op = eng.add_operator(np.einsum("al,xk->alxk", np.eye(3), TB.interp_operator(k, kdata)))
    eng.set_pipeline_operator(op)
    f = float(g["f"])
    bias = bias_row(f, list(g["bsA"]), None, tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5)[None]
    templ, plk = eng.eval_batch(g["Pin"][None], f, float(g["DA"]), float(g["H"]), bias=bias)
    want = plk_interpolate([0, 2, 4], k, g["plk_auto"], [0, 2, 4], kdata)
    assert relerr(plk[0], want) < TOL
    # likelihood on the interpolated templates (still resident on the device)
    nx = kdata.size
    ls, masks = [0, 2], {0: slice(0, nx), 2: slice(2, nx - 3)}
    index = data_index(ls, masks, nx)
    rng = np.random.default_rng(5)
    b1, b2, b4 = g["bsA"][0], g["bsA"][1], g["bsA"][3]
    rows = gaussian_rows(f, (b1, b2, b4), None, 0.7, 0.25, 4.5e-5)
    V = np.einsum("gr,lrx->glx", rows, templ[0]).reshape(rows.shape[0], -1)[:, index]
    sig = 0.04 * np.abs(V[0]) + 20.0
    D = V[0] + sig * rng.normal(size=index.size)
    C = np.diag(1.0 / sig**2)
    loc, scale = np.zeros(7), np.array([2.0, 2.0, 4.0, 4.0, 2.0, 2.0, 2.0])
    like = MarginalLikelihood(eng, index, D, C, loc, scale)
    logp, full, best = like.logp(rows[None], return_best=True)
    w = M.marginalized_logp(V[1:], V[0], D, C, loc, scale, return_best=True)
    assert np.isclose(logp[0], w[0], rtol=1e-10) and np.isclose(full[0], w[1], rtol=1e-9) and relerr(best, w[2][None]) < 1e-8
    # theory + likelihood in ONE C-ABI call (eftb_eval_logp_batch): only inputs and one float per walker cross PCIe
    Pin2 = np.stack([g["Pin"], 1.1 * g["Pin"]])
    rows2 = np.stack([rows, gaussian_rows(f, (b1 + 0.1, b2, b4), None, 0.7, 0.25, 4.5e-5)])
    lp2, full2, best2 = like.eval_logp(Pin2, f, float(g["DA"]), float(g["H"]), rows2, return_best=True)
    assert lp2[0] == logp[0] and full2[0] == full[0] and np.array_equal(best2[0], best[0])
    pinned = eng.pinned_empty((2, 3, 24, nx))  # page-locked landing buffer
    t2 = eng.eval_batch(Pin2, f, float(g["DA"]), float(g["H"]), out=pinned)
    assert t2 is pinned and np.array_equal(t2[0], templ[0])
    V1 = np.einsum("gr,lrx->glx", rows2[1], t2[1]).reshape(rows2.shape[1], -1)[:, index]
    assert np.isclose(lp2[1], M.marginalized_logp(V1[1:], V1[0], D, C, loc, scale), rtol=1e-10)
    # P_l only (templ = NULL)
    plk_only = eng.eval_batch(Pin2, f, float(g["DA"]), float(g["H"]), bias=np.repeat(bias, 2, axis=0), templates=False)
    assert np.array_equal(plk_only[0], plk[0])
    with pytest.raises(ValueError):
        eng.eval_batch(Pin2, f, float(g["DA"]), float(g["H"]), templates=False)
    with pytest.raises(ValueError):
        like.eval_logp(Pin2, f, float(g["DA"]), float(g["H"]), rows2[:1])
    del t2, pinned
    eng.close()


def test_rccl_gather_single_rank(golden):
    """World size 1 exercises the whole RCCL code path short of the wire: lazy dlopen, unique id,
    ncclCommInitRank, gather into the root buffer, copy-out."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.engine import Engine, comm_unique_id
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseA")
    eng = Engine(EngineConfig(Nl=2), max_batch=4)
    uid = comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(1, 0, uid)
    f = float(g["f"])
    bias = np.tile(bias_row(f, list(g["bsA"]), None, tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5), (3, 1))
    eng.load_inputs(np.stack([g["Pin"]] * 3), f, bias=bias)
    eng.run(eng.full_mask(reduce=True), 3)
    out = eng.gather_plk(3, root=0, to_host=True)
    assert out.shape == (1, 3, 2, 50)
    assert relerr(out[0, 0], g["plk_auto"]) < TOL and np.array_equal(out[0, 0], out[0, 2])
    # asynchronous gathers (side stream, P_l snapshot) pipelined with further steps: the last one wins, nothing is torn
    for scale in (2.0, 3.0, 0.5):
        eng.put("BIAS", bias * scale)
        eng.run(eng.full_mask(reduce=True), 3, sync=False)
        eng.gather_plk(3, root=0)
    eng.put("BIAS", bias)
    eng.run(eng.full_mask(reduce=True), 3, sync=False)
    last = eng.gather_plk(3, root=0, to_host=True)
    assert np.array_equal(last, out)
    # every exchange leaves its block in page-locked host memory (DMA behind the exchange): copies and zero-copy views of the last four
    blocks = []
    for scale in (1.0, 2.0, 3.0, 4.0):
        eng.put("BIAS", bias * scale)
        eng.run(eng.full_mask(reduce=True), 3, sync=False)
        eng.gather_plk(3, root=0)
        eng.sync()
        blocks.append(eng.get("PLK", (3, 2, 50)))
    for back in range(4):
        want = blocks[3 - back]
        view = eng.fetch_gathered(3, back=back, copy=False)
        assert view.shape == (1, 3, 2, 50) and not view.flags.writeable
        assert np.array_equal(view[0], want) and np.array_equal(eng.fetch_gathered(3, back=back)[0], want)
    with pytest.raises(L.EftbError):
        eng.fetch_gathered(3, back=16)
    del view
    eng.close()


def test_cfg3_shape_nk512_window_chained_cross(golden):
    """BASELINE cfg 3 shape: Nl=3, Nk=512, IR-resum + AP + window, chained multipoles, cross-tracer biases."""
    from eftpipe_amd import pybird
    from eftpipe_amd.chained import Chained
    from eftpipe_amd.parambasis import reduce_Plk
    from eftpipe_amd.window import Window

    g = golden("caseG")
    co = pybird.Common(Nl=3, No=2, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5, kmB=0.6, krB=0.3, ndB=2.3e-4)
    co.k, co.Nk = g["k"], g["k"].size
    co.kr = co.k[0.02 <= co.k]
    co.Nkr = co.kr.size
    co.Nklow = co.Nk - co.Nkr
    nl, rs = pybird.NonLinear(co=co), pybird.Resum(co=co)
    ap = pybird.APeffect(Om_AP=synth.OM_AP, z_AP=0.7, co=co, APst=True)
    win = Window(window_configspace_file=WIN, co=co, load=False, save=False)
    assert np.array_equal(win.p, g["window_p"])
    assert relerr(win.Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
    bird = pybird.Bird(g["kin"], g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]), 0.7, co=co)
    nl.PsCf(bird)
    bird.setPsCfl()
    rs.Ps(bird)
    ap.AP(bird)
    for n in NAMES:
        assert relerr(getattr(bird, n), g["ap_" + n]) < TOL, n
    win.Window(bird)
    for n in NAMES:
        assert relerr(getattr(bird, n), g["window_" + n]) < TOL, n
    ch = Chained().transform(bird)
    for n in NAMES:
        assert relerr(getattr(ch, n), g["chained_" + n]) < TOL, n
    plk = reduce_Plk(ch, list(g["bsA"]), list(g["bsB"]), tuple(g["es"])).sum()
    assert plk.shape == (2, 512) and relerr(plk, g["plk_chained_cross"]) < TOL


def test_cfg5_shape_nk2048_window_binning(golden):
    """BASELINE cfg 5 shape: Nk=2048, IR-resum + AP + window + binning folded into one device operator.
    Templates before the projection are pinned by the reference (caseF); the projection itself is checked
    against a NumPy application of the same host-built operator (the reference's window at Nk=2048 is too
    heavy to run in the build container)."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseF")
    k = g["k"]
    tab = np.load(WIN)
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wfold, _ = TB.window_fold(k, Wal, p)
    kout = np.arange(0.025, 0.2, 0.01)
    Bm, keff, _, _ = TB.binning_operator(k, kout)
    op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm)
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=2)
    templ = eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), float(g["f"]), float(g["DA"]), float(g["H"]))
    for n, sl in dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24)).items():
        assert relerr(templ[0][:, sl], g["ap_" + n]) < TOL, n
    oid = eng.add_operator(op)
    eng.set_pipeline_operator(oid)
    proj = eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), float(g["f"]), float(g["DA"]), float(g["H"]))
    assert proj.shape == (2, 3, 24, len(kout))
    want = np.einsum("alxk,lrk->arx", op, templ[0])
    assert relerr(proj[0], want) < 1e-10 and np.array_equal(proj[0], proj[1])
    eng.close()


def test_fiber_collision_mirror_and_folded_operator(golden):
    """FiberCollision.fibcolWindow as a device operator (drop-in call, reference pybird.py:1760-1810) and inside
    the folded pipeline operator window -> fibre -> binning with the stochastic rows taking the fibre-less matrix (fiberst=False)."""
    from types import SimpleNamespace

    from eftpipe_amd import pybird
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig
    from oracle import fiber as F

    g, c = golden("fiber"), golden("caseC")
    k, fs, Dfc, kt = c["k"], float(g["fs"]), float(g["Dfc"]), float(g["ktrust"])
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    for fiberst in (False, True):
        fib = pybird.FiberCollision(fs=fs, Dfc=Dfc, ktrust=kt, fiberst=fiberst, co=co)
        bird = SimpleNamespace(co=co, **{n: c["window_" + n].copy() for n in NAMES})
        fib.fibcolWindow(bird)
        tag = "st_" if fiberst else ""
        for n in NAMES:
            assert relerr(getattr(bird, n), g["fiber_" + tag + n]) < 1e-11, (fiberst, n)
    assert relerr(fib.dPuncorr(k, fs=fs, Dfc=Dfc), g["dPuncorr"]) < 1e-14
    sub = c["window_Ploopl"][:, :2]
    assert relerr(fib.dPcorr(k, k, sub, ktrust=kt, fs=fs, Dfc=Dfc).reshape(6, -1), F.dpcorr(k, k, sub, 3, ktrust=kt, fs=fs, Dfc=Dfc).reshape(6, -1)) < 1e-12
    # batched: AP-stage templates -> window -> fibre -> binning in ONE operator; Pstl skips the fibre matrix
    tab = np.load(WIN)
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wfold, _ = TB.window_fold(k, Wal, p)
    Bm, _, _, _ = TB.binning_operator(k, c["kout"])
    Fm = TB.fiber_operator(k, 3, fs, Dfc, kt)
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(c["DA_AP"]), H_AP=float(c["H_AP"])), max_batch=3)
    op = eng.add_operator(TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, fiber=Fm),
                          stochastic=TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm))
    eng.set_pipeline_operator(op)
    Pin = np.stack([c["Pin"], 0.9 * c["Pin"], c["Pin"]])
    templ = eng.eval_batch(Pin, float(c["f"]), float(c["DA"]), float(c["H"]))
    binned = lambda a: np.einsum("bx,lnx->lnb", Bm, a)
    for i in (0, 2):
        assert relerr(templ[i][:, 0:3], binned(g["fiber_P11l"])) < TOL
        assert relerr(templ[i][:, 3:9], binned(g["fiber_Pctl"])) < TOL
        assert relerr(templ[i][:, 9:21], binned(g["fiber_Ploopl"])) < TOL
        assert relerr(templ[i][:, 21:24], c["binned_Pstl"]) < TOL      # no fibre correction on the stochastic rows
    assert np.max(np.abs(templ[1] - templ[0])) > 0
    with pytest.raises(Exception):
        eng.add_operator(np.zeros((3, 3, 5, k.size)), stochastic=np.zeros((3, 3, 6, k.size)))
    eng.close()


def test_window_matrix_on_device(golden):
    """WindowMatrix.Window (reference window.py:566-586) as a device operator, window_st on and off"""
    from types import SimpleNamespace

    from eftpipe_amd import pybird
    from eftpipe_amd import window as W

    g, c = golden("wmat"), golden("caseC")
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    m = W.to_window_matrix(g["stacked"].astype(np.float64), W.PInfo((0, 2, 4), 0, 0.4, 400), W.PInfo((0, 1, 2, 3, 4), 0, 0.4, 40), (0, 2, 4), co.k.max(),
                           tuple(g["ells"]), float(g["kmin"]), float(g["kmax"]))
    for st in (False, True):
        wm = W.WindowMatrix(m, W.PolesInfo(3, 0, co.k.max(), m.shape[3]), W.PolesInfo(2, float(g["kmin"]), float(g["kmax"]), m.shape[2]), co=co,
                            window_st=st)
        bird = SimpleNamespace(co=co, Picc=np.zeros((3, 50)), **{n: c["ap_" + n].copy() for n in NAMES})
        wm.Window(bird)
        tag = "st_" if st else ""
        for n in NAMES:
            assert relerr(getattr(bird, n), g[tag + "wm_" + n]) < TOL, (st, n)
        assert bird.Picc.shape == (2, m.shape[2])


def test_three_tracers_per_likelihood_point(golden):
    """BASELINE cfg 3 / EFTLike(tracers=[LRG, ELG, X]) (reference likelihood.py:483-549): three entries per walker, each with its own
    projection operator (ELG chained, X without window), joint data vector and covariance, 17 jointly marginalised parameters."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.marginal import MarginalLikelihood, data_index
    from eftpipe_amd.parambasis import gaussian_rows
    from eftpipe_amd.tables import EngineConfig
    from oracle import marginal as M

    g = golden("caseC")
    k, nb = g["k"], len(g["kout"])
    tab = np.load(WIN)
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wfold, _ = TB.window_fold(k, Wal, p)
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    opE = np.zeros((3, 3, nb, k.size))
    opE[:2] = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, chained=True)  # chained: 2 multipoles, padded with a zero third
    ops_host = [TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm), opE, TB.compose_operator(3, k.size, binning=Bm)]
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=6)
    ops = [eng.add_operator(o) for o in ops_host]
    nW, ntr = 2, 3
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    Pin = np.stack([g["Pin"] * (1.0 + 0.04 * w) * (1.0 - 0.1 * t) for w in range(nW) for t in range(ntr)])
    f = np.array([f0 * (1.0 + 0.02 * t) for w in range(nW) for t in range(ntr)])
    DA = np.array([DA0 * (1.0 + 0.01 * t + 0.005 * w) for w in range(nW) for t in range(ntr)])
    H = np.array([H0 * (1.0 - 0.01 * t) for w in range(nW) for t in range(ntr)])
    # reference for the templates: the same entries through one operator at a time
    want_t = np.empty((nW * ntr, 3, 24, nb))
    for t in range(ntr):
        eng.set_tracers(1)
        eng.set_pipeline_operator(ops[t])
        want_t[t::ntr] = eng.eval_batch(Pin[t::ntr], f[t::ntr], DA[t::ntr], H[t::ntr])
    eng.set_pipeline_operator(-1)
    eng.set_tracers(ntr, ops)
    templ = eng.eval_batch(Pin, f, DA, H)
    assert templ.shape == (nW * ntr, 3, 24, nb) and relerr(templ.reshape(-1, nb), want_t.reshape(-1, nb)) < 1e-13
    assert np.all(templ[1::ntr, 2] == 0.0)  # the padded multipole of the chained tracer
    # joint likelihood: parameters 0-6 LRG, 7-13 ELG (b3 cct cr1 cr2 ce0 cemono cequad), 14-16 the cross spectrum's stochastic terms
    ngL, ngE = (2.1, 0.5, 0.3), (1.3, -0.2, 0.6)
    nG = 17
    rows = np.zeros((nW * ntr, nG + 1, 24))
    for w in range(nW):
        eL, eE, eX = w * ntr, w * ntr + 1, w * ntr + 2
        rL = gaussian_rows(f[eL], ngL, None, 0.7, 0.25, 4.5e-5)
        rE = gaussian_rows(f[eE], ngE, None, 0.7, 0.25, 2.3e-4)
        rX = gaussian_rows(f[eX], ngL, ngE, 0.7, 0.25, 4.5e-5, 0.6, 0.3, 2.3e-4)  # A_(b3 cct cr1 cr2), B_(...), X_(ce0 cemono cequad)
        rows[eL, 0], rows[eL, 1:8] = rL[0], rL[1:]
        rows[eE, 0], rows[eE, 8:15] = rE[0], rE[1:]
        rows[eX, 0] = rX[0]
        rows[eX, 1:5], rows[eX, 8:12], rows[eX, 15:18] = rX[1:5], rX[5:9], rX[9:12]
    index = np.concatenate([data_index([0, 2, 4], {0: slice(0, 16), 2: slice(2, 14), 4: slice(1, 9)}, nb, tracer=0, nl=3),
                            data_index([0, 2], {0: slice(1, 15), 2: slice(3, 12)}, nb, tracer=1, nl=3),
                            data_index([0, 2, 4], {0: slice(0, 12), 2: slice(0, 12), 4: slice(4, 8)}, nb, tracer=2, nl=3)])
    V = []
    for w in range(nW):
        blocks = [np.einsum("gr,lrx->glx", rows[w * ntr + t], templ[w * ntr + t]) for t in range(ntr)]
        V.append(np.concatenate(blocks, axis=1).reshape(nG + 1, -1)[:, index])
    rng = np.random.default_rng(4)
    nd = index.size
    sig = 0.04 * np.abs(V[0][0]) + 20.0
    D = V[0][0] + sig * rng.normal(size=nd)
    A = rng.normal(size=(nd, nd)) * 0.03
    C = np.linalg.inv(np.diag(sig) @ (np.eye(nd) + A @ A.T) @ np.diag(sig))
    C = 0.5 * (C + C.T)
    loc, scale = rng.normal(scale=0.2, size=nG), rng.uniform(1.0, 4.0, size=nG)
    like = MarginalLikelihood(eng, index, D, C, loc, scale)
    logp, full, best = like.logp(rows, return_best=True)
    assert logp.shape == (nW,) and best.shape == (nW, nG)
    for w in range(nW):
        ww = M.marginalized_logp(V[w][1:], V[w][0], D, C, loc, scale, return_best=True)
        assert np.isclose(logp[w], ww[0], rtol=1e-10) and np.isclose(full[w], ww[1], rtol=1e-9), w
        assert relerr(best[w][None], ww[2][None]) < 1e-7, w
    # theory + likelihood in one call for the grouped batch
    lp2 = like.eval_logp(Pin, f, DA, H, rows)
    assert lp2.shape == (nW,) and np.array_equal(lp2, logp)
    with pytest.raises(Exception):
        eng.run(eng.full_mask(), 5)  # not a multiple of the three tracers
    eng.close()


@pytest.mark.parametrize("case", ["caseC", "caseG"])
def test_window_precompute_device(golden, case):
    """W_al(k, p), the masked dp-weighted Waldk and the folded operator from eftb_window_precompute against the
    reference's own Waldk (window.py:262-359; fixtures hold one k row and the p sums) and against the host table builder."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.window import window_matrix_device

    g = golden(case)
    k = g["k"]
    tab = np.load(WIN)
    timing = {}
    Wal, p, Waldk, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, timing=timing)
    assert np.array_equal(p, g["window_p"]) if "window_p" in g else True
    assert relerr(Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
    if "window_Waldk_sum_p" in g:
        assert relerr(Waldk.sum(axis=-1), g["window_Waldk_sum_p"]) < 1e-9
    Wal_h, p_h = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
    Wfold_h, Waldk_h = TB.window_fold(k, Wal_h, p_h)
    assert np.array_equal(p, p_h)
    for a in range(3):
        for l in range(3):
            assert relerr(Wal[a, l], Wal_h[a, l]) < 1e-11, (a, l)
            assert relerr(Wfold[a, l], Wfold_h[a, l]) < 1e-11, (a, l)
    assert np.array_equal(Waldk == 0.0, Waldk_h == 0.0)          # same band
    assert relerr(Waldk, Waldk_h) < 1e-11
    # no mask, Na < Nl
    Wal2, _, Waldk2, Wfold2 = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 2, 3, withmask=False)
    Wfold2_h, Waldk2_h = TB.window_fold(k, Wal_h[:2], p_h, withmask=False)
    assert relerr(Wal2, Wal_h[:2]) < 1e-11 and relerr(Waldk2, Waldk2_h) < 1e-11 and relerr(Wfold2, Wfold2_h) < 1e-11
    print(case, "window precompute:", timing)


def test_window_precompute_rejects_bad_arguments():
    from eftpipe_amd import _lib as L
    from eftpipe_amd.window import window_matrix_device

    tab = np.load(WIN)
    k = np.linspace(0.01, 0.3, 16)
    with pytest.raises(L.EftbError):
        window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 2)          # Na > Nl
    with pytest.raises(L.EftbError):
        window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, windowk=0.0)


def test_window_with_integral_constraint_on_device(golden, tmp_path):
    """Window(icc=IntegralConstraint(...)).Window(bird) (reference window.py:393-406): W P - W_ic P as ONE device operator, Picc -= PSN;
    against the oracle's restatement (PARITY UNPINNED for the ICC matrix itself, see oracle/icc.py), window_st on and off."""
    from types import SimpleNamespace

    from eftpipe_amd import pybird
    from eftpipe_amd.icc import IntegralConstraint
    from eftpipe_amd.window import Window
    from icc_util import ICC_KW, make_icc_files
    from oracle import icc as O

    c = golden("caseC")
    sn, ic, s, xi, s1 = make_icc_files(tmp_path)
    co = pybird.Common(Nl=3, kmax=0.3, kmA=0.7, krA=0.25, ndA=4.5e-5)
    icc = IntegralConstraint(Pshot=1500.0, icc_configspace_SN_file=sn, icc_configspace_IC_file=ic, co=co, load=False, save=False, **ICC_KW)
    for st in (True, False):
        win = Window(window_configspace_file=WIN, co=co, load=False, save=False, icc=icc, window_st=st)
        bird = SimpleNamespace(co=co, Picc=np.zeros((3, 50)), PctNNLOl=None, **{n: c["ap_" + n].copy() for n in NAMES})
        win.Window(bird)
        want = O.window_with_icc({n: c["ap_" + n] for n in NAMES} | {"Picc": np.zeros((3, 50))}, co.k, win.p, win.Waldk, icc.p, icc.Waldk, icc.PSN,
                                 window_st=st)
        for n in NAMES:
            assert relerr(getattr(bird, n), want[n]) < TOL, (st, n)
        assert np.array_equal(bird.Picc, -icc.PSN)
        # the correction is visible: the plain window gives something else
        plain = Window(window_configspace_file=WIN, co=co, load=False, save=False, window_st=st)
        b2 = SimpleNamespace(co=co, Picc=np.zeros((3, 50)), PctNNLOl=None, **{n: c["ap_" + n].copy() for n in NAMES})
        plain.Window(b2)
        assert relerr(b2.Ploopl, bird.Ploopl) > 1e-6


def test_cfg5_two_tracers_nk2048_against_reference(golden):
    """BASELINE cfg 5 (Nk = 2048, IR-resummation + AP + window + binning, two tracers per point) through the batched engine: device window
    precompute for the LRG and the ELG window, one folded operator per tracer (ELG chained, padded), eftb_set_tracers(2), against the
    REFERENCE's binned / chained templates (tests/golden/cfg5.npz) -- replaces the self-comparison of the Nk = 2048 shape test."""
    import cfg3_util as U
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig
    from eftpipe_amd.window import window_matrix_device

    g, F = golden("cfg5"), golden("caseF")
    k, nb = F["k"], len(g["kout"])
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    ops_host = []
    for t, chained in (("LRG", False), ("ELG", True)):
        tab = U.window_table(t + "_NGC")
        Wal, p, Waldk, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3)
        assert relerr(Waldk[:, :, 1000, :], g[t + "_Waldk_k1000"]) < 1e-9 and relerr(Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
        op = np.zeros((3, 3, nb, k.size))
        o = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, chained=chained)
        op[: o.shape[0]] = o
        ops_host.append(op)
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(F["DA_AP"]), H_AP=float(F["H_AP"])), max_batch=4)
    ops = [eng.add_operator(o) for o in ops_host]
    eng.set_tracers(2, ops)
    Pin = np.stack([F["Pin"], F["Pin"], 1.05 * F["Pin"], 1.05 * F["Pin"]])     # two likelihood points x two tracers
    f, DA, H = float(F["f"]), float(F["DA"]), float(F["H"])
    bias = np.stack([bias_row(f, list(F["bsA"]), None, tuple(F["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5)] * 4)
    templ, plk = eng.eval_batch(Pin, f, DA, H, bias=bias)
    assert templ.shape == (4, 3, 24, nb)
    rows24 = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))
    for i, (t, chained) in enumerate((("LRG", False), ("ELG", True))):
        for n, sl in rows24.items():
            want = g[f"{t}_{'chained' if chained else 'binned'}_{n}"]
            assert relerr(templ[i][: want.shape[0], sl], want) < TOL, (t, n)
        assert not np.allclose(templ[2 + i], templ[i])   # the second likelihood point (another P_lin) is its own evaluation
    assert np.all(templ[1][2] == 0.0)                                               # padded multipole of the chained tracer
    assert relerr(plk[0], np.einsum("r,lrx->lx", bias[0], templ[0])) < 1e-12
    eng.close()


def test_cfg5_window_at_shipped_accuracy_nk2048(golden):
    """BASELINE cfg 5 at the window accuracy the reference ships (accboost 4, windowk 0.1; yaml :63-65): eftb_window_precompute at
    Nk = 2048 x Np = 1540 (Waldk 227 MB) against the reference's own rows and p sums (1e-9), then the Nk = 2048 engine with window -> binning
    folded into one device operator against the reference's binned templates (tests/golden/cfg5_acc4.npz; the reference needs 11 minutes
    for this precompute)."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig
    from eftpipe_amd.window import window_matrix_device

    g, f = golden("cfg5_acc4"), golden("caseF")
    k = f["k"]
    tab = np.load(WIN)
    timing = {}
    Wal, p, Waldk, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, windowk=float(g["windowk"]), accboost=int(g["accboost"]), timing=timing)
    print("cfg5 accboost-4 window precompute:", timing)
    del Wal
    assert np.array_equal(p, g["window_p"]) and p.size == 1540
    assert relerr(Waldk[:, :, 1000, :], g["LRG_Waldk_k1000"]) < 1e-9 and relerr(Waldk[:, :, 77, :], g["LRG_Waldk_k77"]) < 1e-9
    assert relerr(Waldk.sum(axis=-1), g["LRG_Waldk_sum_p"]) < 1e-9
    del Waldk
    Bm, keff, _, _ = TB.binning_operator(k, g["kout"])
    op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm)
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(f["DA_AP"]), H_AP=float(f["H_AP"])), max_batch=2)
    eng.set_pipeline_operator(eng.add_operator(op))
    proj = eng.eval_batch(np.stack([f["Pin"], f["Pin"]]), float(f["f"]), float(f["DA"]), float(f["H"]))
    rows24 = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))
    for n, sl in rows24.items():
        assert relerr(proj[0][:, sl], g["LRG_binned_" + n]) < TOL, n
    assert np.array_equal(proj[0], proj[1])
    # the same engine on direct-P_l runs (round 4: any k grid -- the moment-form AP kernel on B-spline pieces -- and the PROJECT stage on one row per
    # cosmology): P_l against the reference's binned templates contracted with the bias (reduce_Plk, parambasis.py:42-136)
    from eftpipe_amd.parambasis import bias_row

    bs = [[2.1, 0.6, 0.8, 0.5, -1.8, -1.9, -1.5], [1.3, 0.2, 0.4, 0.9, 0.7, -0.6, 1.1]]
    bias = np.stack([bias_row(float(f["f"]), b_, None, (0.3, 0.1, -0.9), kmA=0.7, krA=0.25, ndA=4.5e-5) for b_ in bs])
    T = np.concatenate([g["LRG_binned_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl")], axis=1)   # [Nl][24][nbins], the engine's row order
    want = np.einsum("wr,lrx->wlx", bias, T)
    args = (np.stack([f["Pin"], f["Pin"]]), float(f["f"]), float(f["DA"]), float(f["H"]))
    plk_t = eng.eval_batch(*args, bias=bias, templates=False)
    eng.set_plk_direct(True)
    plk_d = eng.eval_batch(*args, bias=bias, templates=False)
    assert not np.array_equal(plk_d, plk_t)   # (the option took effect at Nk = 2048, behind a PROJECT stage)
    for got in (plk_t, plk_d):
        assert relerr(got, want) < TOL
        big = np.abs(want) > 1e-3 * np.abs(want).max(axis=-1, keepdims=True)
        assert np.max(np.abs(got - want)[big] / np.abs(want)[big]) < 1e-6
    # a staged step behind the PROJECT stage delivers its (binned) rows into the caller's page-locked array: exactly plk_d.size elements suffice
    dest = eng.pinned_empty(plk_d.shape)
    dest.fill(-1.0)
    Pin2, f2, DA2, H2 = args
    eng.step(eng.full_mask(reduce=True), Pin2, np.full(2, f2), np.full(2, DA2), np.full(2, H2), bias=bias, out=dest)   # (the mask carries PROJECT: the engine has a pipeline operator)
    eng.fetch_previous("PLK", plk_d.shape, back=0, copy=False)
    assert np.array_equal(dest, plk_d)
    eng.close()
