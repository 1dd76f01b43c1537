"""GPU parity of the marginalised log-posterior stage through the C ABI, against the reference-generated
fixture tests/golden/marg.npz and the oracle (oracle/marginal.py)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _engine(g, B):
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    nx = g["binned_P11l"].shape[-1]
    eng = Engine(EngineConfig(Nl=3), max_batch=B)
    eng.set_template_dims(3, nx)
    return eng, nx


@pytest.mark.parametrize("tag", ["auto", "cross"])
def test_marginalised_logp_matches_reference(golden, tag):
    from eftpipe_amd.marginal import MarginalLikelihood, data_index, gaussian_rows
    from oracle import marginal as M

    g = golden("marg")
    B = 5
    eng, nx = _engine(g, B)
    T = np.concatenate([g["binned_P11l"], g["binned_Pctl"], g["binned_Ploopl"], g["binned_Pstl"]], axis=1)  # [3, 24, nx]
    f = float(g["f"])
    co = g[tag + "_co"]
    ng = dict(zip(g[tag + "_ng_names"], g[tag + "_ng_values"]))
    ls = list(g["ls"])
    masks = {l: slice(a, b) for l, (a, b) in zip(ls, g["masks"])}
    index = data_index(ls, masks, nx)
    like = MarginalLikelihood(eng, index, g[tag + "_D"], g[tag + "_invcov"], g[tag + "_loc"], g[tag + "_scale"])
    # walker 0 = the fixture; the others: rescaled templates and shifted b1 (checked against the oracle)
    scales = 1.0 + 0.1 * np.arange(B)
    shifts = 0.05 * np.arange(B)
    templ = np.stack([T * s for s in scales])
    rows, want = [], []
    for i in range(B):
        if tag == "auto":
            A, Bp = (ng["b1"] + shifts[i], ng["b2"], ng["b4"]), None
            rows.append(gaussian_rows(f, A, None, *co[:3]))
        else:
            A, Bp = (ng["A_b1"] + shifts[i], ng["A_b2"], ng["A_b4"]), (ng["B_b1"], ng["B_b2"], ng["B_b4"])
            rows.append(gaussian_rows(f, A, Bp, *co))
        V = np.einsum("gr,lrx->glx", rows[-1], templ[i]).reshape(rows[-1].shape[0], -1)[:, index]
        want.append(M.marginalized_logp(V[1:], V[0], g[tag + "_D"], g[tag + "_invcov"], g[tag + "_loc"], g[tag + "_scale"], return_best=True))
    eng.put("TEMPL", templ)
    logp, full, best = like.logp(np.stack(rows), return_best=True)
    assert np.isclose(logp[0], g[tag + "_logp"], rtol=1e-10) and np.isclose(full[0], g[tag + "_fullchi2"], rtol=1e-9)
    assert relerr(best[0][None], g[tag + "_best"][None]) < 1e-8
    for i in range(B):
        assert np.isclose(logp[i], want[i][0], rtol=1e-10), i
        assert np.isclose(full[i], want[i][1], rtol=1e-9), i
        assert relerr(best[i][None], want[i][2][None]) < 1e-8, i
    # Jeffreys and flat-prior variants (reference marginal.py:118-121, 73-75)
    like_j = MarginalLikelihood(eng, index, g[tag + "_D"], g[tag + "_invcov"], g[tag + "_loc"], g[tag + "_scale"], jeffreys=True)
    assert np.isclose(like_j.logp(np.stack(rows))[0], g[tag + "_logp_jeffreys"], rtol=1e-10)
    if tag == "auto":  # (the 11 cross-tracer parameters are degenerate on 36 data points without a prior: cond F2 = 2e18,
        nG = len(g[tag + "_loc"])  # the sign of det F2 is rounding noise there, in the reference as well)
        like_f = MarginalLikelihood(eng, index, g[tag + "_D"], g[tag + "_invcov"], np.zeros(nG), np.full(nG, np.inf))
        assert np.isclose(like_f.logp(np.stack(rows))[0], g[tag + "_logp_flat"], rtol=1e-9)
    eng.close()


def test_marginal_error_paths(golden):
    from eftpipe_amd import _lib as L
    from eftpipe_amd.marginal import MarginalLikelihood

    g = golden("marg")
    eng, nx = _engine(g, 2)
    with pytest.raises(L.EftbError, match="eftb_set_likelihood"):
        eng.run(L.S_LOGP, 1)  # no likelihood registered yet
    D, C = g["auto_D"], g["auto_invcov"]
    idx = np.arange(D.size, dtype=np.int32)
    with pytest.raises(L.EftbError, match="outside"):
        MarginalLikelihood(eng, idx + 3 * nx, D, C, np.zeros(7), np.ones(7))
    bad = C.copy()
    bad[0, 1] += 1.0
    with pytest.raises(L.EftbError, match="symmetric"):
        MarginalLikelihood(eng, idx, D, bad, np.zeros(7), np.ones(7))
    with pytest.raises(ValueError, match="infinite scale"):
        MarginalLikelihood(eng, idx, D, C, np.zeros(7), np.array([np.inf] + [1.0] * 6))
    # a singular Fisher matrix (all derivative rows zero, flat prior) is reported like the reference does
    like = MarginalLikelihood(eng, idx, D, C, np.zeros(7), np.full(7, np.inf))
    with pytest.raises(RuntimeError, match="det of F2ij"):
        like.logp(np.zeros((1, 8, 24)))
    eng.close()


def test_eastcoast_basis_on_device(golden):
    """EastCoastBasis (reference parambasis.py:320-454).  The templates are basis independent, so the
    east-coast parameters enter the device through the 24 reduce coefficients (REDUCE stage) and the Gaussian rows (LOGP stage)."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.marginal import MarginalLikelihood, data_index
    from eftpipe_amd.parambasis import EastCoastBasis
    from oracle import marginal as M

    g, c = golden("east"), golden("caseC")
    B = 4
    nx = c["binned_P11l"].shape[-1]
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    eng = Engine(EngineConfig(Nl=3), max_batch=B)
    eng.set_template_dims(3, nx)
    T = np.concatenate([c["binned_P11l"], c["binned_Pctl"], c["binned_Ploopl"], c["binned_Pstl"]], axis=1)
    f = float(g["f"])
    kmA, krA, ndA = g["co"]
    basis = EastCoastBasis(prefix="")
    full = dict(zip(g["full_names"], g["full_values"]))
    walkers = [dict(full, b1=full["b1"] + 0.1 * i, bG2=full["bG2"] - 0.05 * i, c2=full["c2"] * (1 + i)) for i in range(B)]
    templ = np.stack([T * (1.0 + 0.1 * i) for i in range(B)])
    eng.put("TEMPL", templ)
    # REDUCE: P_l = east-coast coefficients . templates
    eng.put("BIAS", np.stack([basis.bias_row(f, w, kmA=kmA, krA=krA, ndA=ndA) for w in walkers]))
    eng.run(L.S_REDUCE, B)
    plk = eng.get("PLK", (B, 3, nx))
    assert relerr(plk[0], g["plk"]) < 1e-13
    from oracle import OracleConfig, OracleEngine

    orc = OracleEngine(OracleConfig(Nl=3, kmA=kmA, krA=krA, ndA=ndA))
    for i, w in enumerate(walkers):
        st = {"P11l": templ[i][:, 0:3], "Pctl": templ[i][:, 3:9], "Ploopl": templ[i][:, 9:21], "Pstl": templ[i][:, 21:24], "Picc": np.zeros((3, nx))}
        bsA, es = M.eastcoast_bs(f, **w)
        assert relerr(plk[i], orc.reduce_plk(f, st, bsA, None, es, counterform="eastcoast")) < 1e-13, i
    # LOGP: marginalise bGamma3, c0, c2, c4, Pshot, a0, a2
    ls = list(g["ls"])
    index = data_index(ls, {l: slice(a, b) for l, (a, b) in zip(ls, g["masks"])}, nx)
    like = MarginalLikelihood(eng, index, g["D"], g["invcov"], g["loc"], g["scale"])
    rows = np.stack([basis.gaussian_rows(f, w, kmA=kmA, krA=krA, ndA=ndA) for w in walkers])
    logp, fullchi2, best = like.logp(rows, return_best=True)
    assert np.isclose(logp[0], g["logp"], rtol=1e-10) and np.isclose(fullchi2[0], g["fullchi2"], rtol=1e-9)
    assert relerr(best[0][None], g["best"][None]) < 1e-8
    for i, w in enumerate(walkers):
        st = {"Pctl": templ[i][:, 3:9], "Ploopl": templ[i][:, 9:21], "Pstl": templ[i][:, 21:24]}
        PG = np.stack([M.flatten(ls, t, {l: slice(a, b) for l, (a, b) in zip(ls, g["masks"])}) for t in M.eastcoast_derivative_table(st, f, w["b1"], kmA, ndA)])
        PNG = (np.einsum("b,lbx->lx", rows[i][0], templ[i])).reshape(-1)[index]
        want = M.marginalized_logp(PG, PNG, g["D"], g["invcov"], g["loc"], g["scale"])
        assert np.isclose(logp[i], want, rtol=1e-10), i
    like_j = MarginalLikelihood(eng, index, g["D"], g["invcov"], g["loc"], g["scale"], jeffreys=True)
    assert np.isclose(like_j.logp(rows)[0], g["logp_jeffreys"], rtol=1e-10)
    eng.close()


def test_plain_chi2_without_marginalised_parameters(golden):
    """nG = 0: the LOGP stage is the plain Gaussian likelihood -chi2 / 2 of the full model (reference likelihood.py calculate without
    marginalisation: every parameter sampled)."""
    from eftpipe_amd.marginal import MarginalLikelihood, data_index
    from eftpipe_amd.parambasis import bias_row

    g = golden("marg")
    eng, nx = _engine(g, 3)
    T = np.concatenate([g["binned_P11l"], g["binned_Pctl"], g["binned_Ploopl"], g["binned_Pstl"]], axis=1)
    ls = list(g["ls"])
    index = data_index(ls, {l: slice(a, b) for l, (a, b) in zip(ls, g["masks"])}, nx)
    D, C = g["auto_D"], g["auto_invcov"]
    like = MarginalLikelihood(eng, index, D, C, np.zeros(0), np.zeros(0))
    f = float(g["f"])
    rows = np.stack([bias_row(f, [2.1 + 0.1 * i, 0.5, 0.3, 0.2, -1.0, -2.0, 0.5], None, (0.3, 0.1, -0.4), kmA=0.7, krA=0.25, ndA=4.5e-5) for i in range(3)])
    eng.put("TEMPL", np.stack([T, T, 1.1 * T]))
    logp = like.logp(rows[:, None, :])
    for i in range(3):
        r = np.einsum("r,lrx->lx", rows[i], (1.1 if i == 2 else 1.0) * T).reshape(-1)[index] - D
        assert np.isclose(logp[i], -0.5 * r @ C @ r, rtol=1e-11), i
    eng.close()


def test_pipelined_steps_to_log_posterior(golden):
    """Staged inputs + likelihood rows, theory + LOGP per step, results fetched one step behind (Engine.pipeline): every step must equal
    the synchronous eftb_eval_logp_batch on the same inputs."""
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.marginal import MarginalLikelihood, data_index, gaussian_rows
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    k = g["k"]
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    nb = len(g["kout"])
    B = 3
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=B)
    eng.set_pipeline_operator(eng.add_operator(TB.compose_operator(3, k.size, binning=Bm)))
    index = data_index([0, 2], {0: slice(0, nb), 2: slice(1, nb - 1)}, nb)
    rng = np.random.default_rng(21)
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    mk = lambda: dict(Pin=g["Pin"][None] * (1.0 + 0.1 * rng.uniform(-1, 1, (B, 1))), f=f0 * (1.0 + 0.03 * rng.uniform(-1, 1, B)),
                      DA=DA0 * (1.0 + 0.02 * rng.uniform(-1, 1, B)), H=H0 * (1.0 + 0.02 * rng.uniform(-1, 1, B)))
    steps = [mk() for _ in range(5)]
    for st in steps:
        st["rows"] = np.stack([gaussian_rows(fi, (2.0 + 0.1 * rng.uniform(), 0.5, 0.3), None, 0.7, 0.25, 4.5e-5) for fi in st["f"]])
    templ = eng.eval_batch(steps[0]["Pin"], steps[0]["f"], steps[0]["DA"], steps[0]["H"])
    model = np.einsum("r,lrx->lx", steps[0]["rows"][0, 0], templ[0]).reshape(-1)[index]
    sig = 0.05 * np.abs(model) + 10.0
    like = MarginalLikelihood(eng, index, model * 1.02, np.diag(1.0 / sig**2), np.zeros(7), np.full(7, 3.0))
    want = [like.eval_logp(st["Pin"], st["f"], st["DA"], st["H"], st["rows"]) for st in steps]
    for s, res in enumerate(eng.pipeline(steps, fetch="LOGP")):
        assert res.shape == (B, 26) and np.array_equal(res[:, 0], want[s]), s
    eng.close()


def test_coalesced_steps_to_log_posterior(golden):
    """Engine(coalesce=4) with a likelihood: staged steps with their own Gaussian rows, held in the queue so that they leave as ONE launch (ragged batch
    sizes), then a free-running loop -- every step's ln P must be the bits of the synchronous eftb_eval_logp_batch on the same inputs."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.marginal import MarginalLikelihood, data_index, gaussian_rows
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    k = g["k"]
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    nb = len(g["kout"])
    B = 6
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=B, coalesce=4)
    eng.set_pipeline_operator(eng.add_operator(TB.compose_operator(3, k.size, binning=Bm)))
    index = data_index([0, 2], {0: slice(0, nb), 2: slice(1, nb - 1)}, nb)
    rng = np.random.default_rng(22)
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    sizes = [6, 4, 6, 1, 6, 6, 3, 6]

    def mk(n):
        st = dict(Pin=g["Pin"][None] * (1.0 + 0.1 * rng.uniform(-1, 1, (n, 1))), f=f0 * (1.0 + 0.03 * rng.uniform(-1, 1, n)),
                  DA=DA0 * (1.0 + 0.02 * rng.uniform(-1, 1, n)), H=H0 * (1.0 + 0.02 * rng.uniform(-1, 1, n)))
        st["rows"] = np.stack([gaussian_rows(fi, (2.0 + 0.1 * rng.uniform(), 0.5, 0.3), None, 0.7, 0.25, 4.5e-5) for fi in st["f"]])
        return st

    steps = [mk(n) for n in sizes]
    templ = eng.eval_batch(steps[0]["Pin"], steps[0]["f"], steps[0]["DA"], steps[0]["H"])
    model = np.einsum("r,lrx->lx", steps[0]["rows"][0, 0], templ[0]).reshape(-1)[index]
    sig = 0.05 * np.abs(model) + 10.0
    like = MarginalLikelihood(eng, index, model * 1.02, np.diag(1.0 / sig**2), np.zeros(7), np.full(7, 3.0))
    want = [like.eval_logp(st["Pin"], st["f"], st["DA"], st["H"], st["rows"]) for st in steps]
    mask = eng.full_mask() | L.S_LOGP
    eng.set_latency_mode(False)
    eng.set_submit_thread(2)
    eng.submit_stats(enable=True, reset=True)
    for a, b in [(0, 4), (4, 7), (7, 8)]:   # steps [a, b) leave together
        eng.hold_submissions(True)
        for i in range(a, b):
            st = steps[i]
            eng.step(mask, st["Pin"], st["f"], st["DA"], st["H"], rows=st["rows"])
        eng.hold_submissions(False)
        for i in range(a, b):
            got = eng.fetch_previous("LOGP", (sizes[i], 26), back=b - 1 - i)
            assert np.array_equal(got[:, 0], want[i]), (a, b, i)
    st = eng.submit_stats(enable=False)
    assert st["steps"] == 8 and st["launches"] == 3, st
    eng.set_submit_thread(1)
    depth = 3
    for i in range(16):   # free running: whatever grouping the timing produces
        st = steps[i % 8]
        view = eng.step(mask, st["Pin"], st["f"], st["DA"], st["H"], rows=st["rows"], back=depth if i >= depth else -1, fetch="LOGP", shape=(sizes[(i - depth) % 8], 26))
        if i >= depth:
            assert np.array_equal(view[:, 0], want[(i - depth) % 8]), i
    eng.close()
