"""GPU parity: every stage of the HIP engine (through the C ABI) against the CPU oracle and the
reference-generated golden fixtures.  Tolerance: 1e-6 relative is the north-star bar; we assert
1e-8 (row-scaled) because both sides are FP64 and only summation order differs."""
import os

import numpy as np
import pytest

from conftest import relerr
from eftpipe_amd import synth
from oracle_util import oracle_engine

pytestmark = pytest.mark.gpu
TOL = 1e-8
ROWS = dict(P11l=slice(0, 3), Pctl=slice(3, 9), Ploopl=slice(9, 21), Pstl=slice(21, 24))


def make_engine(g, resum, ap, APst=False, max_batch=1):
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    native = g["k"].size == 50
    z = float(g["z"])
    cfg = EngineConfig(Nl=int(g["Nl"]), k=None if native else g["k"], with_resum=resum, with_ap=ap, APst=APst,
                       DA_AP=float(synth.da_func(synth.OM_AP, z)), H_AP=float(synth.hubble(synth.OM_AP, z)))
    return Engine(cfg, max_batch=max_batch)


@pytest.mark.parametrize("name,resum,ap,APst", [("caseA", False, False, False), ("caseB", False, False, False),
                                                 ("caseE", True, True, False), ("caseC", True, True, True),
                                                 ("caseD", True, True, False)])
def test_stage_parity(golden, name, resum, ap, APst):
    from eftpipe_amd import _lib as L

    g = golden(name)
    Nl, Nk = int(g["Nl"]), g["k"].size
    f, DA, H = float(g["f"]), float(g["DA"]), float(g["H"])
    orc = oracle_engine(g, name, window_file=None, kout=None)
    taps = {}
    orc.evaluate(g["kin"], g["Pin"], f, DA, H, pairwise=True, taps=taps)
    eng = make_engine(g, resum, ap, APst)
    eng.load_inputs(g["Pin"], f, DA if ap else None, H if ap else None)

    eng.run(L.S_PREP | L.S_LOOPS)
    assert relerr(eng.get("P11", (Nk,)), g["pscf_P11"]) < 1e-12
    c = eng.get("COEF", (2, 129))
    assert relerr(c[0] + 1j * c[1], g["coef"][:129]) < 1e-12
    assert relerr(eng.get("P22", (28, Nk)), g["pscf_P22"]) < TOL
    assert relerr(eng.get("P13", (10, Nk)), g["pscf_P13"]) < TOL
    if resum:
        eng.run(L.S_CF)
        assert relerr(eng.get("C11", (Nl, 80)), g["pscf_C11"]) < TOL
        assert relerr(eng.get("CCT", (Nl, 80)), g["pscf_Cct"]) < TOL
        cc = eng.get("CC", (Nl * 38, 80))
        assert relerr(cc[: Nl * 28].reshape(Nl, 28, 80), g["pscf_C22"]) < TOL
        assert relerr(cc[Nl * 28 :].reshape(Nl, 10, 80), g["pscf_C13"]) < TOL
    eng.run(L.S_REGROUP)
    T = eng.get("TEMPL", (Nl, 24, Nk))
    for n, sl in ROWS.items():
        assert relerr(T[:, sl], g["setpscfl_" + n]) < TOL, n
    if resum:
        assert relerr(eng.get("CLOOPL", (Nl, 12, 80)), g["setpscfl_Cloopl"]) < TOL
        eng.run(L.S_RESUM)
        xy = eng.get("XY", (2, 80))
        assert relerr(xy[0], g["resum_X"]) < 1e-10 and relerr(xy[1], g["resum_Y"]) < 1e-10
        assert relerr(eng.get("Q", g["resum_Q"].shape).reshape(-1, g["resum_Q"].shape[-1]),
                      g["resum_Q"].reshape(-1, g["resum_Q"].shape[-1])) < 1e-12
        T = eng.get("TEMPL", (Nl, 24, Nk))
        for n in ("P11l", "Pctl", "Ploopl"):
            assert relerr(T[:, ROWS[n]], g["resum_" + n]) < TOL, n
    if ap:
        eng.run(L.S_AP)
        T = eng.get("TEMPL", (Nl, 24, Nk))
        for n, sl in ROWS.items():
            assert relerr(T[:, sl], g["ap_" + n]) < TOL, n
            assert relerr(T[:, sl], taps["ap"][n]) < TOL, n
    # bias contraction on the device (auto spectrum, no Picc)
    bsA, es = list(g["bsA"]), tuple(g["es"])
    b11, bloop, bct, bst = orc.bias_vectors(f, bsA, None, es)
    eng.put("BIAS", np.concatenate([b11, bct, bloop, bst]))
    eng.run(L.S_REDUCE)
    last = taps["ap"] if ap else taps["setpscfl"]
    assert relerr(eng.get("PLK", (Nl, Nk)), orc.reduce_plk(f, last, bsA, es=es)) < TOL
    if name != "caseC":  # caseC's golden P_l is taken after the window stage
        assert relerr(eng.get("PLK", (Nl, Nk)), g["plk_auto"]) < TOL
    eng.close()


def test_eval_batch_matches_stagewise_and_oracle(golden):
    """Batched one-call path (host buffers in/out) over seeded draws == oracle per draw."""
    g = golden("caseE")
    B = 5
    draws = synth.draw_batch(B, z=0.7)
    eng = make_engine(g, True, True, max_batch=8)
    templ = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"])
    orc = oracle_engine(g, "caseE")
    for i in range(B):
        st = orc.evaluate(draws["kin"], draws["Pin"][i], float(draws["f"][i]), float(draws["DA"][i]), float(draws["H"][i]))
        for n, sl in ROWS.items():
            assert relerr(templ[i][:, sl], st[n]) < TOL, (i, n)
    eng.close()


def test_final_nk2048(golden):
    g = golden("caseF")
    eng = make_engine(g, True, True)
    templ = eng.eval_batch(g["Pin"], float(g["f"]), float(g["DA"]), float(g["H"]))[0]
    for n, sl in ROWS.items():
        assert relerr(templ[:, sl], g["ap_" + n]) < TOL, n
    eng.close()


@pytest.mark.parametrize("name,APst,mode", [("caseE", False, None), ("caseD", False, None), ("caseD", True, None), ("caseF", False, None),
                                            ("caseD", False, "1"), ("caseD", True, "1"), ("caseE", False, "1"), ("caseD", False, "2"),
                                            ("caseF", False, "0")])
def test_ap_extreme_distortions(golden, monkeypatch, name, APst, mode):
    """AP stage alone over a batch of strong distortions: many knot intervals crossed (rising and falling k'(mu)),
    extrapolation past both ends of the k grid, the identity, and the isotropic F = 1 case.  mode: the engine's choice (knot weights +
    banded product with the quadrature fallback for tiles it cannot hold; interval moments at Nk = 2048) or one form forced through
    EFTB_AP_MODE (0 weights / fallback, 1 interval moments, 2 the reference's quadrature everywhere)."""
    from eftpipe_amd import _lib as L

    if mode is not None:
        monkeypatch.setenv("EFTB_AP_MODE", mode)
    g = golden(name)
    Nl, Nk = int(g["Nl"]), g["k"].size
    orc = oracle_engine(g, name, window_file=None, kout=None)
    orc.cfg.APst = APst
    qs = [(1.0, 1.0), (0.85, 1.15), (1.15, 0.85), (1.12, 1.12), (0.9, 0.9), (1.0, 1.1), (1.03, 0.999), (0.999, 1.0)]
    B = len(qs)
    eng = make_engine(g, True, True, APst, max_batch=B)
    src = "resum_" if "resum_P11l" in g else "ap_"  # any smooth template set will do
    base = np.concatenate([g[src + "P11l"], g[src + "Pctl"], g[src + "Ploopl"],
                           g["setpscfl_Pstl"] if "setpscfl_Pstl" in g else g["ap_Pstl"]], axis=1)
    templ = np.stack([base * (1.0 + 0.25 * i) for i in range(B)])
    DA = np.array([q[0] * orc.DA_fid for q in qs])
    H = np.array([orc.H_fid / q[1] for q in qs])
    eng.put("TEMPL", templ)
    eng.put("DA", DA)
    eng.put("H", H)
    eng.set_template_dims(Nl, Nk)
    eng.run(L.S_AP, B)
    out = eng.get("TEMPL", (B, Nl, 24, Nk))
    for i in range(B):
        st = {n: templ[i][:, sl] for n, sl in ROWS.items()}
        ref = orc.ap(float(DA[i]), float(H[i]), st)
        # strict wherever k'(mu) stays within 10 knot spacings of the grid; further out both sides extrapolate the end
        # cubic (up to ~90 spacings for 15 % distortions) and the cubic itself amplifies rounding (t/h)^3-fold, so
        # only a sanity bound applies there
        qperp, qpar = qs[i]
        kk = g["k"]
        near = kk / qperp * max(1.0, qperp / qpar) <= kk[-1] + 10 * (kk[-1] - kk[-2])
        for n, sl in ROWS.items():
            scale = np.max(np.abs(ref[n]), axis=-1, keepdims=True) + 1e-300
            err = np.abs(out[i][:, sl] - ref[n]) / scale
            assert np.max(err[..., near]) < TOL, (i, n, qs[i])
            assert np.max(err) < 1e-5, (i, n, qs[i])
    eng.close()


def test_batch_position_and_size_independence(golden):
    """A cosmology's result must not depend on the batch it sits in: 70 seeded draws in one call (two lane groups of the
    anti-diagonal kernel, a ragged last one) against the same draws evaluated alone and in small groups; plus the
    size-independent properties of the path: P_l is linear in the bias vector and the templates do not depend on it."""
    g = golden("caseE")
    B = 70
    draws = synth.draw_batch(B, z=0.7)
    eng = make_engine(g, True, True, max_batch=B)
    full = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"])
    for sl in (slice(0, 1), slice(63, 65), slice(69, 70), slice(17, 40)):
        part = eng.eval_batch(draws["Pin"][sl], draws["f"][sl], draws["DA"][sl], draws["H"][sl])
        assert np.array_equal(part, full[sl]), sl  # same kernels, same order of operations: bit-identical
    # linearity of the bias contraction
    rng = np.random.default_rng(3)
    b1, b2 = rng.normal(size=(B, 24)), rng.normal(size=(B, 24))
    _, p1 = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=b1)
    _, p2 = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=b2)
    t3, p3 = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=2.0 * b1 - 0.5 * b2)
    assert np.array_equal(t3, full)
    scale = np.max(np.abs(p1) + np.abs(p2), axis=-1, keepdims=True)
    assert np.max(np.abs(p3 - (2.0 * p1 - 0.5 * p2)) / scale) < 1e-13
    eng.close()


def test_c_abi_error_paths(golden):
    """Errors come back as status 1 + eftb_last_error (raised as EftbError by the ctypes layer), never as a crash."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseA")
    eng = make_engine(g, False, False, max_batch=2)
    with pytest.raises(L.EftbError, match="batch"):
        eng.run(L.S_PREP, 3)  # larger than max_batch
    with pytest.raises(L.EftbError, match="with_resum"):
        eng.run(L.S_CF, 1)  # stage of a feature the engine was built without
    with pytest.raises(L.EftbError, match="with_ap"):
        eng.run(L.S_AP, 1)
    with pytest.raises(L.EftbError, match="PROJECT"):
        eng.run(L.S_PROJECT, 1)  # no pipeline operator registered
    with pytest.raises(L.EftbError):
        eng.put("PIN", np.zeros(10 * 200 + 1))  # beyond the buffer
    with pytest.raises(L.EftbError, match="not used|expects"):
        eng._set("H", np.zeros(4))  # table of a disabled feature / wrong size
    # input guards (include/eftbird.h "Input guards"; reference fftlog.py:146-151 takes the logarithm of the last two samples):
    # host inputs are refused before anything is copied ...
    bad_tail, bad_nan = g["Pin"].copy(), g["Pin"].copy()
    bad_tail[-1] = -1.0
    bad_nan[57] = np.nan
    with pytest.raises(L.EftbError, match="positive at its last two samples"):
        eng.eval_batch(np.stack([g["Pin"], bad_tail]), float(g["f"]))
    with pytest.raises(L.EftbError, match=r"Pin\[0\]\[57\] is not finite"):
        eng.eval_batch(bad_nan, float(g["f"]))
    with pytest.raises(L.EftbError, match="f\\[1\\] is not finite"):
        eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), np.array([0.7, np.inf]))
    with pytest.raises(L.EftbError, match="positive at its last two samples"):
        eng.stage_inputs(np.stack([g["Pin"], bad_tail]), np.array([0.7, 0.7]))
    # ... inputs placed with eftb_put raise a flag in the first kernel, reported by the next synchronising call, once
    eng.put("PIN", np.stack([g["Pin"], bad_tail]))
    eng.put("F", np.array([0.7, 0.7]))
    with pytest.raises(L.EftbError, match="cosmology 1"):
        eng.run(L.S_PREP | L.S_LOOPS, 2)
    eng.sync()
    # optional output check: non-finite P_l is flagged by the REDUCE stage (huge bias coefficients overflow the contraction)
    eng.load_inputs(np.stack([g["Pin"], g["Pin"]]), float(g["f"]), bias=np.stack([np.ones(24), np.full(24, 1e308)]))
    eng.run(eng.full_mask(reduce=True), 2)                      # check off: silent, as the reference
    assert not np.all(np.isfinite(eng.get("PLK", (2, 2, g["k"].size))[1]))
    eng.set_check_finite(True)
    with pytest.raises(L.EftbError, match="non-finite P_l.*cosmology 1"):
        eng.run(eng.full_mask(reduce=True), 2)
    eng.set_check_finite(False)
    # the engine is still usable afterwards
    eng.load_inputs(g["Pin"], float(g["f"]))
    eng.run(L.S_PREP | L.S_LOOPS)
    assert relerr(eng.get("P22", (28, g["k"].size)), g["pscf_P22"]) < TOL
    eng.close()


@pytest.mark.parametrize("opts", [dict(with_NNLO=True), dict(optiresum=True), dict(IRcutoff="resum", kIR=0.004),
                                  dict(with_NNLO=True, optiresum=True, IRcutoff="loop", kIR=0.006)])
def test_nl2_option_branches_against_oracle(golden, opts):
    """The option branches (NNLO, optiresum, IRcutoff) on the Nl = 2 engine, whose resummation runs on the generic
    kernel instead of the matrix-core one -- against the oracle (pinned to the reference for these options at Nl = 3)."""
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig
    from oracle import OracleConfig, OracleEngine

    g = golden("caseE")
    f, DA, H = float(g["f"]), float(g["DA"]), float(g["H"])
    orc = OracleEngine(OracleConfig(Nl=2, kmA=0.7, krA=0.25, ndA=4.5e-5, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]),
                                    H_AP=float(g["H_AP"]), **opts))
    want = orc.evaluate(g["kin"], g["Pin"], f, DA, H)
    eng = Engine(EngineConfig(Nl=2, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]), **opts), max_batch=2)
    templ = eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), f, DA, H)
    for i in range(2):
        assert relerr(templ[i][:, 0:3], want["P11l"]) < TOL and relerr(templ[i][:, 3:9], want["Pctl"]) < TOL
        assert relerr(templ[i][:, 9:21], want["Ploopl"]) < TOL and relerr(templ[i][:, 21:24], want["Pstl"]) < TOL
    if opts.get("with_NNLO"):
        tn = eng.get("TEMPLN", (2, 2, 24, g["k"].size))
        assert relerr(tn[1][:, 3:6], want["PctNNLOl"]) < TOL
    eng.close()


def test_no_rsd_growth_rate_zero(golden):
    """with_RSD: false sets f = 0 (reference theory.py:565-566): every f-power of the regrouping and of Q(f) at its edge."""
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig
    from oracle import OracleConfig, OracleEngine

    g = golden("caseC")
    kw = dict(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]))
    want = OracleEngine(OracleConfig(kmA=0.7, krA=0.25, ndA=4.5e-5, **kw)).evaluate(g["kin"], g["Pin"], 0.0, float(g["DA"]), float(g["H"]))
    eng = Engine(EngineConfig(**kw), max_batch=2)
    templ = eng.eval_batch(np.stack([g["Pin"], g["Pin"]]), np.array([0.0, float(g["f"])]), float(g["DA"]), float(g["H"]))
    for n, sl in (("P11l", slice(0, 3)), ("Pctl", slice(3, 9)), ("Ploopl", slice(9, 21)), ("Pstl", slice(21, 24))):
        assert relerr(templ[0][:, sl], want[n]) < TOL, n
    assert relerr(templ[1][:, 9:21], g["ap_Ploopl"]) < TOL  # the neighbour with f > 0 is untouched
    eng.close()


def test_graph_replay_is_bit_identical(golden):
    """EFTB_O_GRAPH: captured-graph replay of the whole pipeline (AP swaps the two template blocks, so two graphs alternate) gives
    exactly the plain-launch results, also after the launch state changes (another batch size, a pipeline operator)."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=3)
    Pin = np.stack([g["Pin"], 1.1 * g["Pin"], 0.9 * g["Pin"]])
    args = (float(g["f"]), float(g["DA"]), float(g["H"]))
    plain = [eng.eval_batch(Pin, *args), eng.eval_batch(Pin[:2], *args)]
    eng.set_graph_replay(True)
    for _ in range(4):  # capture (two pointer parities), then replays
        assert np.array_equal(eng.eval_batch(Pin, *args), plain[0])
    assert np.array_equal(eng.eval_batch(Pin[:2], *args), plain[1])
    op = eng.add_operator(np.einsum("al,xk->alxk", np.eye(3), np.eye(50)[::2]))
    eng.set_pipeline_operator(op)
    a = eng.eval_batch(Pin, *args)
    b = eng.eval_batch(Pin, *args)
    assert np.array_equal(a, b) and np.array_equal(a, plain[0][..., ::2])
    eng.load_inputs(Pin, *args)
    for _ in range(3):
        eng.run(eng.full_mask(), 3)
    assert np.array_equal(eng.get("TEMPL", (3, 3, 24, 25)), a)
    eng.close()


def test_bench_workload_against_oracle():
    """The cfg-2 workload exactly as bench.py runs it (Nl = 3, survey k grid of 512 points, IR-resum + AP, seeded synthetic draws, west-coast
    bias contraction, asynchronous overlapped runs): templates and P_l of the first draws against the oracle."""
    import bench
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig
    from oracle import OracleConfig, OracleEngine

    B, ncheck = 16, 3
    k = synth.survey_kgrid(bench.NK)
    cfg = EngineConfig(Nl=bench.NL, k=k, with_resum=True, with_ap=True, DA_AP=float(synth.da_func(synth.OM_AP, bench.Z)),
                       H_AP=float(synth.hubble(synth.OM_AP, bench.Z)))
    eng = Engine(cfg, max_batch=B)
    draws = synth.draw_batch(B, z=bench.Z, seed=12345)
    bias = np.stack([bias_row(float(f), bench.BS, None, bench.ES, kmA=0.7, krA=0.25, ndA=4.5e-5) for f in draws["f"]])
    eng.load_inputs(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias)
    mask = eng.full_mask(reduce=True)
    for _ in range(3):
        eng.run(mask, B, sync=False)
    templ, plk = eng.get("TEMPL", (B, bench.NL, 24, bench.NK)), eng.get("PLK", (B, bench.NL, bench.NK))
    orc = OracleEngine(OracleConfig(Nl=bench.NL, k=k, ndA=4.5e-5, with_resum=True, with_ap=True, Om_AP=synth.OM_AP, z_AP=bench.Z))
    for i in range(ncheck):
        f = float(draws["f"][i])
        st = orc.evaluate(draws["kin"], draws["Pin"][i], f, float(draws["DA"][i]), float(draws["H"][i]), pairwise=True)
        for n, sl in ROWS.items():
            assert relerr(templ[i][:, sl], st[n]) < TOL, (i, n)
        assert relerr(plk[i], orc.reduce_plk(f, st, list(bench.BS), es=tuple(bench.ES))) < TOL, i
    eng.close()


def test_full_size_properties_of_the_bench_workload():
    """BASELINE cfg 2 at the bench's full size (128 cosmologies per step, Nk = 512, Nl = 3), where the oracle takes minutes: properties that
    need no reference values.  (1) Without resummation and AP the templates are homogeneous in P_lin: P11l, Pctl scale with alpha, Ploopl
    with alpha^2 (reference pybird.py:1074-1086, 737-866).  (2) AP at the fiducial (DA, H) is the 3 x 3 multipole
    mixing matrix of the reference's own mu quadrature, applied row by row.  (3) A
    permutation of the batch permutes the output, bit for bit.  (4) P_l is linear in the 24 bias coefficients."""
    import bench
    from eftpipe_amd import _lib as L
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    B, NK, NL = 128, bench.NK, bench.NL
    k = synth.survey_kgrid(NK)
    DAf, Hf = float(synth.da_func(synth.OM_AP, bench.Z)), float(synth.hubble(synth.OM_AP, bench.Z))
    draws = synth.draw_batch(B, z=bench.Z, seed=2024)
    rng = np.random.default_rng(5)
    # (1) homogeneity, no resummation / AP
    eng = Engine(EngineConfig(Nl=NL, k=k), max_batch=B)
    alpha = 1.7
    t1 = eng.eval_batch(draws["Pin"], draws["f"])
    t2 = eng.eval_batch(alpha * draws["Pin"], draws["f"])
    for sl, power in ((ROWS["P11l"], 1), (ROWS["Pctl"], 1), (ROWS["Ploopl"], 2)):
        scale = np.max(np.abs(t1[:, :, sl]), axis=-1, keepdims=True) + 1e-300
        assert np.max(np.abs(t2[:, :, sl] - alpha**power * t1[:, :, sl]) / scale) < 1e-11, power
    assert np.array_equal(t2[:, :, ROWS["Pstl"]], t1[:, :, ROWS["Pstl"]])
    eng.close()
    # (2) - (4) on the full path
    eng = Engine(EngineConfig(Nl=NL, k=k, with_resum=True, with_ap=True, DA_AP=DAf, H_AP=Hf), max_batch=B)
    eng.load_inputs(draws["Pin"], draws["f"], np.full(B, DAf), np.full(B, Hf))
    eng.run(eng.full_mask() & ~L.S_AP, B)
    before = eng.get("TEMPL", (B, NL, 24, NK))
    eng.run(L.S_AP, B)
    after = eng.get("TEMPL", (B, NL, 24, NK))
    # at q_perp = q_par = 1 every node of the mu quadrature sits at k' = k (a knot of the spline), so the stage reduces to the 3 x 3 matrix of
    # the reference's trapezoid rule on 200 nodes, M[l][l'] = 2 sum_j w_j (2l+1)/2 L_l(mu_j) L_l'(mu_j) (pybird.py:1581-1596): not the identity
    from eftpipe_amd.tables import build_tables
    from scipy.special import eval_legendre

    tb_ = build_tables(EngineConfig(Nl=NL, k=k, with_resum=True, with_ap=True, DA_AP=DAf, H_AP=Hf))
    M = np.array([[2.0 * np.sum(tb_["wmu"] * tb_["legmu"][l] * eval_legendre(2 * lp, tb_["mu"])) for lp in range(NL)] for l in range(NL)])
    assert np.max(np.abs(M - np.eye(NL))) < 1e-3
    want = np.einsum("lm,bmrk->blrk", M, before[:, :, :21])
    scale = np.max(np.abs(want), axis=-1, keepdims=True) + 1e-300
    assert np.max(np.abs(after[:, :, :21] - want) / scale) < 1e-10
    assert np.array_equal(after[:, :, 21:], before[:, :, 21:])  # Pstl is copied through (APst off)
    bias = rng.normal(size=(B, 24))
    tf, pf = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=bias)
    perm = rng.permutation(B)
    tp, pp = eng.eval_batch(draws["Pin"][perm], draws["f"][perm], draws["DA"][perm], draws["H"][perm], bias=bias[perm])
    assert np.array_equal(tp, tf[perm]) and np.array_equal(pp, pf[perm])
    b2 = rng.normal(size=(B, 24))
    _, p2 = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=b2)
    _, p3 = eng.eval_batch(draws["Pin"], draws["f"], draws["DA"], draws["H"], bias=0.5 * bias - 3.0 * b2)
    ps = np.max(np.abs(pf) + np.abs(p2), axis=-1, keepdims=True)
    assert np.max(np.abs(p3 - (0.5 * pf - 3.0 * p2)) / ps) < 1e-13
    assert np.all(np.isfinite(tf)) and np.all(np.isfinite(pf))
    eng.close()


def test_overlapped_steps_are_bit_identical(golden, monkeypatch):
    """Asynchronous runs overlap on three streams (front half of run i+1 and back half -- spline, AP, reduce -- of run i beside the
    resummation, three template blocks rotating): any number of queued runs, and input changes in between, must give exactly what an
    engine with every overlap switched off gives."""
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    B = 5
    cfg = EngineConfig(Nl=3, with_resum=True, with_ap=True, APst=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]))
    monkeypatch.setenv("EFTB_AP_OVERLAP", "0")
    monkeypatch.setenv("EFTB_PREP_OVERLAP", "0")
    serial = Engine(cfg, max_batch=B)
    monkeypatch.delenv("EFTB_AP_OVERLAP")
    monkeypatch.delenv("EFTB_PREP_OVERLAP")
    eng = Engine(cfg, max_batch=B)
    rng = np.random.default_rng(11)
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    mask = eng.full_mask(reduce=True)
    shape_t, shape_p = (B, 3, 24, g["k"].size), (B, 3, g["k"].size)
    for nrun in (1, 2, 3, 4, 7):
        scale = 1.0 + 0.1 * rng.standard_normal(B)
        f = f0 * (1.0 + 0.05 * rng.standard_normal(B))
        DA, H = DA0 * (1.0 + 0.03 * rng.standard_normal(B)), H0 * (1.0 + 0.03 * rng.standard_normal(B))
        Pin = scale[:, None] * g["Pin"][None, :]
        bias = np.stack([bias_row(float(fi), list(g["bsA"]), None, tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5) for fi in f])
        serial.load_inputs(Pin, f, DA, H, bias)
        serial.run(mask, B)
        want_t, want_p = serial.get("TEMPL", shape_t), serial.get("PLK", shape_p)
        eng.load_inputs(Pin, f, DA, H, bias)
        for _ in range(nrun):
            eng.run(mask, B, sync=False)
        assert np.array_equal(eng.get("PLK", shape_p), want_p), nrun
        assert np.array_equal(eng.get("TEMPL", shape_t), want_t), nrun
    serial.close()
    eng.close()


def test_dominant_kernel_timer(golden):
    """EFTB_O_TIME_DOMINANT: HIP events around every launch of the resummation kernel, on its own stream (what bench.py's roofline divides by)."""
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=16)
    eng.load_inputs(np.stack([g["Pin"]] * 16), float(g["f"]), float(g["DA"]), float(g["H"]))
    mask = eng.full_mask()
    eng.run(mask, 16)
    assert eng.dominant_time() == (0.0, 0)  # off by default
    eng.time_dominant(True)
    for _ in range(11):
        eng.run(mask, 16, sync=False)
    ms, n = eng.dominant_time(reset=False)
    if os.environ.get("EFTB_GRAPH", "0") not in ("", "0"):  # graph replay: the timer stays off (its event records would be captured)
        assert n == 0
        eng.close()
        return
    assert n == 11 and 0.0 < ms / n < 5.0
    assert eng.dominant_time()[1] == 11 and eng.dominant_time() == (0.0, 0)
    eng.time_dominant(False)
    eng.run(mask, 16)
    assert eng.dominant_time()[1] == 0
    eng.close()


def test_overlapped_nnlo_steps_are_bit_identical(golden, monkeypatch):
    """with_NNLO steps on the three-stream layout (the NNLO block rides through regrouping, the fused accumulator of the resummation kernel,
    its own AP pass over the three counter-term rows and REDUCE inside one call, with its own three rotating blocks): queued asynchronous
    runs must give exactly what an engine with every overlap switched off gives, and what the two-pass path gives."""
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    B = 16  # >= 16: unsplit s sums, i.e. the fused NNLO accumulator
    cfg = EngineConfig(Nl=3, with_resum=True, with_ap=True, with_NNLO=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"]))
    monkeypatch.setenv("EFTB_AP_OVERLAP", "0")
    monkeypatch.setenv("EFTB_PREP_OVERLAP", "0")
    serial = Engine(cfg, max_batch=B)
    monkeypatch.delenv("EFTB_AP_OVERLAP")
    monkeypatch.delenv("EFTB_PREP_OVERLAP")
    monkeypatch.setenv("EFTB_NNLO_INLINE", "0")
    twopass = Engine(cfg, max_batch=B)
    monkeypatch.delenv("EFTB_NNLO_INLINE")
    eng = Engine(cfg, max_batch=B)
    rng = np.random.default_rng(12)
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    mask = eng.full_mask(reduce=True)
    nk = g["k"].size
    shape_t, shape_p = (B, 3, 24, nk), (B, 3, nk)
    for nrun in (1, 2, 3, 5):
        scale = 1.0 + 0.1 * rng.standard_normal(B)
        f = f0 * (1.0 + 0.05 * rng.standard_normal(B))
        DA, H = DA0 * (1.0 + 0.03 * rng.standard_normal(B)), H0 * (1.0 + 0.03 * rng.standard_normal(B))
        Pin = scale[:, None] * g["Pin"][None, :]
        bias = np.stack([bias_row(float(fi), list(g["bsA"]), None, tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5) for fi in f])
        biasn = 0.1 * rng.standard_normal((B, 3))
        want = None
        for e in (serial, twopass, eng):
            e.load_inputs(Pin, f, DA, H, bias)
            e.put("BIASN", biasn)
            for _ in range(1 if e is serial else nrun):
                e.run(mask, B, sync=(e is serial))
            got = (e.get("PLK", shape_p), e.get("TEMPL", shape_t), e.get("TEMPLN", shape_t))
            if want is None:
                want = got
                assert np.any(got[2][:, :, 3:6] != 0.0) and not np.any(got[2][:, :, :3]) and not np.any(got[2][:, :, 6:])
            else:
                for a_, w_ in zip(got, want):
                    assert np.array_equal(a_, w_), (nrun, e is eng)
    for e in (serial, twopass, eng):
        e.close()


def test_pipelined_steps_with_changing_inputs(golden):
    """eftb_stage_inputs / eftb_run_staged / eftb_fetch_previous: a sampler loop whose inputs change every step, with the next step's
    inputs staged and the previous step's results fetched while a step is in flight -- every step must equal the synchronous call."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    B, nsteps = 4, 6
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=B)
    rng = np.random.default_rng(8)
    f0, DA0, H0 = float(g["f"]), float(g["DA"]), float(g["H"])
    steps = []
    for s in range(nsteps):
        f = f0 * (1.0 + 0.05 * rng.uniform(-1, 1, B))
        steps.append(dict(Pin=g["Pin"][None] * (1.0 + 0.2 * rng.uniform(-1, 1, (B, 1))), f=f, DA=DA0 * (1.0 + 0.03 * rng.uniform(-1, 1, B)),
                          H=H0 * (1.0 + 0.03 * rng.uniform(-1, 1, B)),
                          bias=np.stack([bias_row(fi, [2.0 + 0.1 * s, 0.5, 0.3, 0.2, -1.0, -2.0, 0.5], None, (0.3, 0.1, -0.4), kmA=0.7, krA=0.25, ndA=4.5e-5) for fi in f])))
    want = [eng.eval_batch(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"], templates=False) for st in steps]
    mask = eng.full_mask(reduce=True)
    got = []
    st = steps[0]
    eng.stage_inputs(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"])
    eng.run_staged(mask, B)
    for s in range(1, nsteps):
        st = steps[s]
        eng.stage_inputs(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"])   # while step s - 1 runs
        eng.run_staged(mask, B)                                                     # queued behind it, front half overlapping
        got.append(eng.fetch_previous("PLK", (B, 3, g["k"].size)))                  # step s - 1
    eng.sync()
    got.append(eng.get("PLK", (B, 3, g["k"].size)))
    for s in range(nsteps):
        assert np.array_equal(got[s], want[s]), s
    with pytest.raises(Exception):
        eng.run_staged(mask, B)  # nothing staged
    # the same loop through the generator (which keeps three steps queued: eftb_fetch_back(3))
    for s, res in enumerate(eng.pipeline(steps)):
        assert np.array_equal(res, want[s]), s
    for n in (1, 2, 3, 4):
        assert len(list(eng.pipeline(steps[:n]))) == min(n, nsteps)
    # explicit loops of depth 2 and 3: the result of step s - depth is copied out after step s has been launched
    for depth in (2, 3):
        got2 = []
        for s in range(nsteps):
            st = steps[s]
            eng.stage_inputs(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"])
            eng.run_staged(mask, B)
            if s >= depth:
                got2.append(eng.fetch_previous("PLK", (B, 3, g["k"].size), back=depth))
        for back in range(min(depth, nsteps) - 1, 0, -1):
            got2.append(eng.fetch_previous("PLK", (B, 3, g["k"].size), back=back))
        eng.sync()
        got2.append(eng.get("PLK", (B, 3, g["k"].size)))
        assert len(got2) == nsteps
        for s in range(nsteps):
            assert np.array_equal(got2[s], want[s]), (depth, s)
    assert np.array_equal(eng.fetch_previous("PLK", (B, 3, g["k"].size), back=0), want[nsteps - 1])  # back = 0: the step launched last
    view = eng.fetch_previous("PLK", (B, 3, g["k"].size), back=1, copy=False)                      # the engine's own page-locked copy, no memcpy
    assert not view.flags.writeable and np.array_equal(view, want[nsteps - 2])
    # the dependent-sampler loop: every step is staged with the GPU idle and runs in latency mode (one queue, P_lin read from the staging block,
    # P_l written to mapped host memory by the kernel that forms it) -- same bits as the pipelined steps and the synchronous call
    for s in range(nsteps):
        st = steps[s]
        eng.stage_inputs(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"])
        eng.run_staged(mask, B)
        assert np.array_equal(eng.fetch_previous("PLK", (B, 3, g["k"].size), back=0, copy=False), want[s]), s
        assert np.array_equal(eng.get("PLK", (B, 3, g["k"].size)), want[s]) and np.array_equal(eng.get("PIN", (B, 200)), st["Pin"]), s   # the device copies too
    with pytest.raises(L.EftbError):
        eng.fetch_previous("PLK", (B, 3, g["k"].size), back=16)
    eng.close()


@pytest.mark.parametrize("direct", [False, True])
def test_pipelined_flags_belong_to_their_step(golden, direct):
    """the guard flags are per rotating set (direct: the same on direct-P_l runs, where the AP quadrature kernel raises the flag).  In a depth-3 pipeline one step is poisoned (huge bias coefficients overflow its
    P_l; EFTB_O_CHECK_FINITE) -- exactly that step's fetch fails, naming the cosmology; the steps queued around it fetch clean results.  And
    eftb_fetch_back refuses a step that was never launched instead of handing out a zero-filled block."""
    from eftpipe_amd import _lib as L
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    g = golden("caseC")
    B, nsteps, depth, poisoned = 2, 6, 3, 2
    eng = Engine(EngineConfig(Nl=3, with_resum=True, with_ap=True, DA_AP=float(g["DA_AP"]), H_AP=float(g["H_AP"])), max_batch=B)
    eng.set_check_finite(True)
    eng.set_plk_direct(direct)
    shape = (B, 3, g["k"].size)
    b0 = bias_row(float(g["f"]), [2.0, 0.5, 0.3, 0.2, -1.0, -2.0, 0.5], None, (0.3, 0.1, -0.4), kmA=0.7, krA=0.25, ndA=4.5e-5)
    steps = [dict(Pin=np.stack([g["Pin"], (1.0 + 0.01 * s) * g["Pin"]]), f=float(g["f"]), DA=float(g["DA"]), H=float(g["H"]), bias=np.stack([b0, b0]))
             for s in range(nsteps)]
    steps[poisoned]["bias"] = np.stack([b0, np.full(24, 1e308)])
    want = [None if s == poisoned else eng.eval_batch(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"], templates=False) for s, st in enumerate(steps)]
    mask = eng.full_mask(reduce=True)
    # nothing launched yet through the staged API
    eng.stage_inputs(steps[0]["Pin"], steps[0]["f"], steps[0]["DA"], steps[0]["H"], bias=steps[0]["bias"])
    with pytest.raises(L.EftbError, match="no staged run yet|staged step"):
        eng.fetch_previous("PLK", shape, back=0)
    eng.run_staged(mask, B)
    with pytest.raises(L.EftbError, match="only 1 staged step"):
        eng.fetch_previous("PLK", shape, back=1)
    assert np.array_equal(eng.fetch_previous("PLK", shape, back=0), want[0])
    outcome = {}

    def fetch(step, back):
        try:
            outcome[step] = eng.fetch_previous("PLK", shape, back=back)
        except L.EftbError as exc:
            outcome[step] = str(exc)

    for s in range(1, nsteps):
        st = steps[s]
        eng.stage_inputs(st["Pin"], st["f"], st["DA"], st["H"], bias=st["bias"])
        eng.run_staged(mask, B)
        if s - depth >= 1:
            fetch(s - depth, depth)
    for back in range(depth - 1, -1, -1):
        fetch(nsteps - 1 - back, back)
    for s in range(1, nsteps):
        if s == poisoned:
            assert isinstance(outcome[s], str) and "non-finite P_l" in outcome[s] and "cosmology 1" in outcome[s], outcome[s]
        else:
            assert isinstance(outcome[s], np.ndarray) and np.array_equal(outcome[s], want[s]), (s, outcome[s])
    eng.sync()   # nothing left to report
    eng.close()
