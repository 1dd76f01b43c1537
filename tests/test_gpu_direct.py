"""GPU parity of direct-P_l runs (EFTB_O_PLK_DIRECT): the bias contraction of reduce_Plk (reference parambasis.py:42-136) taken BEFORE Resum.Ps
and APeffect.AP (pybird.py:1413-1464, 1581-1621) -- one row per multipole goes through them instead of 24.  Held to the oracle at the same
1e-8 (row-scaled) as the template path, and to the template path of the same engine."""
import numpy as np
import pytest

from conftest import relerr
from eftpipe_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _engine(B, APst=False, Nl=3):
    import bench
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.tables import EngineConfig

    k = synth.survey_kgrid(bench.NK)
    cfg = EngineConfig(Nl=Nl, k=k, with_resum=True, with_ap=True, APst=APst, DA_AP=float(synth.da_func(synth.OM_AP, bench.Z)),
                       H_AP=float(synth.hubble(synth.OM_AP, bench.Z)))
    return Engine(cfg, max_batch=B), k


def _draws(B, seed, Nl=3):
    import bench
    from eftpipe_amd.parambasis import bias_row

    d = synth.draw_batch(B, z=bench.Z, seed=seed)
    rng = np.random.default_rng(seed)
    # a different bias vector per cosmology (the contraction runs inside the kernels: every row's coefficient must be its own)
    d["bs"] = [list(np.asarray(bench.BS) * (1.0 + 0.2 * rng.standard_normal(len(bench.BS)))) for _ in range(B)]
    d["bias"] = np.stack([bias_row(float(f), bs, None, bench.ES, kmA=0.7, krA=0.25, ndA=4.5e-5) for f, bs in zip(d["f"], d["bs"])])
    return d


def _plk(eng, d, B, direct):
    import bench

    eng.set_plk_direct(direct)
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    eng.run(eng.full_mask(reduce=True), B, sync=True)
    return eng.get("PLK", (B, eng.cfg.Nl, bench.NK)).copy()


@pytest.mark.parametrize("APst", [False, True])
def test_direct_plk_against_oracle_and_template_path(APst):
    """cfg-2 workload (Nl = 3, survey grid of 512 k, IR-resum + AP, seeded synthetic draws, one bias vector per cosmology): P_l of direct runs
    against the oracle's templates contracted with the same bias (reduce_Plk), and against the engine's own template path."""
    import bench
    from oracle import OracleConfig, OracleEngine

    B, ncheck = 16, 3
    eng, k = _engine(B, APst)
    d = _draws(B, 777)
    tm = _plk(eng, d, B, False)
    dr = _plk(eng, d, B, True)
    assert np.isfinite(dr).all()
    assert relerr(dr, tm) < 1e-9
    assert not np.array_equal(dr, tm)  # (it is another order of summation: identical bits would mean the option did nothing)
    orc = OracleEngine(OracleConfig(Nl=bench.NL, k=k, ndA=4.5e-5, with_resum=True, with_ap=True, APst=APst, Om_AP=synth.OM_AP, z_AP=bench.Z))
    for i in range(ncheck):
        f = float(d["f"][i])
        st = orc.evaluate(d["kin"], d["Pin"][i], f, float(d["DA"][i]), float(d["H"][i]), pairwise=True)
        ref = orc.reduce_plk(f, st, d["bs"][i], es=tuple(bench.ES))
        assert relerr(dr[i], ref) < TOL, i
        big = np.abs(ref) > 1e-3 * np.abs(ref).max(axis=-1, keepdims=True)
        assert np.max(np.abs(dr[i] - ref)[big] / np.abs(ref)[big]) < 1e-6, i  # the north-star bar, pointwise
    # back on the template path the engine returns what it returned before
    assert np.array_equal(_plk(eng, d, B, False), tm)
    eng.close()


def test_direct_plk_with_fallback_tiles():
    """Strong distortions: on the template path the tiles whose k'(mu) crosses more knots than the weight tables hold go through
    ap_direct_kernel; in direct runs ap_plk_kernel's window of piecewise polynomials overflows (more than 192 intervals) and the pieces are
    formed from global memory.  Against the template path of the same engine."""
    B = 8
    eng, _ = _engine(B)
    d = _draws(B, 31)
    qs = [(1.0, 1.0), (0.85, 1.15), (1.15, 0.85), (1.12, 1.12), (0.9, 0.9), (1.0, 1.1), (1.03, 0.999), (0.999, 1.0)]
    DAf, Hf = eng.cfg.DA_AP, eng.cfg.H_AP
    d["DA"] = np.array([q[0] * DAf for q in qs])
    d["H"] = np.array([Hf / q[1] for q in qs])
    tm = _plk(eng, d, B, False)
    dr = _plk(eng, d, B, True)
    assert np.isfinite(dr).all()
    for i in range(B):
        assert relerr(dr[i], tm[i]) < 1e-8, (i, qs[i])
    eng.close()


def test_direct_pipelined_steps_are_bit_identical_to_synchronous_runs():
    """Staged, overlapped direct steps with new inputs every step (three streams, rotating blocks, coefficient tables in two sets) return
    the bits of one synchronous direct run per input set; switching the option between steps of one engine is safe."""
    import bench

    B, K = 32, 13
    eng, _ = _engine(B)
    sets = [_draws(B, 500 + i) for i in range(3)]
    ref = [_plk(eng, s, B, True) for s in sets]
    tm = [_plk(eng, s, B, False) for s in sets]
    mask = eng.full_mask(reduce=True)
    eng.set_latency_mode(False)
    out = np.zeros((K, B, bench.NL, bench.NK))
    want = []
    for i in range(K):
        direct = i % 4 != 3  # every fourth step on the template path
        eng.set_plk_direct(direct)
        s = sets[i % 3]
        want.append((ref if direct else tm)[i % 3])
        eng.stage_inputs(s["Pin"], s["f"], s["DA"], s["H"], bias=s["bias"])
        eng.run_staged(mask, B)
        if i >= 5:  # five steps stay queued (the engine rotates eight sets: back <= 7)
            eng.fetch_previous("PLK", (B, bench.NL, bench.NK), out=out[i - 5], back=5)
    for back in (4, 3, 2, 1, 0):
        eng.fetch_previous("PLK", (B, bench.NL, bench.NK), out=out[K - 1 - back], back=back)
    eng.sync()
    for i in range(K):
        assert np.array_equal(out[i], want[i]), i
    eng.close()


def test_direct_option_leaves_other_runs_alone():
    """The option only concerns whole-pipeline runs that end in REDUCE on the fast AP path at Nl = 3: a run that stops at AP still produces the
    templates, and an Nl = 2 engine ignores it."""
    import bench
    from eftpipe_amd import _lib as L

    B = 4
    eng, _ = _engine(B)
    d = _draws(B, 5)
    eng.load_inputs(d["Pin"], d["f"], d["DA"], d["H"], d["bias"])
    eng.set_plk_direct(False)
    eng.run(eng.full_mask(reduce=False), B, sync=True)
    t0 = eng.get("TEMPL", (B, 3, 24, bench.NK)).copy()
    eng.set_plk_direct(True)
    eng.run(eng.full_mask(reduce=False), B, sync=True)
    assert np.array_equal(eng.get("TEMPL", (B, 3, 24, bench.NK)), t0)
    eng.close()
    eng2, _ = _engine(B, Nl=2)
    d2 = _draws(B, 6)
    a = _plk(eng2, d2, B, False)
    b = _plk(eng2, d2, B, True)
    assert np.array_equal(a, b)
    eng2.close()
