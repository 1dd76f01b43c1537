#!/usr/bin/env python3
"""BASELINE cfg 3 and cfg 5 on one GPU, as extra keys of bench.py (GPU box; no CPU fallback).

cfg 3 (reference cobaya/yamls/DR16_noric_LEX_NS_LP024_kmax0.20_EQ02_kmax0.20_XP024_kmax0.20.yaml:6-27, 63-70): a likelihood point = three
kernels (LRG, ELG chained, their cross spectrum; own P_lin / AP fiducial each) at Nl = 3, Nk = 512 through the DR16 windows at accboost 4 /
windowk 0.1 and the binning onto each tracer's data k.  Walker 0 of every step IS the point of tests/golden/cfg3_nk512.npz (inputs and P_l
written by the real reference, tools/make_fixtures.py cfg3_nk512), the other walkers are perturbations of it; its P_l as fetched inside the
timed loop is compared with the reference's.  Two rates: P_l per tracer back to the host (direct-P_l runs, the operators on one row per
cosmology) and the jointly marginalised ln P (templates first: LOGP needs 1 + n_G contracted rows; synthetic data vector around the model).

cfg 5's single-GPU half: Nk = 2048, two tracers (LRG, ELG chained), window at accboost 4 + binning; P_l of the LRG entries against the
reference's binned templates (tests/golden/cfg5_acc4.npz) contracted with the bias.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def _pointwise(got, want):
    big = np.abs(want) > 1e-3 * np.max(np.abs(want), axis=-1, keepdims=True)
    return float(np.max(np.abs(got - want)[big] / np.abs(want)[big]))


COALESCE = int(os.environ.get("EFTB_CFG_COALESCE", "4"))   # queued steps leave as one launch, as in bench.py's own loop


def _warm(run, ms=40.0, chunk=40):
    """Untimed warm-up: the loop about to be timed, run continuously (chunks of `chunk` steps over the warm-up draw sets) for `ms` of wall time -- a GPU
    that has just left idle runs its first milliseconds ~10 % below its sustained state (bench.py, warmup_note)."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        run(chunk)


def cfg3_rates(device=0, npoints=42, steps=20, depth=int(os.environ.get("EFTB_CFG_DEPTH", "12"))):
    import cfg3_util as U
    from eftpipe_amd import _lib as L
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.marginal import MarginalLikelihood, data_index
    from eftpipe_amd.parambasis import bias_row, gaussian_rows
    from eftpipe_amd.tables import EngineConfig
    from eftpipe_amd.window import window_matrix_device

    g = _golden("cfg3_nk512")
    k, NK, NTR = g["k"], g["k"].size, 3
    t0 = U.TRACERS[0]
    B = npoints * NTR
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, APst=True, DA_AP=float(g[t0 + "_DA_AP"]), H_AP=float(g[t0 + "_H_AP"])),
                 max_batch=B, device=device, coalesce=COALESCE)
    nb = max(g[t + "_kout"].size for t in U.TRACERS)
    ops = []
    for t in U.TRACERS:
        tab = U.window_table(t)
        _, p, _, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, windowk=float(g["windowk"]), accboost=int(g["accboost"]), device=device)
        Bm, _, _, _ = TB.binning_operator(k, g[t + "_kout"])
        op = TB.compose_operator(3, NK, Wfold=Wfold, binning=Bm, chained=U.CHAINED[t])
        full = np.zeros((3, 3, nb, NK))
        full[: op.shape[0], :, : op.shape[2]] = op
        ops.append(eng.add_operator(full))
    eng.set_tracers(NTR, ops)
    # inputs: walker 0 = the reference's point, walkers w > 0 = perturbations of it (new ones every step)
    Pin0 = np.stack([g[t + "_Pin"] for t in U.TRACERS])
    f0 = np.array([float(g[t + "_f"]) for t in U.TRACERS])
    DA0 = np.array([float(g[t + "_DA"]) * float(g[t0 + "_DA_AP"]) / float(g[t + "_DA_AP"]) for t in U.TRACERS])
    H0 = np.array([float(g[t + "_H"]) * float(g[t0 + "_H_AP"]) / float(g[t + "_H_AP"]) for t in U.TRACERS])
    p, sc = U.params(g), U.scales(g)

    def bias_of(f):
        out = []
        for i, t in enumerate(U.TRACERS):
            A, Bx = U.CROSS.get(t, (t, t))
            bsA = [p[A + "_b1"], p[A + "_b2"], 0.0, p[A + "_b4"], 0.0, 0.0, 0.0]
            bsB = [p[Bx + "_b1"], p[Bx + "_b2"], 0.0, p[Bx + "_b4"], 0.0, 0.0, 0.0] if t in U.CROSS else None
            out.append(bias_row(float(f[i]), bsA, bsB, (0.0, 0.0, 0.0), **sc[i]))
        return np.stack(out)

    def draw(step):
        rng = np.random.default_rng(9000 + step)
        amp = 1.0 + 0.03 * rng.uniform(-1, 1, (npoints, 1, 1))
        q = 1.0 + 0.01 * rng.uniform(-1, 1, (npoints, 2, 1))
        amp[0], q[0] = 1.0, 1.0
        Pin = (Pin0[None] * amp).reshape(B, -1)
        f = np.tile(f0, npoints)
        return dict(Pin=np.ascontiguousarray(Pin), f=f, DA=(DA0[None] * q[:, 0]).reshape(B), H=(H0[None] * q[:, 1]).reshape(B), bias=np.tile(bias_of(f0), (npoints, 1)))

    sets = [draw(s) for s in range(steps + 3)]
    out = {}
    # ---- (1) P_l of every tracer back to the host, direct-P_l runs, staged + fetched every step
    eng.set_latency_mode(False)
    eng.set_plk_direct(True)
    mask = eng.full_mask(reduce=True)
    shape = (B, 3, nb)
    worst = 0.0

    def loop(first, n, check, ring=0):
        nonlocal worst
        for i in range(n):
            d = sets[first + (i % ring if ring else i)]
            view = eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=depth if i >= depth else -1, shape=shape)
            if view is not None and check:
                for j, t in enumerate(U.TRACERS):
                    ref = g[t + "_plk"]
                    worst = max(worst, _pointwise(view[j][: ref.shape[0], : ref.shape[1]], ref))
        for back in range(min(depth, n) - 1, -1, -1):
            view = eng.fetch_previous("PLK", shape, back=back, copy=False)
            if check:
                for j, t in enumerate(U.TRACERS):
                    ref = g[t + "_plk"]
                    worst = max(worst, _pointwise(view[j][: ref.shape[0], : ref.shape[1]], ref))
        eng.sync()

    _warm(lambda n: loop(steps, n, False, ring=3))
    t1 = time.perf_counter()
    loop(0, steps, True)
    dt = (time.perf_counter() - t1) / steps
    assert worst < 1e-6, f"cfg 3: P_l of the reference's point differs from tests/golden/cfg3_nk512.npz by {worst:.2e}"
    out["cfg3_plk_points_per_s"] = npoints / dt
    out["cfg3_plk_evaluations_per_s"] = B / dt
    out["cfg3_plk_max_rel_err_vs_reference"] = worst
    # ---- (2) the jointly marginalised ln P per point (templates first), staged, one float per point back
    eng.set_plk_direct(False)
    nG = 17
    ngL, ngE = (p["LRG_NGC_b1"], p["LRG_NGC_b2"], p["LRG_NGC_b4"]), (p["ELG_NGC_b1"], p["ELG_NGC_b2"], p["ELG_NGC_b4"])
    rows = np.zeros((B, nG + 1, 24))
    for w in range(npoints):
        eL, eE, eX = w * NTR, w * NTR + 1, w * NTR + 2
        rL, rE = gaussian_rows(f0[0], ngL, None, **{k_: v for k_, v in sc[0].items() if k_.endswith("A")}), gaussian_rows(f0[1], ngE, None, **{k_: v for k_, v in sc[1].items() if k_.endswith("A")})
        rX = gaussian_rows(f0[2], ngL, ngE, **sc[2])
        rows[eL, 0], rows[eL, 1:8] = rL[0], rL[1:]
        rows[eE, 0], rows[eE, 8:15] = rE[0], rE[1:]
        rows[eX, 0], rows[eX, 1:5], rows[eX, 8:12], rows[eX, 15:18] = rX[0], rX[1:5], rX[5:9], rX[9:12]
    nds = [g[t + "_kout"].size for t in U.TRACERS]
    index = np.concatenate([l * nb + np.arange(nds[j]) + (j * 3) * nb for j, t in enumerate(U.TRACERS) for l in range(len(g[t + "_ls"]))]).astype(np.int32)
    d0 = sets[0]
    templ = eng.eval_batch(d0["Pin"][:NTR], d0["f"][:NTR], d0["DA"][:NTR], d0["H"][:NTR])
    model = np.concatenate([np.einsum("r,lrx->lx", rows[t, 0], templ[t]) for t in range(NTR)]).reshape(-1)[index]
    sig = 0.05 * np.abs(model) + 10.0
    MarginalLikelihood(eng, index, model * 1.01, np.diag(1.0 / sig**2), np.zeros(nG), np.full(nG, 2.0))
    lmask = eng.full_mask() | L.S_LOGP

    def lloop(first, n, ring=0):
        last = None
        for i in range(n):
            d = sets[first + (i % ring if ring else i)]
            v = eng.step(lmask, d["Pin"], d["f"], d["DA"], d["H"], rows=rows, back=depth if i >= depth else -1, fetch="LOGP", shape=(npoints, 26))
            last = v if v is not None else last
        for back in range(min(depth, n) - 1, -1, -1):
            last = eng.fetch_previous("LOGP", (npoints, 26), back=back, copy=False)
        eng.sync()
        return last

    _warm(lambda n: lloop(steps, n, ring=3))
    t1 = time.perf_counter()
    lp = lloop(0, steps)
    dt = (time.perf_counter() - t1) / steps
    assert np.all(np.isfinite(lp[:, 0])), "cfg 3: non-finite marginalised ln P"
    lp = lp.copy()
    d = sets[steps - 1]   # the last timed step once more, by itself: same bits whatever launch it left in
    eng.step(lmask, d["Pin"], d["f"], d["DA"], d["H"], rows=rows)
    alone = eng.fetch_previous("LOGP", (npoints, 26), back=0, copy=False)
    assert np.array_equal(alone, lp), "cfg 3: ln P of a step launched with others differs from the same step by itself"
    out["cfg3_likelihood_points_per_s"] = npoints / dt
    out["cfg3_likelihood_theory_evaluations_per_s"] = B / dt
    out["cfg3_note"] = (f"{npoints} likelihood points x 3 tracers (LRG, ELG chained, X) per step at Nl = 3, Nk = 512, DR16 windows at accboost {int(g['accboost'])} / windowk "
                        f"{float(g['windowk'])}, binned onto {nds} data k; new inputs staged every step, {depth} steps ahead of the one fetched, Engine(coalesce={COALESCE}).  cfg3_plk_*: P_l of every tracer fetched (direct-P_l runs, "
                        "operators on one row per cosmology); walker 0 of every step is the point of tests/golden/cfg3_nk512.npz and its P_l is compared with the "
                        f"reference's inside the loop (max pointwise {worst:.1e}).  cfg3_likelihood_*: {index.size} data points, {nG} jointly marginalised parameters, "
                        "one ln P per point fetched (templates first; synthetic data vector around the model)")
    eng.close()
    return out


def cfg5_rate(device=0, walkers=64, steps=int(os.environ.get("EFTB_CFG5_STEPS", "20")), depth=int(os.environ.get("EFTB_CFG5_DEPTH", "10"))):
    from eftpipe_amd import synth
    from eftpipe_amd import tables as TB
    from eftpipe_amd.engine import Engine
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig
    from eftpipe_amd.window import window_matrix_device

    g, f = _golden("cfg5_acc4"), _golden("caseF")
    k, NK, NTR = f["k"], f["k"].size, 2
    B = walkers * NTR
    eng = Engine(EngineConfig(Nl=3, k=k, with_resum=True, with_ap=True, DA_AP=float(f["DA_AP"]), H_AP=float(f["H_AP"])), max_batch=B, device=device, coalesce=COALESCE)
    Bm, _, _, _ = TB.binning_operator(k, g["kout"])
    nb = g["kout"].size
    ops = []
    for name, chained in (("LRG", False), ("ELG", True)):
        tab = np.load(os.path.join(GOLD, f"win_NGC_{name}_sQ024.npy"))
        _, _, _, Wfold = window_matrix_device(k, tab[:, 0], tab[:, 1:].T, 3, 3, windowk=float(g["windowk"]), accboost=int(g["accboost"]), device=device)
        op = TB.compose_operator(3, NK, Wfold=Wfold, binning=Bm, chained=chained)
        full = np.zeros((3, 3, nb, NK))
        full[: op.shape[0]] = op
        ops.append(eng.add_operator(full))
        del Wfold
    eng.set_tracers(NTR, ops)
    fg = float(f["f"])
    bs = [2.1, 0.6, 0.8, 0.5, -1.8, -1.9, -1.5]
    bias1 = bias_row(fg, bs, None, (0.3, 0.1, -0.9), kmA=0.7, krA=0.25, ndA=4.5e-5)
    T = np.concatenate([g["LRG_binned_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl")], axis=1)
    want = np.einsum("r,lrx->lx", bias1, T)

    def draw(step):
        rng = np.random.default_rng(700 + step)
        amp = 1.0 + 0.03 * rng.uniform(-1, 1, (B, 1))
        q = 1.0 + 0.01 * rng.uniform(-1, 1, (B, 2))
        amp[0], q[0] = 1.0, 1.0
        return dict(Pin=np.ascontiguousarray(f["Pin"][None] * amp), f=np.full(B, fg), DA=float(f["DA"]) * q[:, 0], H=float(f["H"]) * q[:, 1], bias=np.tile(bias1, (B, 1)))

    sets = [draw(s) for s in range(steps + 2)]
    eng.set_latency_mode(False)
    eng.set_plk_direct(True)
    mask = eng.full_mask(reduce=True)
    shape = (B, 3, nb)
    worst = 0.0

    def loop(first, n, check, ring=0):
        nonlocal worst
        for i in range(n):
            d = sets[first + (i % ring if ring else i)]
            view = eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=depth if i >= depth else -1, shape=shape)
            if view is not None and check:
                worst = max(worst, _pointwise(view[0], want))
        for back in range(min(depth, n) - 1, -1, -1):
            view = eng.fetch_previous("PLK", shape, back=back, copy=False)
            if check:
                worst = max(worst, _pointwise(view[0], want))
        eng.sync()

    _warm(lambda n: loop(steps, n, False, ring=2))
    t1 = time.perf_counter()
    loop(0, steps, True)
    dt = (time.perf_counter() - t1) / steps
    assert worst < 1e-6, f"cfg 5: P_l of the reference's cosmology differs from tests/golden/cfg5_acc4.npz by {worst:.2e}"
    eng.close()
    return {"cfg5_evaluations_per_s": B / dt, "cfg5_ms_per_step": dt * 1e3, "cfg5_max_rel_err_vs_reference": worst,
            "cfg5_note": (f"single-GPU half of cfg 5: Nk = {NK}, Nl = 3, IR-resum + AP, two tracers (LRG, ELG chained) x {walkers} walkers per step, window at accboost "
                          f"{int(g['accboost'])} / windowk {float(g['windowk'])} + binning onto {nb} data k, direct-P_l runs, new inputs staged and P_l fetched every step ({depth} steps ahead, Engine(coalesce={COALESCE})); entry 0 "
                          "of every step is the cosmology of tests/golden/cfg5_acc4.npz (the real reference's binned LRG templates, contracted with the bias) and is "
                          f"compared inside the loop (max pointwise {worst:.1e}); the ELG window has no reference fixture at this accuracy")}


if __name__ == "__main__":
    import json

    if len(sys.argv) > 1 and sys.argv[1] == "--json":   # bench.py's child process: one JSON line with both
        dev = int(sys.argv[2]) if len(sys.argv) > 2 else 0
        out = cfg3_rates(dev)
        out.update(cfg5_rate(dev))
        print(json.dumps(out))
    else:
        print(json.dumps(cfg3_rates(), indent=1))
        print(json.dumps(cfg5_rate(), indent=1))
