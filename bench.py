#!/usr/bin/env python3
"""Headline benchmark: theory P_l(k) evaluations / second (single tracer, Nk=512, l=0,2,4) -- SURVEY.md 8(d).

Workload = BASELINE.json configs[1]: LRG z=0.7, Nl=3, Nk=512, IR-resummation + AP, SYNTH-PLIN v1 draws, `--batch` cosmologies per GPU
per step.  A step = one pass of the hot path (reference theory.py:557-609: FFTLog coefficients -> anti-diagonal sums ->
P22/P13/C11/Cct/C22/C13 -> regroup -> resum -> AP -> bias contraction) over one batch of NEW inputs:

    timed region, per step:  eftb_step = eftb_stage_inputs (Pin, f, DA, H, bias rows of a draw set never seen before -> page-locked block)
                                         eftb_run_staged   (all stages; issued by the library's submission thread, queued steps leave as one launch)
                                         eftb_fetch_view   (P_l of the step DEPTH back, in page-locked host memory)   [N > 1: RCCL gather to rank 0]
                             N = 1: every step names its slice of the sampler's own page-locked results array as the destination of its P_l
                             (eftb_set_step_output): the copy-out behind the step writes it, the fetch only waits for it

so `value` is the input-to-output rate a sampler sees (H2D + D2H inclusive, every step's P_l lands in host memory inside the timed
region; pipeline fill and drain are inside it too).  All draws are generated before the clock starts.  The rate of the same kernels
over inputs that stay resident in HBM, and the rate with the full template block coming back, are extra keys.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line carrying `roofline` (the kernel with the largest time per launch inside the timed region -- on direct-P_l runs the
resummation kernel resum_plk_kernel -- timed live with HIP events on its own stream; `frac` = EXECUTED FP64 flops / time / peak, the executed count
derived from the compiled kernel's instruction counts -- eftpipe_amd/csrc/isa_counts.json, written by tools/isa_counts.py at build time -- and from
the batch each bracketed launch carried; `frac_alone`: the same kernel with nothing beside it; `roofline_step`: all launches of a step) and
`cpu_baseline` (the NumPy oracle, "port", timed on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NL, NK, Z = 3, 512, 0.7
BS = [2.1401334, 0.77616816 / np.sqrt(2.0), 0.77003455, 0.77616816 / np.sqrt(2.0), -1.8396613, -1.8918368, -1.4856405]
ES = (0.26033594, 0.0, -0.92895016)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix peak (not in the local guide; measured issue rate reported beside it)
NPOW, NS_DEV, NKLOW = 257, 80, 7
COALESCE = int(os.environ.get("EFTB_BENCH_COALESCE", "4"))  # steps the engine may launch together when they are still queued (Engine(coalesce=...): device state for COALESCE x batch)
DEPTH = int(os.environ.get("EFTB_BENCH_DEPTH", "12"))  # steps queued on the GPU ahead of the one being fetched (1..6; direct-P_l steps are four pipeline stages deep: 618-650 k evaluations/s at 2, 660-668 k at 3, 663-695 k at 4; templates first, round 2: measured 276-279 k evaluations/s at 2, 279 k at 3 over 40 steps, and 273 k vs 266 k over 20: the drain is longer)
DIRECT = os.environ.get("EFTB_BENCH_DIRECT", "1") != "0"   # `value` on direct-P_l runs (EFTB_O_PLK_DIRECT); 0: templates first


STEP_TIMES = [] if os.environ.get("EFTB_BENCH_STEP_TIMES") else None
HOST_PROFILE = [0.0, 0.0] if os.environ.get("EFTB_BENCH_HOST_PROFILE") else None  # diagnostics: seconds of the timed loop inside eng.step() / inside keeper.put()  # diagnostics: host time at which each timed step's P_l had been fetched


def cpu_baseline(picks, budget_s=20.0):
    """Time the oracle (NumPy restatement of the reference path) on this host, bounded sample.  `picks` = cosmologies taken from the draw
    sets of the timed loop (dicts with kin, Pin, f, DA, H): the sample cycles through them, and the oracle's templates of the first two come
    back for the parity check of the timed loop's own outputs."""
    from eftpipe_amd import synth
    from oracle import OracleConfig, OracleEngine

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    k = synth.survey_kgrid(NK)
    orc = OracleEngine(OracleConfig(Nl=NL, k=k, ndA=4.5e-5, with_resum=True, with_ap=True, Om_AP=synth.OM_AP, z_AP=Z))
    kept = {}

    def run(pairwise, budget, at_least):
        n, t0 = 0, time.perf_counter()
        while True:
            i = n % len(picks)
            c = picks[i]
            st = orc.evaluate(c["kin"], c["Pin"], float(c["f"]), float(c["DA"]), float(c["H"]), pairwise=pairwise)
            if i not in kept:
                kept[i] = np.concatenate([st["P11l"], st["Pctl"], st["Ploopl"], st["Pstl"]], axis=1)  # [Nl, 24, Nk] in the engine's row order
            n += 1
            el = time.perf_counter() - t0
            if (el > budget and n >= at_least) or n >= 64:
                return n, el

    n1, t1 = run(False, budget_s * 0.6, 2)   # (at least the two parity cosmologies, whatever the host's speed)
    n2, t2 = run(True, budget_s * 0.4, 1)
    templ = [kept[0], kept[1]]
    return {
        "value": n1 / t1, "unit": "evaluations/s", "cores": int(threads), "kind": "port",
        "sample": f"{n1} evaluations of the cfg-2 workload (cosmologies of the timed draw sets) with the reference's einsum paths (as-is) in {t1:.1f}s; "
                  f"with the pairwise P22 path forced: {n2 / t2:.3f} evaluations/s ({n2} in {t2:.1f}s)",
        "value_pairwise_path": n2 / t2,
    }, templ


def executed_flops_per_launch(B):
    """FP64 flops one launch of resum_mfma_kernel issues, from the compiled kernel's own instruction counts: per trip of its s loop
    (one wave, 16 k x 1 s) `flops_per_wave_trip` = 2048 per v_mfma_f64_16x16x4 + 64 lanes x (2 per v_fma/v_fmac_f64, 1 per v_mul/v_add_f64),
    times waves (4 per 64 k of the resummed range, per cosmology) times 80 trips."""
    path = os.path.join(ROOT, "eftpipe_amd", "csrc", "isa_counts.json")
    name = "resum_mfma_kernel<0>"  # the build the engine launches (<1>: with the NNLO accumulators)
    with open(path) as fh:
        allinfo = json.load(fh)
    info = allinfo[name]
    from eftpipe_amd import _lib as L

    lib_hash = L.load().eftb_source_hash().decode()
    if allinfo.get("_source_hash") != lib_hash:  # e.g. EFTB_LIB points at another build: the counts do not describe the code that runs
        raise SystemExit(f"bench.py: isa_counts.json describes sources {allinfo.get('_source_hash')}, the loaded library was built from {lib_hash}; "
                         "run `python -c 'import __graft_entry__ as g; g.build()'` (or tools/isa_counts.py) for this build")
    loop = max(info["loops"], key=lambda b: b["mfma"] + b["valu_f64"])
    waves = ((NK - (NKLOW & ~15) + 63) // 64) * 4 * B  # k tiles start at a multiple of 16
    per_trip = {"kernel_build": name, "mfma": loop["mfma"], "valu_f64": loop["valu_f64"], "f64_ops": loop["f64_ops"], "flops_per_wave_trip": loop["flops_per_wave_trip"],
                "vgprs": info.get("vgprs"), "scratch_bytes": info.get("scratch_bytes")}
    return float(loop["flops_per_wave_trip"]) * waves * NS_DEV, float(loop["mfma_flops_per_wave_trip"]) * waves * NS_DEV, per_trip


def synth_flops(t, B, direct):
    """2 M N K of the products ONE synth_kernel launch of a whole-pipeline step carries, with the rows the engine queues (eftbird.hip, the
    queue_synth calls of launch_stages_impl): templates first 7 basis rows of P22, Nl (7 + 2) weighted xi rows, 10 of P13, Nl + Nl of C11 / Cct per
    cosmology; a direct-P_l run queues 3 contracted rows in each of the first three."""
    nb, KS, KL = t["comb22"].shape[1], t["syn_k"].shape[0], t["lin_k"].shape[0]
    ncf = NL * (nb + t["comb13"].shape[1])
    r22, rcf, r13 = (3, 3, 3) if direct else (nb, ncf, 10)
    flops = 2.0 * B * (r22 * NK * KS + rcf * NS_DEV * KS + r13 * NK * KL + 2 * NL * NS_DEV * KL)
    note = f"2 M N K: [{B} x {r22}] x {KS} x {NK} + [{B} x {rcf}] x {KS} x {NS_DEV} + [{B} x {r13}] x {KL} x {NK} + [{B} x {2 * NL}] x {KL} x {NS_DEV}"
    return flops, note


def ap_slots_per_k(d, k, DA_fid, H_fid, nmu=200):
    """mean number of knot intervals k'(mu) = k / q_perp sqrt(1 + mu^2 (1 / F^2 - 1)) crosses per (cosmology, k) of draw set d: what the moment-form AP
    kernel of a direct-P_l run does its work per (the executed flops of ap_plk_fused_kernel scale with it)"""
    qperp, qpar = d["DA"] / DA_fid, H_fid / d["H"]
    g = 1.0 / (qpar / qperp) ** 2 - 1.0
    mu = np.linspace(0.0, 1.0, nmu)
    root = np.sqrt(1.0 + mu[None, :] ** 2 * g[:, None])
    lo, hi = root.min(axis=1), root.max(axis=1)
    kq = k[None, :] / qperp[:, None]
    i0 = np.clip(np.searchsorted(k, kq * lo[:, None], side="right") - 1, 0, k.size - 2)
    i1 = np.clip(np.searchsorted(k, kq * hi[:, None], side="right") - 1, 0, k.size - 2)
    return float(np.mean(i1 - i0 + 1))


AP_FLOPS_PER_SLOT = 3 * 20 + 9 * 12   # per (k, interval): the cubic of each l' re-expanded in rho (10 FMAs), 9 x (4 differences + 4 FMAs)


def direct_step_flops(eng, B, allinfo, slots_per_k):
    """FP64 flops the launches of ONE direct-P_l step of B cosmologies execute (what `roofline_step` divides by the step time): per kernel, from the
    compiled loops (tools/isa_counts.py) where the kernel is a counted loop, else from the shape of its sums."""
    t = eng.tables
    kp = lambda n: (n + 47) // 48 * 48
    Nkin, ntail, nxt = t["kin"].size, t["lnx_tail"].size, t["lnx_xtail"].size
    nmu = int(np.asarray(t["mu"]).size)
    out = {}
    info = allinfo["resum_plk_kernel<4,4>"]
    loop = max(info["loops"], key=lambda b: b["valu_f64"])
    out["resum_plk_kernel"] = float(loop["flops_per_wave_trip"]) * B * 3 * ((NK + 255) // 256) * 12 * (NS_DEV // 8)   # 3 v x 4 slices of s = 12 waves per (256 k, l, cosmology), two s per trip
    out["synth_kernel"] = synth_flops(t, B, direct=True)[0]
    # moment-form AP stage in one launch: prefix sums (36 sequences x nmu nodes, 3 flop, twice per cosmology), pieces (3 x (Nk - 1) x 32 flop, twice), the walk
    out["ap_plk_fused_kernel"] = float(B) * (2 * 36 * nmu * 3 + 2 * 3 * (NK - 1) * 32 + NK * slots_per_k * AP_FLOPS_PER_SLOT)
    nb = t["comb22"].shape[1] + t["comb13"].shape[1]
    out["antidiag_kernel"] = 8.0 * B * nb * (NPOW * (NPOW + 1) // 2)     # one complex multiply-add per (matrix, pair n <= m) and cosmology
    out["gemm_direct_kernel"] = 2.0 * B * (kp(Nkin) * NK + 2 * kp(Nkin + ntail) * 2 * 129 + kp(Nkin + nxt) * 2 * NS_DEV)  # P11, coefficients (+ transpose), X / Y
    out["spline_kernel"] = 2.0 * B * NL * NK * 65                         # banded operator, half-width 32
    out["build_rows_plk_kernel"] = 8.0 * B * NPOW * (3 * nb + 3 * NL * nb + 3 + 2 * NL)  # contraction of the anti-diagonal sums into 3 + 3 rows, the single-sum rows
    out["back_prep_plk_kernel"] = 2.0 * B * (NL * NK * 12 + NS_DEV * 160 * 6)             # regrouping of the contracted rows; coefficient table (6 products per entry)
    return out


def direct_rooflines(eng, cfg, B, ktimes, d0, templates_first, peak_tflops, measured_peak):
    """`roofline` of a direct-P_l run (EFTB_O_PLK_DIRECT).  Its step has no single dominant kernel any more: the resummation (resum_plk_kernel,
    FP64 vector work with scalar coefficients), the synthesis GEMMs of the loop stages (synth_kernel, FP64 MFMA) and the AP quadrature
    (ap_plk_kernel, FP64 vector, issue / latency bound) each take about a fifth of the kernel time.  All three were bracketed with HIP events on their own
    streams inside the timed region; the one with the largest time per launch is `roofline`, the other two follow in `roofline_others`, the
    templates-first step's dominant kernel (resum_mfma_kernel, timed alone) in `roofline_templates_first`."""
    from eftpipe_amd import _lib as L

    t = eng.tables
    path = os.path.join(ROOT, "eftpipe_amd", "csrc", "isa_counts.json")
    with open(path) as fh:
        allinfo = json.load(fh)
    pmc, pmc_name = {}, None
    for name in ("r04_pmc_per_kernel.json", "r03_pmc_per_kernel.json"):
        pmc_path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc, pmc_name = json.load(fh), name
            break

    def traffic_of(prefix, Bl=None):
        # HBM bytes per launch from the committed counter runs (separate --pmc passes at 128 cosmologies per launch), scaled to this launch's batch
        for name, v in pmc.items():
            if name.startswith("eftb::" + prefix) and isinstance(v, dict) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                scale = (Bl / float(pmc.get("_batch", 128))) if Bl else 1.0
                return (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 * scale, f"profiles/{pmc_name} (2 FETCH_SIZE + WRITE_SIZE, x {scale:.2f} for the batch of this launch)"
        return None, None

    entries = []
    # (0) resum_plk_kernel<4, 4>: executed FP64 vector flops from the compiled loop (two s steps per trip), waves = B x 3 l x Nk / 256 x 12, NS / 8 trips
    info = allinfo["resum_plk_kernel<4,4>"]
    loop = max(info["loops"], key=lambda b: b["valu_f64"])
    ms, n, nc = ktimes[0]
    Bl = nc / n if n else B   # cosmologies per timed launch (coalesced staged steps leave as one launch)
    waves, trips = Bl * 3 * ((NK + 255) // 256) * 12, NS_DEV // 8
    flops = float(loop["flops_per_wave_trip"]) * waves * trips
    if n:
        tr, src = traffic_of("resum_plk_kernel", Bl)
        entries.append({"bound": "valu_f64", "kernel": "resum_plk_kernel<4, 4> (Resum.Ps of a direct-P_l run: nine Horner chains of degree 15 per (k, s) with scalar coefficients; FP64 vector "
                                                   "instructions, which share the DP pipe -- and its 78.6 TFLOP/s peak -- with the matrix cores)",
                        "achieved": flops / (ms / n * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s", "frac": flops / (ms / n * 1e-3) / 1e12 / peak_tflops,
                        "traffic": tr, "traffic_source": src, "ms_per_launch": ms / n, "launches_timed": n, "cosmologies_per_launch": Bl, "executed_flops_per_launch": flops,
                        "instructions_per_wave_trip": {"f64_ops": loop["f64_ops"], "valu_f64": loop["valu_f64"], "flops_per_wave_trip": loop["flops_per_wave_trip"],
                                                       "vgprs": info.get("vgprs")},
                        "counts_source": "eftpipe_amd/csrc/isa_counts.json (tools/isa_counts.py, from the compiled gfx950 assembly)"})
    # (1) synth_kernel: 2 M N K of the four products of the launch (P22 basis rows, xi basis rows, P13, C11 / Cct)
    ms, n, nc = ktimes[1]
    Bl = nc / n if n else B
    flops, note = synth_flops(t, Bl, direct=True)
    if n:
        tr, src = traffic_of("synth_kernel", Bl)
        entries.append({"bound": "mfma", "kernel": "synth_kernel (makeP22 / makeC22 / makeC13 / makeP13 / makeC11 / makeCct as FP64-MFMA GEMMs in one launch: the synthesis of the "
                                                   "anti-diagonal sums at every k and s; in a direct-P_l run the rows arrive contracted with the bias: 3 per cosmology and product)",
                        "achieved": flops / (ms / n * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s", "frac": flops / (ms / n * 1e-3) / 1e12 / peak_tflops,
                        "traffic": tr, "traffic_source": src, "ms_per_launch": ms / n, "launches_timed": n, "cosmologies_per_launch": Bl, "algorithmic_flops_per_launch": flops,
                        "flops_note": note})
    # (2) the AP stage of a direct-P_l run: ap_plk_fused_kernel (moment form: prefix sums over mu, spline pieces and the interval walk in LDS); its
    # executed FP64 work is a few dozen flops per (k, interval crossed) -- the kernel is bound by LDS round trips, not by arithmetic
    nmu = int(np.asarray(t["mu"]).size)
    ms, n, nc = ktimes[2]
    Bl = nc / n if n else B
    slots = ap_slots_per_k(d0, np.asarray(cfg.k), cfg.DA_AP, cfg.H_AP, nmu)
    flops = float(Bl) * (2 * 36 * nmu * 3 + 2 * 3 * (NK - 1) * 32 + NK * slots * AP_FLOPS_PER_SLOT)
    if n:
        tr, src = traffic_of("ap_plk_fused_kernel", Bl)
        entries.append({"bound": "valu_f64", "kernel": "ap_plk_fused_kernel<3> (APeffect.AP of a direct-P_l run in moment form: the mu quadrature of P_l'(k'(mu)) L_l'(mu') L_l(mu) on one row "
                                                       "per multipole as prefix sums over mu x the cubic pieces of the intervals k'(mu) crosses, all tables in LDS; latency-bound, not arithmetic-bound)",
                        "achieved": flops / (ms / n * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s", "frac": flops / (ms / n * 1e-3) / 1e12 / peak_tflops,
                        "traffic": tr, "traffic_source": src, "ms_per_launch": ms / n, "launches_timed": n, "cosmologies_per_launch": Bl, "executed_flops_per_launch": flops,
                        "flops_note": f"{Bl:.0f} x ({NK} k x {slots:.2f} intervals x {AP_FLOPS_PER_SLOT} flop + 2 x 36 x {nmu} x 3 prefix terms + 2 x 3 x {NK - 1} x 32 for the pieces)"})
    if not entries:
        return templates_first
    entries.sort(key=lambda r: -r["ms_per_launch"])
    top = dict(entries[0])
    top["ms_per_launch_note"] = ("HIP events on the kernel's own stream around every second of its launches inside the timed region, where the kernels of the other three "
                                 "streams share the CUs with it; the three heaviest kernels of a direct-P_l step were timed this way (the other two in an untimed repeat of the loop), this is the one with the largest time per launch")
    top["measured_mfma_f64_issue_peak_tflops"] = measured_peak
    top["roofline_others"] = entries[1:]
    top["roofline_templates_first"] = templates_first
    return top


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="cosmologies per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra rates (resident loop, templates back, drop-in latency)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): this process becomes the launcher -- N FRESH child processes, one rank per GPU, with the
        # environment torch.distributed.run would have set.  Nothing here has touched HIP yet, and nothing is exec'ed: children are spawned.
        raise SystemExit(spawn_ranks(args.gpus))

    from eftpipe_amd import _lib as L
    from eftpipe_amd import dist, synth
    from eftpipe_amd.engine import Engine, comm_unique_id, mfma_f64_peak
    from eftpipe_amd.parambasis import bias_row
    from eftpipe_amd.tables import EngineConfig

    cp = dist.ControlPlane()
    rank, world = cp.rank, cp.world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("EFTB_BENCH_RENDEZVOUS_ONLY"):  # launch-path check for boxes without a GPU (tests/test_host_logic.py): every rank joins, rank 0 reports
        cp.barrier()
        ranks = cp.max(float(rank)) + 1
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "highest_rank_seen": int(ranks) - 1}))
        cp.barrier()
        cp.close()
        return
    B, K, W = args.batch, args.steps, args.warmup
    cfg = EngineConfig(Nl=NL, k=synth.survey_kgrid(NK), with_resum=True, with_ap=True,
                       DA_AP=float(synth.da_func(synth.OM_AP, Z)), H_AP=float(synth.hubble(synth.OM_AP, Z)))
    # host tables with a bounded BLAS pool: the init-time NumPy work must not eat the CPU share the launch loop needs right after
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover
        import contextlib

        threadpool_limits = lambda limits: contextlib.nullcontext()
    device = cp.local_rank
    shared_device = False
    if world > 1:
        ndev = int(os.environ.get("EFTB_VISIBLE_DEVICES", "0")) or _device_count()
        if 0 < ndev <= cp.local_rank:  # fewer GPUs than ranks: ranks share devices (a rehearsal of the launch path, not a measurement)
            device, shared_device = cp.local_rank % ndev, True
            print(f"[bench] rank {rank}: only {ndev} GPU(s) visible, sharing device {device}", file=sys.stderr)
    with threadpool_limits(limits=8):
        eng = Engine(cfg, max_batch=B, device=device, coalesce=COALESCE)
    shared_device = cp.max(1.0 if shared_device else 0.0) > 0.0   # any rank sharing a GPU makes the run a rehearsal
    exchange = "none"
    force_comm = world == 1 and os.environ.get("EFTB_BENCH_FORCE_COMM") == "1"  # one-GPU rehearsal of the N > 1 loop (RCCL self exchange)
    if world > 1 or force_comm:
        # RCCL communicator for the P_l gather; if it cannot be built on this node every rank agrees to fall back to a host-side
        # gather over the control plane so that the run still completes -- flagged "valid": false, never to be read as an RCCL number
        ok = 1.0
        try:
            uid = comm_unique_id() if rank == 0 else None
        except Exception as exc:  # pragma: no cover
            uid, ok = None, 0.0
            print(f"[bench] rank {rank}: RCCL unavailable: {exc}", file=sys.stderr)
        uid = cp.broadcast_bytes(uid)
        if uid is not None:
            try:
                eng.comm_init(world, rank, uid)
            except Exception as exc:  # pragma: no cover
                ok = 0.0
                print(f"[bench] rank {rank}: ncclCommInitRank failed: {exc}", file=sys.stderr)
        else:
            ok = 0.0
        exchange = "rccl" if cp.max(1.0 - ok) == 0.0 else "host-fallback"

    # ---- every draw set of the run, generated before the clock starts: W + K sets of B new cosmologies per rank
    def draw_set(i, nb=None):
        d = synth.draw_batch(nb or B, z=Z, seed=12345 + 7919 * i + 104729 * rank)
        d["bias"] = np.stack([bias_row(float(f), BS, None, ES, kmA=0.7, krA=0.25, ndA=4.5e-5) for f in d["f"]])
        return d

    sets = [draw_set(i) for i in range(W + K)]
    # the untimed extension of the warm-up (below) cycles through a ring of sixteen draw sets of its own (the W asked for + as many more as it takes)
    NRING = 16
    ring_sets = (sets[:W] + [draw_set(W + K + j) for j in range(max(0, NRING - W))])[:NRING] if W > 0 else []
    # ... and K more for the side key bare_warmup_*: the same K-step loop timed right behind the W warm-up steps asked for, before the extension
    bare_sets = [draw_set(W + K + NRING + j) for j in range(K)] if W > 0 and os.environ.get("EFTB_BENCH_BARE", "1") != "0" else []
    mask = eng.full_mask(reduce=True)
    # every timed step's P_l of this rank.  N = 1: a page-locked array the sampler owns -- each step's copy-out writes its slice directly
    # (eng.step(out=...) = eftb_set_step_output: no second host copy; EFTB_BENCH_OWN_OUTPUT=0: the engine's own host block and a copy by a second
    # host thread, the form up to round 4's first session -- 1.6 MB per step at 10-20 GB/s of one thread: 0.4 ms of a 20-step run)
    OWN_OUT = exchange == "none" and os.environ.get("EFTB_BENCH_OWN_OUTPUT", "1") != "0" and K * B * NL * NK * 8 <= (2 << 30)
    results = eng.pinned_empty((K, B, NL, NK)) if OWN_OUT else np.zeros((K, B, NL, NK))
    results.fill(0.0)                                                     # (touched: the sampler's output buffers exist before the clock starts, no first-touch page faults inside it)
    warm_out = eng.pinned_empty((max(W, NRING), B, NL, NK)) if OWN_OUT else None   # where the untimed warm-up steps deliver
    # (multi-GPU: the root takes every step's gathered block [world, B, NL, NK] as a view of the engine's page-locked host copy -- 12.6 MB per
    # step at 8 ranks, more than one host thread can copy again in a step's time -- and keeps a copy of its own rank's slice for the check below)

    def stage_and_run(d):
        eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
        eng.run_staged(mask, B)

    class Keeper:
        """The sampler's use of a finished step: P_l arrives in page-locked host memory behind the step (DMA); fetch_previous(copy=False) waits for it and
        hands out that block, and a second host thread copies it into `results` (the bookkeeping of this benchmark: every step is compared with the
        synchronous path afterwards) while the first one stages and launches the next step -- one Python thread doing both is slower than the GPU
        since the direct-P_l runs (0.21 ms of host work per step against 0.16 ms of GPU work).  The block is reused four steps later; the queue
        holds at most one pending copy."""

        def __init__(self):
            import queue
            import threading

            self.q = queue.Queue(maxsize=1)
            self.t = threading.Thread(target=self.run, daemon=True)
            self.t.start()

        def run(self):
            while True:
                view, idx = self.q.get()
                np.copyto(results[idx], view)
                self.q.task_done()

        def put(self, view, idx):
            self.q.put((view, idx))

        def join(self):
            self.q.join()

    keeper = None if OWN_OUT else Keeper()

    def take(idx, back, keep):
        view = eng.fetch_previous("PLK", (B, NL, NK), back=back, copy=False)   # (waits for the step; with OWN_OUT the view IS the step's slice of `results`)
        if keep and keeper is not None:
            keeper.put(view, idx)
        if keep and STEP_TIMES is not None:
            STEP_TIMES.append(time.perf_counter())

    def loop(first, n, keep, src=None):
        """n pipelined steps over sets[first : first + n] (src: over that ring of draw sets instead, round and round); every step's output is fetched
        to the host before the function returns.
        The output of step i - DEPTH is copied out after step i has been launched, so DEPTH steps are always queued on the GPU while the
        host copies and prepares (the engine keeps DEPTH + 1 sets of per-step inputs / outputs)."""
        for i in range(n):
            if exchange == "none":  # one library call per step: stage + launch + the view of the step DEPTH back (eftb_step)
                d = sets[first + i] if src is None else src[i % len(src)]
                if HOST_PROFILE is not None:
                    ta = time.perf_counter()
                view = eng.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=DEPTH if i >= DEPTH else -1, shape=(B, NL, NK),
                                out=(results[i] if keep else warm_out[i % len(warm_out)]) if OWN_OUT else None)
                if HOST_PROFILE is not None:
                    tb = time.perf_counter()
                if view is not None:
                    if keep:
                        if keeper is not None:
                            keeper.put(view, i - DEPTH)
                        if STEP_TIMES is not None:
                            STEP_TIMES.append(time.perf_counter())
                if HOST_PROFILE is not None and keep:
                    HOST_PROFILE[0] += tb - ta
                    HOST_PROFILE[1] += time.perf_counter() - tb
                continue
            stage_and_run(sets[first + i] if src is None else src[i % len(src)])
            if exchange == "rccl":
                eng.gather_plk(B, root=0)
                if i >= DEPTH and rank == 0:
                    block = eng.fetch_gathered(B, back=DEPTH, copy=False)
                    if keep:
                        keeper.put(block[rank], i - DEPTH)   # (the root's own slice, copied by the second host thread: 1.6 MB per step would hold the launching thread ~0.1 ms)
            elif exchange == "host-fallback":
                eng.sync()
                cp.gather_host(eng.get("PLK", (B, NL, NK)))
        # drain: the last DEPTH steps
        if exchange == "rccl":
            eng.sync()
            if rank == 0:
                for back in range(min(DEPTH, n) - 1, -1, -1):
                    block = eng.fetch_gathered(B, back=back, copy=False)
                    if keep:
                        keeper.put(block[rank], n - 1 - back)
                keeper.join()
        elif exchange == "none":
            if not os.environ.get("EFTB_BENCH_NO_FLUSH"):
                eng.flush()   # (the burst ends here: the last queued steps do not wait for a launch to fill)
            for back in range(min(DEPTH, n) - 1, -1, -1):  # (back = 0: the step launched last)
                take(n - 1 - back, back, keep)
            if keeper is not None:
                keeper.join()  # every kept P_l is in `results` before the function returns (inside the timed region)
            eng.sync()
        else:
            eng.sync()

    # Warm-up: the W steps asked for, then the SAME untimed loop again until WARM_MS of wall time have passed.  A GPU that has sat idle while the host
    # built tables runs its first ~10 ms of work below its sustained clocks (measured on the MI355X pool: 20 timed steps = 8 ms right behind 3 warm-up
    # steps 0.426-0.429 ms per step, behind 50 ms of the same loop 0.388-0.397, behind 500 ms 0.393; --warmup 20 alone 0.409): a chain that runs
    # for hours sees the latter.  EFTB_BENCH_PREWARM_MS=0 gives the bare W steps.
    eng.set_latency_mode(False)  # the pipelined loop keeps DEPTH steps queued: its first step is not a dependent sampler's step
    # `value` is the rate of P_l(k) evaluations: the engine's direct-P_l runs (EFTB_O_PLK_DIRECT: the bias contraction taken before the
    # resummation and the AP stage, with which it commutes) -- the templates-first rate of the same loop is the side key
    # templates_first_evaluations_per_s, and both loops' outputs are compared below.  EFTB_BENCH_DIRECT=0: templates first as `value`.
    eng.set_plk_direct(DIRECT)
    loop(0, W, keep=False)
    bare_elapsed = None
    if bare_sets and float(os.environ.get("EFTB_BENCH_PREWARM_MS", "50")) > 0:
        cp.barrier()
        tb0 = time.perf_counter()
        loop(0, K, keep=False, src=bare_sets)
        cp.barrier()
        bare_elapsed = cp.max(time.perf_counter() - tb0)
    # HIP events around every second launch of the dominant kernel inside the timed region, on the stream it runs on (every launch costs the loop
    # about 1.5 %: two more packets per step on the queue the resummation waits in) -- switched on before the warm-up extension, so that nothing
    # but the resets below stands between its last step and the timed region
    eng.time_kernels(1)  # the resummation kernel: the launch with the largest time of either kind of step (direct-P_l runs: the synthesis and the AP kernel are timed in an untimed repeat of the loop, below)
    eng.time_dominant(0 if os.environ.get("EFTB_BENCH_NO_EVENTS") else int(os.environ.get("EFTB_BENCH_EVENT_EVERY", "2")))
    import gc

    gc.collect()
    gc.disable()   # (a generation-2 collection inside a 2.5 ms timed region is a 20 % outlier; the loop allocates next to nothing)
    # The extension: the same untimed loop, CONTINUOUSLY, in chunks of WARM_CHUNK steps over a ring of draw sets until WARM_MS of wall time have passed.
    # (Second session of round 4: repeating the W-step loop -- mostly pipeline fill and drain at W = 5 -- left the GPU short of its sustained state: of
    # three 200-step passes run back to back the first took 89.8 us per step, the others 82.9 and 81.2.)
    warm_ms, warm_steps, WARM_CHUNK = float(os.environ.get("EFTB_BENCH_PREWARM_MS", "50")), W, int(os.environ.get("EFTB_BENCH_PREWARM_CHUNK", "200"))
    tw = time.perf_counter()
    # (every rank runs the same number of chunks: each step ends in a collective, so the decision is taken together)
    while W > 0 and cp.max(1.0 if (time.perf_counter() - tw) * 1e3 < warm_ms else 0.0) > 0.0:
        loop(0, WARM_CHUNK, keep=False, src=ring_sets)
        warm_steps += WARM_CHUNK
    for kind in range(3):
        eng.kernel_time(kind, reset=True)
    eng.submit_stats(enable=True, reset=True)   # host clocks of the staged steps (two steady_clock reads per call: ~50 ns)
    cp.barrier()
    t0 = time.perf_counter()
    loop(W, K, keep=True)
    cp.barrier()
    elapsed = cp.max(time.perf_counter() - t0)
    gc.enable()
    host_stats = eng.submit_stats(enable=False, reset=True)
    if HOST_PROFILE is not None:
        print(f"[bench] main thread, per timed step: eng.step {HOST_PROFILE[0] / K * 1e6:.1f} us (of which waiting for results {host_stats['wait_us_per_step']:.1f}, filling the staging "
              f"block {host_stats['fill_us_per_step']:.1f}), keeper.put {HOST_PROFILE[1] / K * 1e6:.1f} us; wall {elapsed / K * 1e6:.1f} us", file=sys.stderr)
    if STEP_TIMES:
        print("[bench] fetch-complete times since t0 (ms):", " ".join(f"{(t - t0) * 1e3:.3f}" for t in STEP_TIMES), file=sys.stderr)
    ktimes = [eng.kernel_time(kind, reset=True, cosmologies=True) for kind in range(3)]  # (resummation, synthesis, AP kernel): (ms, launches, cosmologies carried) inside the timed region
    dom_ms, dom_n = ktimes[0][:2]
    if DIRECT and not os.environ.get("EFTB_BENCH_NO_EVENTS"):
        # the two next-heaviest kernels of a direct-P_l step (`roofline_others`), bracketed in an UNTIMED repeat of the same loop: every bracketed launch
        # is two more packets on its queue, and three kernels' worth of them cost the timed loop 3 %
        eng.time_kernels(6)
        loop(W, K, keep=False)
        ktimes[1:] = [eng.kernel_time(kind, reset=True, cosmologies=True) for kind in (1, 2)]
    eng.time_dominant(False)
    eng.time_kernels(1)

    # ---- the timed loop's own outputs, checked: finite, and EVERY timed step bit-identical to the synchronous one-call path on the same draws
    steps_checked = 0
    if exchange == "none" or (exchange == "rccl" and rank == 0):
        assert np.all(np.isfinite(results)), "non-finite P_l(k) in the timed loop"
    for i in range(K):
        d = sets[W + i]
        sync_plk = eng.eval_batch(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], templates=False)
        if exchange == "none" or (exchange == "rccl" and rank == 0):
            assert np.array_equal(results[i], sync_plk), f"pipelined step {i} differs from the synchronous path"
            steps_checked += 1
    if exchange == "rccl" and rank == 0:  # every rank's block of the last exchange (still in place: nothing was exchanged after it)
        last = eng.fetch_gathered(B, back=0, copy=False)
        assert last.shape == (world, B, NL, NK) and np.all(np.isfinite(last)), "non-finite P_l(k) in the gathered block"
        assert np.array_equal(last[rank], results[K - 1]), "the root's own slice of the gathered block differs from its P_l"

    # ---- the same timed loop with the templates first (every rank takes part: the N > 1 loop exchanges); its P_l against the direct runs'
    tf_elapsed, tf_err, tf_dom_ms, tf_dom_n, tf_dom_nc = None, None, 0.0, 0, 0
    if DIRECT:
        direct_results = results.copy()
        eng.set_plk_direct(False)
        loop(0, W, keep=False)
        if W > 0 and warm_ms > 0:
            loop(0, WARM_CHUNK // 2, keep=False, src=ring_sets)   # (the templates-first loop's own warm-up: ~35 ms of it)
        eng.time_kernels(1)   # resum_mfma_kernel inside THIS loop: the in-pipeline time `roofline_templates_first` is priced with
        eng.time_dominant(0 if os.environ.get("EFTB_BENCH_NO_EVENTS") else int(os.environ.get("EFTB_BENCH_EVENT_EVERY", "2")))
        eng.kernel_time(0, reset=True)
        cp.barrier()
        t0 = time.perf_counter()
        loop(W, K, keep=True)
        cp.barrier()
        tf_elapsed = cp.max(time.perf_counter() - t0)
        tf_dom_ms, tf_dom_n, tf_dom_nc = eng.kernel_time(0, reset=True, cosmologies=True)
        eng.time_dominant(False)
        if exchange == "none" or (exchange == "rccl" and rank == 0):
            scale = np.max(np.abs(results), axis=-1, keepdims=True)
            tf_err = float(np.max(np.abs(direct_results - results) / scale))
            assert tf_err < 1e-8, f"direct-P_l runs differ from the templates-first runs of the same draws: {tf_err:.2e}"
        results[...] = direct_results
        eng.set_plk_direct(True)

    if rank == 0:
        extras = {}
        if bare_elapsed is not None:
            extras["bare_warmup_evaluations_per_s"], extras["bare_warmup_ms_per_step"] = B * world * K / bare_elapsed, bare_elapsed / K * 1e3
            extras["bare_warmup_note"] = (f"the same {K}-step loop over {K} other draw sets, timed right behind the {W} warm-up steps asked for and BEFORE the warm-up extension "
                                          "(warmup_note): what the first milliseconds after idle look like; `value` is the loop in the GPU's sustained state")
        if tf_elapsed is not None:
            extras["templates_first_evaluations_per_s"], extras["templates_first_ms_per_step"] = B * world * K / tf_elapsed, tf_elapsed / K * 1e3
            extras["templates_first_note"] = ("the same timed loop with EFTB_O_PLK_DIRECT off: the 24 templates per multipole go through resummation and AP and the "
                                              "bias contraction rides in the AP epilogue (the path `value` was measured on up to round 2)"
                                              + (f"; all {K} timed steps of the two paths agree to {tf_err:.1e} of each multipole's maximum" if tf_err is not None else ""))
        eng.set_latency_mode(True)
        if world == 1 and not force_comm:
            # (0) what a DEPENDENT sampler sees (reference likelihood.py:570-594 inside Model.logpost: step i + 1 needs step i's P_l): stage ->
            # run -> fetch the step just launched, nothing queued behind it.  New inputs every step, P_l back every step.
            chi = 0.0
            for phase, cnt in (("warm", 3), ("timed", K)):
                if phase == "timed":
                    eng.sync()
                    t1 = time.perf_counter()
                for i in range(cnt):
                    d = sets[(W + i) % len(sets)]
                    eng.stage_inputs(d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"])
                    eng.run_staged(mask, B)
                    out1 = eng.fetch_previous("PLK", (B, NL, NK), back=0, copy=False)   # P_l in page-locked host memory, consumed in place
                    chi += float(out1[0, 0, 7])                                           # (the sampler's use of it: here one element)
            dt1 = time.perf_counter() - t1
            extras["sync_step_evaluations_per_s"], extras["sync_step_ms"] = B * K / dt1, dt1 / K * 1e3
            extras["sync_step_note"] = ("stage -> run -> P_l of the step just launched, nothing queued behind it (the engine finds the GPU idle when the step is staged and "
                                        "runs it in latency mode: one queue, P_lin read from the staging block, P_l written to mapped host memory by the AP epilogue)")
            assert np.array_equal(out1, results[K - 1]), "dependent-sampler loop differs from the pipelined loop on the same draws"
        if not args.no_extras:
            # (1) the same kernels over inputs resident in HBM (what round 1 reported as `value`): K asynchronous runs of one batch
            d0 = sets[0]
            eng.load_inputs(d0["Pin"], d0["f"], d0["DA"], d0["H"], d0["bias"])
            for _ in range(3):
                eng.run(mask, B, sync=False)
            eng.sync()
            t1 = time.perf_counter()
            for _ in range(K):
                eng.run(mask, B, sync=False)
            eng.sync()
            extras["resident_evaluations_per_s"] = B * K / (time.perf_counter() - t1)
            extras["resident_note"] = (f"eftb_run over inputs that stay in HBM, one launch per {B} cosmologies, nothing fetched: no staging and no copy-out, but no "
                                       f"coalescing either -- the staged loop of `value` sends up to {COALESCE} queued steps as one launch, which is why it can be the faster of the two")
            # (2) host inputs in, the whole template block [B][3][24][512] (37.7 MB) back into page-locked memory, synchronous call
            pin = eng.pinned_empty((B, NL, 24, NK))
            eng.eval_batch(d0["Pin"], d0["f"], d0["DA"], d0["H"], out=pin)
            t1 = time.perf_counter()
            for i in range(4):
                d = sets[1 + i % (len(sets) - 1)]
                eng.eval_batch(d["Pin"], d["f"], d["DA"], d["H"], out=pin)
            extras["templates_back_evaluations_per_s"] = B * 4 / (time.perf_counter() - t1)
            # (2b) the batch size is a throughput knob: the same pipelined loop (new inputs staged and P_l fetched every step) at 256 per step
            if world == 1 and not force_comm and B == 128:
                try:
                    B2, n2 = 256, 20
                    eng2 = Engine(cfg, max_batch=B2, device=device, coalesce=max(1, COALESCE // 2))   # (as many cosmologies per launch as the loop of `value`)
                    eng2.set_latency_mode(False)
                    eng2.set_plk_direct(DIRECT)
                    sets2 = [draw_set(1000 + i, B2) for i in range(n2 + 8)]
                    out2 = eng2.pinned_empty((n2 + 8, B2, NL, NK))

                    def loop2(first, n, ring=0):
                        for i in range(n):
                            j = first + (i % ring if ring else i)
                            d = sets2[j]
                            eng2.step(mask, d["Pin"], d["f"], d["DA"], d["H"], bias=d["bias"], back=DEPTH if i >= DEPTH else -1, shape=(B2, NL, NK), out=out2[j])
                        eng2.flush()
                        for back in range(min(DEPTH, n) - 1, -1, -1):
                            eng2.fetch_previous("PLK", (B2, NL, NK), back=back, copy=False)
                        eng2.sync()

                    tw2 = time.perf_counter()
                    while (time.perf_counter() - tw2) * 1e3 < warm_ms:   # (the same continuous warm-up as the headline loop, over eight draw sets of its own)
                        loop2(n2, 100, ring=8)
                    t1 = time.perf_counter()
                    loop2(0, n2)
                    extras["value_at_batch_256"] = B2 * n2 / (time.perf_counter() - t1)
                    eng2.close()
                except Exception as exc:  # pragma: no cover
                    extras["value_at_batch_256_error"] = repr(exc)
            # (2c) BASELINE cfg 3 (three tracers per likelihood point, production windows, marginalised ln P) and the single-GPU half of cfg 5
            # (Nk = 2048, window + binning, two tracers), each checked inside its loop against the reference's own outputs (tests/golden)
            if world == 1 and not force_comm:
                try:  # (a fresh child process: its own engines, queues and thread pools -- this process idles meanwhile)
                    import subprocess

                    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cfg_rates.py"), "--json", str(device)], capture_output=True, text=True, timeout=600)
                    if res.returncode:
                        raise RuntimeError(res.stderr[-1500:])
                    extras.update(json.loads(res.stdout.strip().splitlines()[-1]))
                except Exception as exc:  # pragma: no cover
                    extras["cfg_rates_error"] = repr(exc)
            # (3) the drop-in path as theory.py drives it (one cosmology per call through the pybird mirror classes, production grid)
            try:
                from tools.dropin_probe import dropin_latency_ms

                extras.update(dropin_latency_ms(repeats=20))
            except Exception as exc:  # pragma: no cover
                extras["dropin_error"] = repr(exc)
        # per-kernel times, HIP events around back-to-back launches on the engine stream (stage by stage: the templates-first kernels)
        eng.set_plk_direct(False)
        d0 = sets[W]
        eng.load_inputs(d0["Pin"], d0["f"], d0["DA"], d0["H"], d0["bias"])
        eng.run(mask, B)
        reps = 10
        ms_resum = eng.run_timed(L.K_RESUM, B, reps)   # resum_mfma_kernel alone: the dominant kernel of the step
        ms_p22 = eng.run_timed(L.K_P22, B, reps)       # makeP22 path: anti-diagonal sums + rows + synthesis + expansion
        ms_c22 = eng.run_timed(L.K_C22, B, reps)
        stages = {n: eng.run_timed(m, B, 3) for n, m in (("prep", L.S_PREP), ("loops", L.S_LOOPS), ("cf", L.S_CF),
                                                        ("regroup", L.S_REGROUP), ("resum", L.S_RESUM), ("ap", L.S_AP), ("reduce", L.S_REDUCE))}
        # SURVEY 8(f) rank 1, reported beside the headline: the marginalised log-posterior stage on the same batch (synthetic data
        # vector: l = 0, 2, 4 at every 8th k in [0.02, 0.2], 7 marginalised parameters) -- one float per walker leaves the GPU
        from eftpipe_amd.marginal import MarginalLikelihood, gaussian_rows
        kk = cfg.k
        sel = np.nonzero((kk >= 0.02) & (kk <= 0.2))[0][::8]
        index = np.concatenate([l * NK + sel for l in range(NL)]).astype(np.int32)
        rng = np.random.default_rng(7)
        rows = np.stack([gaussian_rows(float(f), (BS[0], BS[1], BS[3]), None, 0.7, 0.25, 4.5e-5) for f in d0["f"]])
        eng.run(eng.full_mask(), B)
        model = np.einsum("r,lrx->lx", rows[0][0], eng.get("TEMPL", (B, NL, 24, NK))[0]).reshape(-1)[index]
        sig = 0.05 * np.abs(model) + 10.0
        like = MarginalLikelihood(eng, index, model + sig * rng.normal(size=index.size), np.diag(1.0 / sig**2), np.zeros(7), np.full(7, 2.0))
        like.logp(rows)
        stages["logp_marginalised"] = eng.run_timed(L.S_LOGP, B, 3)

        # ---- roofline of the dominant kernel: EXECUTED FP64 work / live-measured time / FP64 matrix peak
        exe_flops, exe_mfma_flops, per_trip = executed_flops_per_launch(B)
        ms_alone = ms_resum
        # average over the launches of the timed region (beside the look-ahead / back-half kernels).  In a direct-P_l bench the timed region's kind-0
        # timer bracketed resum_plk_kernel, not this kernel: its own in-pipeline time comes from the templates-first loop above
        if DIRECT:
            dom_tf_ms, dom_tf_n, dom_tf_nc = tf_dom_ms, tf_dom_n, tf_dom_nc
        else:
            dom_tf_ms, dom_tf_n, dom_tf_nc = dom_ms, dom_n, ktimes[0][2]
        ms_resum = dom_tf_ms / dom_tf_n if dom_tf_n else ms_alone
        # (coalesced staged steps leave as one launch: the in-pipeline launches carried dom_tf_nc / dom_tf_n cosmologies each, the stand-alone one B)
        bl_tf = dom_tf_nc / dom_tf_n if dom_tf_n else B
        exe_flops_alone, exe_flops, exe_mfma_flops = exe_flops, exe_flops * bl_tf / B, exe_mfma_flops * bl_tf / B
        achieved = exe_flops / (ms_resum * 1e-3) / 1e12
        try:
            measured_peak = mfma_f64_peak(device)
        except Exception:
            measured_peak = None
        traffic, traffic_src = None, None
        for name in ("r03_pmc_dominant.json", "r02_pmc_dominant.json", "r01_pmc_dominant.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                with open(pmc) as fh:
                    pj = json.load(fh)
                    traffic, traffic_src = pj.get("templates_first", pj).get("hbm_bytes_per_launch"), "profiles/" + name
                break
        # ALGORITHMIC flops of SURVEY.md 8(d), as the reference computes the stage (side keys: the engine executes an algebraically
        # reduced form, so these are NOT what the hardware does):  F_IRn = 8 Na [Nl 14 2NIR] 193 (Nk - 7),  F_P22 = 8 28 Nk 257^2
        NIR, NA = 16, 3
        alg_resum = 8.0 * NA * (NL * 14 * 2 * NIR) * 193 * (NK - NKLOW) * B
        alg_p22 = 8.0 * 28 * NK * NPOW**2 * B
        roofline = {
            "bound": "mfma", "kernel": "resum_mfma_kernel<NNLO> (Resum.Ps on v_mfma_f64_16x16x4_f64 + FP64 VALU, which share the DP pipe)",
            "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src,
            "ms_per_launch": ms_resum, "launches_timed": dom_tf_n,
            "ms_per_launch_note": "HIP events on the kernel's own stream around every second of its launches inside the timed templates-first loop, where the look-ahead and "
                                  "back-half kernels of the neighbouring steps share the CUs (and the FP64 pipe) with it" + ("" if dom_tf_n else " -- NOT AVAILABLE in this run: the stand-alone time is used"),
            "cosmologies_per_launch": bl_tf,
            "ms_per_launch_alone": ms_alone, "frac_alone": exe_flops_alone / (ms_alone * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "executed_flops_per_launch": exe_flops, "executed_mfma_flops_per_launch": exe_mfma_flops,
            "mfma_only_frac": exe_mfma_flops / (ms_resum * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "instructions_per_wave_trip": per_trip, "counts_source": "eftpipe_amd/csrc/isa_counts.json (tools/isa_counts.py, from the compiled gfx950 assembly)",
            "measured_mfma_f64_issue_peak_tflops": measured_peak,
            "note": "achieved = FP64 flops the kernel issues (MFMA + vector FMA/MUL/ADD, padded lanes included) / HIP-event time; the "
                    "reference's algorithmic count of the same stage is a side key because the kernel executes an algebraically reduced form",
            "reference_algorithmic_flops_per_launch": alg_resum, "reference_algorithmic_tflops": alg_resum / (ms_resum * 1e-3) / 1e12,
            "p22_path_ms": ms_p22, "p22_reference_algorithmic_flops_per_launch": alg_p22, "c22_path_ms": ms_c22, "stage_ms": stages,
            "stage_ms_note": "stages timed alone on the main stream (no overlap between consecutive steps)",
        }
        if DIRECT:
            roofline = direct_rooflines(eng, cfg, B, ktimes, sets[W], roofline, FP64_MFMA_PEAK_TFLOPS, measured_peak)
            # the same three kernels with nothing beside them: synchronous direct-P_l runs of one batch, every launch bracketed
            eng.set_plk_direct(True)
            eng.time_kernels(7)
            eng.time_dominant(1)
            for kind in range(3):
                eng.kernel_time(kind, reset=True)
            for _ in range(6):
                eng.run(mask, B, sync=True)
            alone = [eng.kernel_time(kind, reset=True, cosmologies=True) for kind in range(3)]
            eng.time_dominant(False)
            eng.time_kernels(1)
            tag = {"resum_plk": 0, "synth_kernel": 1, "ap_plk_fused": 2}
            for ent in [roofline] + roofline.get("roofline_others", []):
                for name, kind in tag.items():
                    if ent["kernel"].startswith(name) and alone[kind][1]:
                        ms_a = alone[kind][0] / alone[kind][1]
                        fl = ent.get("executed_flops_per_launch", ent.get("algorithmic_flops_per_launch")) * B / ent["cosmologies_per_launch"]
                        ent["ms_per_launch_alone"], ent["frac_alone"] = ms_a, fl / (ms_a * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
                        ent["alone_note"] = f"synchronous direct-P_l runs of {B} cosmologies, nothing of another step beside the kernel (HIP events on its stream)" 
            # the step as a whole: executed FP64 flops of all its launches over the step time (a direct step has no dominant kernel)
            with open(os.path.join(ROOT, "eftpipe_amd", "csrc", "isa_counts.json")) as fh:
                per_kernel = direct_step_flops(eng, B, json.load(fh), ap_slots_per_k(sets[W], np.asarray(cfg.k), cfg.DA_AP, cfg.H_AP))
            step_flops = sum(per_kernel.values())
            ms_step = elapsed / K * 1e3
            res = extras.get("resident_evaluations_per_s")
            roofline["roofline_step"] = {
                "bound": "valu_f64 + mfma (one DP pipe)", "executed_flops_per_step": step_flops, "flops_per_kernel": per_kernel,
                "achieved": step_flops / (ms_step * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": step_flops / (ms_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms_per_step": ms_step,
                "frac_resident_inputs": (step_flops / (B / res) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if res else None,
                "note": "sum over the launches of one direct-P_l step of the FP64 flops they execute (compiled-loop counts for the resummation, 2 M N K for the matrix-core "
                        "products, sums' shapes for the rest) / ms_per_step of the timed loop / FP64 peak (vector and matrix FP64 instructions share one arithmetic budget on "
                        "gfx950: tools/probe/mixed_f64_probe.hip); frac_resident_inputs: the same over the step time of the resident loop"}

        def check_fracs(node, path="roofline"):
            # no utilisation above 1 leaves this program: such a number is a pricing error, not a measurement
            if isinstance(node, dict):
                for k_, v in node.items():
                    if k_.startswith("frac") or k_ == "mfma_only_frac":
                        assert v is None or 0.0 <= v <= 1.0, f"{path}.{k_} = {v}: outside [0, 1]"
                    check_fracs(v, path + "." + k_)
            elif isinstance(node, list):
                for i_, v in enumerate(node):
                    check_fracs(v, f"{path}[{i_}]")

        check_fracs(roofline)
        value = B * world * K / elapsed
        valid = exchange in ("none", "rccl") and not shared_device
        out = {
            "metric": "theory P_l(k) evaluations/sec (single tracer, Nk=512, l=0,2,4)",
            "value": value, "unit": "evaluations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (SYNTH-PLIN v1; a new draw set of `batch_per_gpu` cosmologies every step, seeds 12345 + 7919 step + 104729 rank)",
            "config": {"workload": "cfg2: single-tracer LRG z=0.7, Nl=3 (l=0,2,4), Nk=512, IR-resum + AP, P_l via west-coast bias contraction; "
                                   "H2D+D2H inclusive: every step stages new inputs from host memory and its P_l is fetched to host memory inside the timed region; "
                                   f"`value` pipelines INDEPENDENT batches at depth {DEPTH} (step i's P_l is taken after step i + {DEPTH} has been handed in; steps still queued when "
                                   f"the library's submission thread reaches them leave as one launch of up to {COALESCE} steps -- Engine(coalesce={COALESCE}) --, each fetched by itself); the rate a "
                                   "sampler sees whose next step depends on this step's P_l is the side key sync_step_evaluations_per_s"
                                   + ("; every step's P_l is delivered into the sampler's own page-locked results array by the copy-out behind the step (eftb_set_step_output), "
                                      "where it is compared with the synchronous path afterwards" if OWN_OUT else "")
                                   + ("; `value` is measured on direct-P_l runs (EFTB_O_PLK_DIRECT: the bias contraction is taken before the synthesis of the loop pieces, the "
                                      "resummation and the AP stage, with which it commutes -- same P_l, no template block); the templates-first rate of the same loop (rounds 1-2) is "
                                      "the side key templates_first_evaluations_per_s, and every timed step of the two is compared" if DIRECT else ""),
                       "batch_per_gpu": B,
                       "parallelism": (f"batch-sharded x{world}, per-step gather of P_l to rank 0 via {exchange}, rank 0 receives every step's gathered block in page-locked host memory"
                                       if world > 1 or force_comm else "single GPU")},
            "valid": bool(valid), "timed_steps_checked_against_sync_path": steps_checked,
            "host_us_per_step": host_stats,
            "warmup_steps_run": warm_steps,
            "warmup_note": f"{W} untimed pipelined steps as asked, then the same untimed loop run continuously (chunks of {WARM_CHUNK} steps over a ring of {NRING} further draw sets) until "
                           f"{warm_ms:.0f} ms had passed ({warm_steps} steps in all): the timed region (a few ms) otherwise runs in the state of a GPU that has just left idle -- of three "
                           "200-step passes back to back the first takes 10 % longer per step than the others; EFTB_BENCH_PREWARM_MS=0 switches the extension off",
            "roofline": roofline,
        }
        if not valid:
            out["invalid_reason"] = ("P_l was gathered over host sockets, not RCCL" if exchange == "host-fallback" else "ranks shared one GPU") + " -- a rehearsal of the launch path, not a measurement"
        out.update(extras)
        if not args.no_cpu_baseline:
            # two cosmologies OF THE TIMED SETS (first of the first timed step, last of the last timed step) lead the CPU sample; the oracle's
            # templates of those two are compared with what the timed loop itself fetched
            pick = lambda st, w: dict(kin=sets[W + st]["kin"], Pin=sets[W + st]["Pin"][w], f=sets[W + st]["f"][w], DA=sets[W + st]["DA"][w],
                                      H=sets[W + st]["H"][w], step=st, w=w)
            picks = [pick(0, 0), pick(K - 1, B - 1)] + [pick((3 * j) % K, (37 * j) % B) for j in range(1, 7)]
            out["cpu_baseline"], templ01 = cpu_baseline(picks)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
            if not args.no_extras:
                try:
                    from tools.dropin_probe import oracle_ms

                    out.update(oracle_ms())   # the CPU port on the drop-in configuration (50-point grid, window, binning), beside dropin_ms_per_eval
                except Exception as exc:  # pragma: no cover
                    out["dropin_cpu_port_error"] = repr(exc)
            # parity of the TIMED outputs against the oracle, pointwise where |P_l| is not tiny
            worst = 0.0
            for c, templ in zip(picks[:2], templ01):
                got = results[c["step"], c["w"]]
                want = np.einsum("r,lrx->lx", sets[W + c["step"]]["bias"][c["w"]], templ)
                big = np.abs(want) > 1e-3 * np.max(np.abs(want), axis=-1, keepdims=True)
                worst = max(worst, float(np.max(np.abs(got - want)[big] / np.abs(want)[big])))
            out["max_rel_err_vs_oracle"] = worst
            out["parity_note"] = (f"all {steps_checked} timed steps bit-identical to the synchronous eftb_eval_batch path; P_l of (step 0, cosmology 0) and "
                                  f"(step {K - 1}, cosmology {B - 1}) as fetched inside the timed loop against the oracle: max relative error {worst:.2e} (bar 1e-6)")
            assert worst < 1e-6, worst
        print(json.dumps(out))
    cp.barrier()
    eng.close()
    cp.close()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set as torch.distributed.run sets them), let rank 0's stdout (the JSON line) through, and return the worst child status."""
    import socket
    import subprocess
    import uuid

    with socket.socket() as sk:  # a free port for the rendezvous name (the control plane is a Unix socket named from it unless EFTB_CP_TCP_PORT is set)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_id = uuid.uuid4().hex[:12]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID=run_id)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst = 0
    for r, pr in enumerate(procs):
        rc = pr.wait()
        if rc:
            print(f"[bench] rank {r} exited with status {rc}", file=sys.stderr)
            worst = max(worst, rc if rc > 0 else 128 - rc)
    return worst


def _device_count():
    """Visible HIP devices, from the runtime's own counter (hipGetDeviceCount: this does initialise the HIP runtime in this process, which is
    about to create its engine anyway; no SMI tool is needed)."""
    import ctypes

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) == 0:
            return n.value
    except OSError:
        pass
    return 0


if __name__ == "__main__":
    main()
