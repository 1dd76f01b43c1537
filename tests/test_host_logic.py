"""Host logic on the CPU: Common mirror, bias vectors, wave plan, sharding, socket control plane."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr


def test_common_mirror_matches_reference_tables(golden):
    from eftpipe_amd import pybird

    g = golden("tables")
    for Nl in (2, 3):
        co = pybird.Common(Nl=Nl, kmax=0.3)
        for n in ("l11", "lct", "l22", "l13"):
            assert np.array_equal(getattr(co, n), g[f"{n}_Nl{Nl}"])
        assert co.Nk == 50 and co.Ns == 80 and co.Nklow == 7 and co.Nloop == 12
    assert np.array_equal(pybird.Common(Nl=3).k, golden("caseC")["k"])
    with pytest.raises(ValueError):
        pybird.Common(Nl=2, No=3)
    co = pybird.Common(Nl=2, optiresum=True)
    assert co.Ns == 52 and co.s[0] == 70.0 and co.s[-1] == 197.5
    from oracle import tables as OT

    con = pybird.Common(Nl=3, with_NNLO=True)
    assert con.with_NNLO and np.array_equal(con.lctNNLO, OT.mu_weights(3)["lctNNLO"]) and con.NctNNLO == 3
    # reference tests/test_pybird.py:5-11
    assert np.isclose(pybird.Hubble(0.2, 1.0), 1.549193338482967, atol=0.0)
    assert np.isclose(pybird.DAfunc(0.2, 1.0), 0.4117451980802465, atol=0.0)


def test_bias_vectors_match_oracle(golden):
    from eftpipe_amd.parambasis import bias_row, bias_vectors
    from oracle import OracleConfig, OracleEngine

    g = golden("caseA")
    orc = OracleEngine(OracleConfig(Nl=2, ndA=4.5e-5, kmB=0.6, krB=0.3, ndB=2.3e-4))
    f = 0.77
    want = orc.bias_vectors(f, list(g["bsA"]), list(g["bsB"]), tuple(g["es"]))  # b11, bloop, bct, bst
    got = bias_vectors(f, list(g["bsA"]), list(g["bsB"]), tuple(g["es"]), kmA=0.7, krA=0.25, ndA=4.5e-5, kmB=0.6, krB=0.3, ndB=2.3e-4)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[2])
    assert np.array_equal(got[2], want[1]) and np.array_equal(got[3], want[3])
    assert bias_row(f, list(g["bsA"])).shape == (24,)


def test_antidiagonal_table_covers_every_pair_once():
    """AD[c, j', t] holds the symmetrised weight of pair (j' + t, 256 - t): every unordered pair with n + m >= 256 once."""
    from eftpipe_amd.tables import antidiagonal_tables

    rng = np.random.default_rng(0)
    M = rng.normal(size=(1, 257, 257)) + 1j * rng.normal(size=(1, 257, 257))
    AD = antidiagonal_tables(M, None)
    seen = set()
    for jp in range(257):
        cnt = ((256 + jp) >> 1) - jp + 1
        assert np.all(AD[0, jp, cnt:] == 0)
        for t in range(cnt):
            n, m = jp + t, 256 - t
            assert n <= m and n + m == 256 + jp
            want = M[0, n, m] + M[0, m, n] if n < m else M[0, n, n]
            assert AD[0, jp, t] == want
            seen.add((n, m))
    assert len(seen) == sum(1 for n in range(257) for m in range(n, 257) if n + m >= 256)


def test_shard_bounds():
    from eftpipe_amd.dist import shard_bounds

    for total in (1, 7, 128, 1024, 1000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, b = shard_bounds(total, world, r)
                cover += list(range(a, b))
                assert b - a in (total // world, total // world + 1)
            assert cover == list(range(total))


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
from eftpipe_amd import dist
cp = dist.ControlPlane()
a, b = dist.shard_bounds(10, cp.world, cp.rank)
local = np.arange(a, b, dtype=float)[:, None] * np.ones((1, 3))
payload = cp.broadcast_bytes(b"x" * 128 if cp.rank == 0 else None)
assert payload == b"x" * 128
tmax = cp.max(1.0 + cp.rank)
assert tmax == float(cp.world)
g = cp.gather_host(local)
cp.barrier()
if cp.rank == 0:
    assert g.shape == (cp.world, 5, 3) and np.array_equal(g.reshape(-1, 3)[:, 0], np.arange(10.0))
    print("GATHER_OK")
cp.close()
"""


def test_control_plane_world_size_2_under_torchrun(tmp_path):
    """The N > 1 control path (rendezvous bytes, barrier, max over ranks, host gather) launched the way the driver launches bench.py:
    `python -m torch.distributed.run` only provides RANK / WORLD_SIZE / MASTER_*; the control plane itself is socket-only."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT + "assert 'torch' not in sys.modules, 'the control plane must not import torch'\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


SHARD_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
from eftpipe_amd import dist, synth
cp = dist.ControlPlane()
total = 1024
draws = synth.draw_batch(total, z=0.7, seed=12345)              # every rank derives the same 1 024 draws ...
a, b = dist.shard_bounds(total, cp.world, cp.rank)              # ... and evaluates its contiguous share
stand_in = lambda Pin, f: np.stack([Pin[:, 100] * f, Pin[:, 150] + f, np.arange(Pin.shape[0]) * 0.0 + f], axis=1)   # [n, 3] per-draw result
mine = stand_in(draws["Pin"][a:b], draws["f"][a:b])
parts = cp.gather_ragged(mine)
tmax = cp.max(float(cp.rank))
cp.barrier()
if cp.rank == 0:
    assert tmax == cp.world - 1
    got = dist.reassemble(parts)
    want = stand_in(draws["Pin"], draws["f"])
    assert [p.shape[0] for p in parts] == [dist.shard_bounds(total, cp.world, r)[1] - dist.shard_bounds(total, cp.world, r)[0] for r in range(cp.world)]
    assert got.shape == (total, 3) and np.array_equal(got, want)
    print("REASSEMBLED_OK", cp.world)
else:
    assert parts is None
assert "torch" not in sys.modules
cp.close()
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_shards_reassemble_in_rank_order(tmp_path, world):
    """BASELINE cfg 4 on the CPU: 1 024 draws sharded over `world` ranks (plain processes, no launcher, no torch), each rank's block
    gathered to rank 0 and reassembled in draw order."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29540 + world),
                   OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        env.pop("TORCHELASTIC_RUN_ID", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    assert f"REASSEMBLED_OK {world}" in outs[0][0]


def test_control_plane_tcp_transport(tmp_path):
    """EFTB_CP_TCP_PORT switches the hub to TCP on MASTER_ADDR (same protocol)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", EFTB_CP_TCP_PORT="29562",
                   EFTB_CP_TOKEN="s3cret-of-this-launch")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    assert "GATHER_OK" in outs[0][0]


def test_control_plane_tcp_needs_token_and_loopback(monkeypatch):
    from eftpipe_amd import dist

    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("EFTB_CP_TCP_PORT", "29563")
    monkeypatch.delenv("EFTB_CP_TOKEN", raising=False)
    with pytest.raises(RuntimeError, match="EFTB_CP_TOKEN"):
        dist.ControlPlane()
    monkeypatch.setenv("EFTB_CP_TOKEN", "t")
    monkeypatch.setenv("MASTER_ADDR", "192.0.2.1")   # not loopback: refused unless EFTB_CP_TCP_ANY=1
    with pytest.raises(RuntimeError, match="loopback"):
        dist.ControlPlane()


@pytest.mark.parametrize("transport", ["unix", "tcp"])
def test_control_plane_rejects_foreign_connections(tmp_path, transport):
    """nothing that arrives on the hub socket is unpickled, and a connection becomes a peer only after the handshake.
    A foreign client that (a) sends a pickle, (b) claims a rank with the wrong token / a bad tag, (c) says nothing, is dropped; the real
    rank 1 then joins and the collectives run.  On the Unix socket the uid check is exercised through a patched SO_PEERCRED reader."""
    import pickle
    import socket
    import struct
    import threading
    import time

    from eftpipe_amd import dist

    port = 29570 if transport == "unix" else 29572
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EFTB_CP_TOKEN="right-token")
    env.pop("TORCHELASTIC_RUN_ID", None)
    env.pop("TORCHELASTIC_RESTART_COUNT", None)
    if transport == "tcp":
        env["EFTB_CP_TCP_PORT"] = str(port + 1)
        family, target = socket.AF_INET, ("127.0.0.1", port + 1)
    else:
        family, target = socket.AF_UNIX, "\0eftb-cp-127.0.0.1-%d--0" % port
    script = tmp_path / "hub.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\nfrom eftpipe_amd import dist\ncp = dist.ControlPlane()\n"
                      "print('REJECTED', cp.rejected, flush=True)\nassert cp.max(1.0) == 2.0\ncp.barrier()\ncp.close()\nprint('HUB_OK')\n" % ROOT)
    hub = subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def connect():
        t0 = time.monotonic()
        while True:
            s = socket.socket(family, socket.SOCK_STREAM)
            try:
                s.connect(target)
                s.settimeout(30)
                return s
            except (ConnectionRefusedError, FileNotFoundError):
                s.close()
                assert time.monotonic() - t0 < 60
                time.sleep(0.05)

    def dropped(s):
        try:
            return s.recv(64) == b""
        except OSError:
            return True

    # (a) a pickle instead of the handshake message
    s = connect()
    nonce = s.recv(16)
    assert len(nonce) == 16
    s.sendall(pickle.dumps({"rank": 1}) + b"\0" * 64)
    assert dropped(s)
    s.close()
    # (b) the right message layout with the wrong token
    s = connect()
    nonce = s.recv(16)
    s.sendall(struct.pack("<I", 1) + b"n" * 16 + dist._tag(b"wrong-token", nonce, 1))
    assert dropped(s)
    s.close()
    # (c) rank out of range with a valid tag
    s = connect()
    nonce = s.recv(16)
    s.sendall(struct.pack("<I", 7) + b"n" * 16 + dist._tag(b"right-token", nonce, 7))
    assert dropped(s)
    s.close()
    # the real rank 1, in this process
    for k, v in dict(env, RANK="1", LOCAL_RANK="1").items():
        os.environ[k] = v
    try:
        cp = dist.ControlPlane()
        assert cp.max(2.0) == 2.0
        cp.barrier()
        cp.close()
    finally:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "EFTB_CP_TOKEN", "EFTB_CP_TCP_PORT"):
            os.environ.pop(k, None)
    out, err = hub.communicate(timeout=120)
    assert hub.returncode == 0 and "HUB_OK" in out and "REJECTED 3" in out, out + err[-2000:]


def test_control_plane_unix_socket_checks_the_peer_uid(monkeypatch):
    """_admit drops a Unix-socket connection whose SO_PEERCRED uid is not ours (patched reader: the suite runs as one user)."""
    import socket

    from eftpipe_amd import dist

    cp = dist.ControlPlane.__new__(dist.ControlPlane)
    cp.world = 2
    a, b = socket.socketpair()
    assert dist._peer_uid(a) == os.getuid()
    monkeypatch.setattr(dist, "_peer_uid", lambda sock: os.getuid() + 1)
    assert cp._admit(a, True, b"", {}) is None
    a.close()
    b.close()


@pytest.mark.parametrize("Nl", [2, 3])
def test_pyegg_cache_layout_matches_reference(golden, Nl):
    """`tables.loop_matrices` reproduces the reference's loop-matrix cache (file name, keys, shapes,
    dtypes, values at sampled positions) -- so caches written by either side are interchangeable."""
    from eftpipe_amd.tables import PYEGG_KEYS, loop_matrices, pyegg_path

    g = golden("pyegg")
    assert pyegg_path("", 256, Nl) == str(g[f"name_Nl{Nl}"])
    assert list(g[f"keys_Nl{Nl}"]) == list(PYEGG_KEYS)
    m = loop_matrices(Nl)
    for k in PYEGG_KEYS:
        assert m[k].shape == tuple(g[f"shape_{k}_Nl{Nl}"]) and str(m[k].dtype) == str(g[f"dtype_{k}_Nl{Nl}"]), k
        got, want = m[k].reshape(-1)[g[f"idx_{k}_Nl{Nl}"]], g[f"val_{k}_Nl{Nl}"]
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-11, k


def test_tables_from_loop_cache_equal_recomputed():
    """Tables built from a pyegg-style cache are bit-identical to the recomputed ones; a cache for another FFTLog setup is refused."""
    from eftpipe_amd.tables import EngineConfig, build_tables, loop_matrices

    cfg = EngineConfig(Nl=2, with_resum=True)
    mats = loop_matrices(2)
    a, b = build_tables(cfg), build_tables(cfg, {k: mats[k] for k in ("Pow", "M22", "M13", "Mcf11", "Mcfct")})
    for k in ("ad", "linvec", "comb22", "comb13"):
        assert np.array_equal(a[k], b[k]), k
    bad = dict(mats, Pow=mats["Pow"] + 0.1)
    with pytest.raises(ValueError, match="different FFTLog"):
        build_tables(cfg, bad)


def test_plk_interpolator_operator(golden):
    """PlkInterpolator (theory.py:75-106) as a linear operator == the reference's scipy.interpolate.interp1d
    call (scipy 1.15 make_interp_spline, not-a-knot), inside the grid and extrapolated beyond both ends."""
    from eftpipe_amd.tables import interp_operator
    from eftpipe_amd.transformer import PlkInterpolator
    from oracle.engine import plk_interpolate

    g = golden("caseD")
    k, plk = g["k"], g["plk_auto"]
    kout = np.concatenate([[5e-4, 0.0123], np.arange(0.02, 0.3, 0.0137), [0.31]])
    want = plk_interpolate([0, 2, 4], k, plk, [0, 2, 4], kout)
    got = plk @ interp_operator(k, kout).T
    assert relerr(got, want) < 1e-11
    fn = PlkInterpolator([0, 2, 4], k, plk)
    assert relerr(fn([0, 2, 4], kout), want) < 1e-11
    assert relerr(fn(2, kout)[None], want[1][None]) < 1e-11 and fn(4, 0.1).shape == ()
    with pytest.raises(ValueError, match="not in"):
        fn(6, kout)


def test_parameter_bases_mirror_reference(golden):
    """WestCoastBasis / EastCoastBasis / find_param_basis (reference parambasis.py:166-466) on reference-generated fixtures."""
    from types import SimpleNamespace

    from eftpipe_amd import parambasis as P
    from eftpipe_amd import pybird

    c = golden("caseC")
    templ = {n: c["binned_" + n] for n in ("P11l", "Pctl", "Ploopl", "Pstl", "Picc")}
    # east coast
    g = golden("east")
    basis = P.find_param_basis("eastcoast")(prefix="")
    assert basis.get_name() == basis.counterform() == "eastcoast"
    assert basis.gaussian_params() == list(g["gaussian_params"]) and basis.non_gaussian_params() == list(g["non_gaussian_params"])
    with pytest.raises(NotImplementedError):
        P.EastCoastBasis(prefix="X_", cross_prefix=["A_", "B_"])
    kmA, krA, ndA = g["co"]
    co = pybird.Common(Nl=3, kmA=kmA, krA=krA, ndA=ndA, counterform="eastcoast")
    bird = SimpleNamespace(f=float(g["f"]), co=co, **templ)
    full = dict(zip(g["full_names"], g["full_values"]))
    comp = basis.reduce_Plk(bird, full)
    for n in ("Plin", "Ploop", "Pct", "Pst"):
        assert relerr(getattr(comp, n), g[n]) < 1e-14, n
    assert relerr(comp.sum(), g["plk"]) < 1e-14
    ng = {p: full[p] for p in ("b1", "b2", "bG2")}
    table = basis.reduce_Plk_gaussian_table(bird, ng)
    assert list(table) == list(g["names"])
    assert relerr(np.stack(list(table.values())).reshape(7, -1), g["table"].reshape(7, -1)) < 1e-14
    assert list(basis.reduce_Plk_gaussian_table(bird, ng, requires={"c2", "a2"})) == ["c2", "a2"]
    T = np.concatenate([templ[n] for n in ("P11l", "Pctl", "Ploopl", "Pstl")], axis=1)
    row = basis.bias_row(bird.f, full, kmA=kmA, krA=krA, ndA=ndA)
    assert relerr(np.einsum("b,lbx->lx", row, T), g["plk"]) < 1e-14  # what the device reduce contracts
    rows = basis.gaussian_rows(bird.f, ng, kmA=kmA, krA=krA, ndA=ndA)
    assert relerr(np.einsum("b,lbx->lx", rows[0], T), g["PNGl"]) < 1e-14
    with pytest.raises(ValueError):
        pybird.Common(Nl=3, counterform="southcoast")
    # west coast, auto and cross
    m = golden("marg")
    for tag, basis in (("auto", P.find_param_basis("westcoast")(prefix="")), ("cross", P.WestCoastBasis(prefix="X_", cross_prefix=["A_", "B_"]))):
        kmA, krA, ndA, kmB, krB, ndB = m[tag + "_co"]
        bird = SimpleNamespace(f=float(m["f"]), co=pybird.Common(Nl=3, kmA=kmA, krA=krA, ndA=ndA, kmB=kmB, krB=krB, ndB=ndB), **templ)
        ng = dict(zip(m[tag + "_ng_names"], m[tag + "_ng_values"]))
        assert list(ng) == basis.non_gaussian_params() and basis.get_name() == "westcoast"
        table = basis.reduce_Plk_gaussian_table(bird, ng)
        assert list(table) == list(m[tag + "_names"])
        assert relerr(np.stack(list(table.values())).reshape(len(table), -1), m[tag + "_table"].reshape(len(table), -1)) < 1e-14
        assert relerr(basis.reduce_Plk(bird, ng).sum(), m[tag + "_PNGl"]) < 1e-14


def test_fiber_operator_matches_reference(golden):
    """tables.fiber_operator (the dense form of FiberCollision.dPcorr, reference pybird.py:1703-1757) against the reference-generated
    fixture, the oracle's loop form on a non-native grid with kout != kPS, and its place in the folded pipeline operator."""
    from eftpipe_amd import tables as TB
    from oracle import fiber as F

    g, c = golden("fiber"), golden("caseC")
    k, fs, Dfc, kt = c["k"], float(g["fs"]), float(g["Dfc"]), float(g["ktrust"])
    Fm = TB.fiber_operator(k, 3, fs, Dfc, kt)
    for n in ("P11l", "Pctl", "Ploopl", "Pstl"):
        got = c["window_" + n] + np.einsum("alxk,lnk->anx", Fm, c["window_" + n])
        assert relerr(got, g["fiber_st_" + n]) < 1e-12, n
    rng = np.random.default_rng(3)
    kps, kout = np.sort(rng.uniform(0.004, 0.3, 37)), np.sort(rng.uniform(0.01, 0.28, 11))
    PS = rng.normal(size=(3, 4, 37))
    want = F.dpcorr(kout, kps, PS, 3, ktrust=0.22, fs=0.3, Dfc=0.5)
    got = np.einsum("alxk,lnk->anx", TB.fiber_operator(kps, 3, 0.3, 0.5, 0.22, kout=kout), PS)
    assert relerr(got.reshape(12, -1), want.reshape(12, -1)) < 1e-12
    # order in the folded operator: window -> fibre -> binning -> chained (reference theory.py:583-600)
    W = rng.normal(size=(3, 3, 50, 50)) * 0.1
    Bn = rng.normal(size=(7, 50))
    P = rng.normal(size=(3, 5, 50))
    op = TB.compose_operator(3, 50, Wfold=W, binning=Bn, chained=True, fiber=Fm)
    x = np.einsum("alxk,lnk->anx", W, P)
    x = x + np.einsum("alxk,lnk->anx", Fm, x)
    x = np.einsum("bx,anx->anb", Bn, x)
    x = np.einsum("ca,anb->cnb", TB.chained_matrix(3), x)
    assert relerr(np.einsum("alxk,lnk->anx", op, P).reshape(10, -1), x.reshape(10, -1)) < 1e-12


def test_window_matrix_host_side(golden):
    """to_window_matrix / WindowMatrix.convolve mirrors (reference window.py:426-564) against reference outputs"""
    from eftpipe_amd import pybird
    from eftpipe_amd import window as W

    g, c = golden("wmat"), golden("caseC")
    co = pybird.Common(Nl=3, kmax=0.3)
    m = W.to_window_matrix(g["stacked"].astype(np.float64), W.PInfo((0, 2, 4), 0, 0.4, 400), W.PInfo((0, 1, 2, 3, 4), 0, 0.4, 40), (0, 2, 4), co.k.max(),
                           tuple(g["ells"]), float(g["kmin"]), float(g["kmax"]))
    assert m.shape == tuple(g["matrix_shape"]) and np.array_equal(m[:, :, ::5, ::37], g["matrix_spot"])
    wm = W.WindowMatrix(m, W.PolesInfo(3, 0, co.k.max(), m.shape[3]), W.PolesInfo(2, float(g["kmin"]), float(g["kmax"]), m.shape[2]), co=co)
    for n in ("P11l", "Pctl", "Ploopl"):
        assert relerr(wm.convolve(c["ap_" + n]), g["wm_" + n]) < 1e-10, n
    with pytest.raises(ValueError):
        W.WindowMatrix(m[:, :2], W.PolesInfo(3, 0, 0.3, 300), W.PolesInfo(2, 0.02, 0.2, 18), co=co)


@pytest.mark.parametrize("case", ["caseC", "caseG"])
def test_window_tables_against_reference(golden, case):
    """tables.window_tables / window_matrix (FFT + power-law sum collapsed into one real table per l) + window_fold reproduce the
    reference's Waldk (window.py:262-359): one k row and the p sums held by the fixtures."""
    import os

    from eftpipe_amd import tables as TB

    g = golden(case)
    tab = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "win_NGC_LRG_sQ024.npy"))
    Wal, p = TB.window_matrix(g["k"], tab[:, 0], tab[:, 1:].T, 3, 3)
    _, Waldk = TB.window_fold(g["k"], Wal, p)
    assert relerr(Waldk[:, :, 10, :], g["window_Waldk_k10"]) < 1e-9
    if "window_Waldk_sum_p" in g:
        assert relerr(Waldk.sum(axis=-1), g["window_Waldk_sum_p"]) < 1e-9


def test_cfg3_host_tables_and_joint_rows(golden):
    """BASELINE cfg 3 at the reference's production settings, host side: the window table builder at accboost 4 / windowk 0.1 against
    the reference's Waldk rows for the LRG and ELG windows (the cross window goes through the oracle in test_oracle_golden.py), the folded
    window -> binning [-> chained] operators against the reference's binned / chained templates, and joint_gaussian_rows + data_index
    against EFTLike's P_NG / P_G on the reference's DR16 data layout."""
    import cfg3_util as U
    from eftpipe_amd import tables as TB
    from eftpipe_amd.marginal import data_index, joint_gaussian_rows

    g = golden("cfg3")
    k = g["k"]
    assert np.array_equal(TB.window_pgrid(float(k.max()), 4), g["window_p"])
    for t in ("LRG_NGC", "ELG_NGC"):
        tab = U.window_table(t)
        Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3, accboost=4)
        Wfold, Waldk = TB.window_fold(k, Wal, p, windowk=0.1)
        assert relerr(Waldk[:, :, 10, :], g[t + "_Waldk_k10"]) < 1e-9 and relerr(Waldk[:, :, 37, :], g[t + "_Waldk_k37"]) < 1e-9
        assert relerr(Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
        Bm, keff, _, _ = TB.binning_operator(k, g[t + "_kout"])
        assert relerr(keff[None], g[t + "_keff"][None]) < 1e-13
        op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, chained=U.CHAINED[t])
        want = U.final_templates(g, t)
        for n in U.NAMES:
            assert relerr(np.einsum("alxk,lnk->anx", op, g[f"{t}_ap_{n}"]), want[n]) < 1e-9, (t, n)
    # rows x final templates == EFTLike.PNG / PG
    nb = 18
    p = U.params(g)
    fs = [float(g[t + "_f"]) for t in U.TRACERS]
    index = np.concatenate([data_index([int(l) for l in g[t + "_ls"]], U.masks(g, t), nb, tracer=i, nl=3) for i, t in enumerate(U.TRACERS)])
    assert index.size == g["data_vector"].size == 142
    block = np.zeros((3, 3, 24, nb))
    for i, t in enumerate(U.TRACERS):
        st = U.final_templates(g, t)
        no, nx = st["P11l"].shape[0], st["P11l"].shape[-1]
        block[i, :no, :, :nx] = np.concatenate([st[n] for n in U.NAMES], axis=1)
    for tag in ("full", "xnost"):
        names = [str(n) for n in g[tag + "_names"]]
        rows = joint_gaussian_rows(U.bases(), fs, p, names, U.scales(g))
        V = np.einsum("tgr,tlrx->gtlx", rows, block).reshape(len(names) + 1, -1)[:, index]
        assert relerr(V[0][None], g["PNG"][None]) < 1e-13 and relerr(V[1:], g[tag + "_PG"]) < 1e-13, tag


def test_cfg3_nk512_host_tables_at_production_window(golden):
    """BASELINE cfg 3 on the BASELINE grid (Nk = 512) with the window the reference ships (accboost 4, windowk 0.1; yaml :63-65), host side:
    the cross-spectrum window `win_NGC_X` through the table builder against the reference's own Waldk rows / p sums at Np = 1540, the folded
    window operator against the reference's convolved templates, window -> binning against its binned templates
    (tests/golden/cfg3_nk512.npz, tools/make_fixtures.py cfg3_nk512; reference window.py:27-33, 262-415)."""
    import cfg3_util as U
    from eftpipe_amd import tables as TB

    g = golden("cfg3_nk512")
    k, t = g["k"], "X_NGC"
    assert k.size == 512 and int(g["accboost"]) == 4 and float(g["windowk"]) == 0.1
    assert np.array_equal(TB.window_pgrid(float(k.max()), 4), g["window_p"]) and g["window_p"].size == 1540
    tab = U.window_table(t)
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3, accboost=4)
    Wfold, Waldk = TB.window_fold(k, Wal, p, windowk=0.1)
    assert relerr(Waldk[:, :, 100, :], g[t + "_Waldk_k100"]) < 1e-9 and relerr(Waldk[:, :, 411, :], g[t + "_Waldk_k411"]) < 1e-9
    assert relerr(Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
    Bm, keff, _, _ = TB.binning_operator(k, g[t + "_kout"])
    assert relerr(keff[None], g[t + "_keff"][None]) < 1e-13
    op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm)
    for n in U.NAMES:
        assert relerr(np.einsum("alxk,lnk->anx", Wfold, g[f"{t}_ap_{n}"]), g[f"{t}_window_{n}"]) < 1e-9, n
        assert relerr(np.einsum("alxk,lnk->anx", op, g[f"{t}_ap_{n}"]), g[f"{t}_binned_{n}"]) < 1e-9, n


def test_cfg5_acc4_host_tables_nk2048(golden):
    """BASELINE cfg 5 at the shipped window accuracy (accboost 4, windowk 0.1) on the Nk = 2048 grid, host side: the table builder at
    Np = 1540 against the reference's own Waldk rows / p sums, the folded window operator against every 4th k of the reference's convolved
    templates and window -> binning against its binned templates (tests/golden/cfg5_acc4.npz, tools/make_fixtures.py cfg5_acc4: the
    reference's precompute takes 11 minutes here)."""
    import cfg3_util as U
    from eftpipe_amd import tables as TB

    g, f = golden("cfg5_acc4"), golden("caseF")
    k = f["k"]
    assert k.size == 2048 and g["window_p"].size == 1540
    tab = U.window_table("LRG_NGC")
    Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3, accboost=4)
    assert np.array_equal(p, g["window_p"])
    Wfold, Waldk = TB.window_fold(k, Wal, p, windowk=0.1)
    del Wal
    assert relerr(Waldk[:, :, 1000, :], g["LRG_Waldk_k1000"]) < 1e-9 and relerr(Waldk[:, :, 77, :], g["LRG_Waldk_k77"]) < 1e-9
    assert relerr(Waldk.sum(axis=-1), g["LRG_Waldk_sum_p"]) < 1e-9
    del Waldk
    Bm, keff, _, _ = TB.binning_operator(k, g["kout"])
    assert relerr(keff[None], g["keff"][None]) < 1e-13
    op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm)
    for n in U.NAMES:
        assert relerr(np.einsum("alxk,lnk->anx", Wfold[:, :, ::4], f["ap_" + n]), g[f"LRG_window_{n}_k512"]) < 1e-9, n
        assert relerr(np.einsum("alxk,lnk->anx", op, f["ap_" + n]), g[f"LRG_binned_{n}"]) < 1e-9, n


def test_integral_constraint_against_oracle(tmp_path):
    """SURVEY 8(f) rank 4, PARITY UNPINNED (the reference's interp2d call no longer exists in SciPy, so it cannot produce fixtures):
    eftpipe_amd.icc.IntegralConstraint -- PSN, the 2-D FFTLog matrix with the documented interp2d replacement, mask / dp weights, the
    host convolution and the reference's cache format (.npz keys PSN, Wal + .json meta) -- against the oracle's line-by-line restatement
    of reference icc.py:119-500 / fftlog2d.py:13-166."""
    import json

    from eftpipe_amd import pybird
    from eftpipe_amd.icc import FFTLog2D, ICpannel_to_ndarray, IntegralConstraint, bessel_matrix
    from eftpipe_amd.window import MetaInfoError
    from icc_util import ICC_KW, make_icc_files
    from oracle import icc as O

    sn, ic, s, xi, s1 = make_icc_files(tmp_path)
    co = pybird.Common(Nl=3, kmax=0.3)
    cache = tmp_path / "icc_cache.npz"
    obj = IntegralConstraint(Pshot=1500.0, icc_fourier_file=str(cache), icc_configspace_SN_file=sn, icc_configspace_IC_file=ic, co=co, load=True,
                             save=True, **ICC_KW)
    k, p = co.k, obj.p
    psn = O.compute_psn(k, s, xi, 3, Nmax=ICC_KW["Nmax"])
    assert relerr(obj.PSN, 1500.0 * psn) < 1e-12
    panel, meta = ICpannel_to_ndarray(np.load(ic))
    assert panel.shape == (3, 3, 36, 36) and np.array_equal(meta["s1"], s1)
    Wal = O.compute_wal(k, p, s1, s1, panel, 3, 3, Nxmax=128, Nymax=128)
    assert relerr(obj.Wal.reshape(9, -1), Wal.reshape(9, -1)) < 1e-11
    assert relerr(obj.Waldk.reshape(9, -1), O.waldk(k, p, Wal).reshape(9, -1)) < 1e-11
    rng = np.random.default_rng(5)
    P = rng.normal(size=(3, 4, k.size)) * 1e3
    assert relerr(obj.integrWindow(P).reshape(12, -1), O.integr_window(k, p, O.waldk(k, p, Wal), P).reshape(12, -1)) < 1e-9
    # pieces
    nu = -2.0 + 1.5j
    assert np.isclose(bessel_matrix(nu, 2), O.bessel_matrix(nu, 2))
    f2 = FFTLog2D(64, 64, 1e-3, 2500.0, 1e-3, 2500.0, -2.0, -2.0)
    assert np.allclose(f2.window(0.5), O.taper2d(64, 64, 0.5))
    # cache: written in the reference's format, loaded back bit-identically, meta mismatch refused
    with np.load(cache) as z:
        assert sorted(z.files) == ["PSN", "Wal"] and z["Wal"].shape == (3, 3, k.size, p.size)
        assert relerr(z["PSN"], psn) < 1e-12  # stored WITHOUT the Pshot factor, as the reference does (icc.py:276-278)
    with open(cache.with_suffix(".json")) as fh:
        assert json.load(fh)["Nxmax"] == 128
    again = IntegralConstraint(Pshot=2.0, icc_fourier_file=str(cache), co=co, **ICC_KW)
    assert np.array_equal(again.Wal, obj.Wal) and relerr(again.PSN, 2.0 * psn) < 1e-12
    with pytest.raises(MetaInfoError):
        IntegralConstraint(Pshot=2.0, icc_fourier_file=str(cache), co=co, Nmax=512, Nxmax=128, Nymax=128)
    with pytest.raises(ValueError):
        IntegralConstraint(Pshot=1.0, co=co)


def test_cfg5_host_operators_nk2048(golden):
    """BASELINE cfg 5 after the projection stages, host side: the window table builder + fold at Nk = 2048 and the folded
    window -> binning [-> chained] operators against the reference's own Waldk rows, convolved templates and binned / chained results
    (tests/golden/cfg5.npz: reference Window + Binning + Chained on the Nk = 2048 AP-stage templates of caseF, LRG and ELG windows)."""
    import cfg3_util as U
    from eftpipe_amd import tables as TB

    g, F = golden("cfg5"), golden("caseF")
    k = F["k"]
    assert k.size == 2048
    Bm, keff, _, _ = TB.binning_operator(k, g["kout"])
    assert relerr(keff[None], g["keff"][None]) < 1e-13
    for t, chained in (("LRG", False), ("ELG", True)):
        tab = U.window_table(t + "_NGC")
        Wal, p = TB.window_matrix(k, tab[:, 0], tab[:, 1:].T, 3, 3)
        assert np.array_equal(p, g["window_p"])
        Wfold, Waldk = TB.window_fold(k, Wal, p)
        assert relerr(Waldk[:, :, 1000, :], g[t + "_Waldk_k1000"]) < 1e-9 and relerr(Waldk[:, :, 77, :], g[t + "_Waldk_k77"]) < 1e-9
        assert relerr(Waldk.sum(axis=-1), g[t + "_Waldk_sum_p"]) < 1e-9
        for n in U.NAMES:
            conv = np.einsum("alxk,lnk->anx", Wfold, F["ap_" + n])
            assert relerr(conv[..., ::4], g[f"{t}_window_{n}_k512"]) < 1e-9, (t, n)
        op = TB.compose_operator(3, k.size, Wfold=Wfold, binning=Bm, chained=chained)
        for n in U.NAMES:
            want = g[f"{t}_{'chained' if chained else 'binned'}_{n}"]
            assert relerr(np.einsum("alxk,lnk->anx", op, F["ap_" + n]), want) < 1e-9, (t, n)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's recorded command shape): the script spawns two fresh rank
    processes itself and they meet on the control plane (EFTB_BENCH_RENDEZVOUS_ONLY stops them there: this box has no GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["EFTB_BENCH_RENDEZVOUS_ONLY"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line == {"rendezvous": "ok", "world": 2, "highest_rank_seen": 1}
    # a rank that fails makes the launcher fail
    env["EFTB_BENCH_RENDEZVOUS_ONLY"] = ""
    env["EFTB_LIB"] = "/nonexistent/libeftbird.so"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
