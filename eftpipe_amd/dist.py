"""Sharding of an MCMC batch of cosmologies over the GPUs of one node.

Each cosmology is an independent evaluation, so the batch axis is split contiguously over ranks (one
process per GPU), constant tables are replicated, and the only exchange is the gather of the
per-cosmology P_l(k) to rank 0 (RCCL over xGMI inside libeftbird; see include/eftbird.h).  The
reference has no counterpart: cobaya chains are independent MPI processes (reference README.md:29-34).

The control plane (rendezvous of the 128-byte RCCL id, barriers, max-over-ranks timing, the host gather of the
CPU tests) is a star of local stream sockets with rank 0 as the hub -- standard library only, no PyTorch: the
launcher (``python -m torch.distributed.run`` or anything else that sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT) is only asked for the environment.  No array of the data path goes through it.
"""
from __future__ import annotations

import os
import pickle
import socket
import struct
import time

import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous, balanced [start, stop) of `total` items for `rank` of `world`."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reassemble(blocks):
    """Rank-ordered shards [world][n_r, ...] -> [total, ...] in the original draw order (the inverse of ``shard_bounds`` slicing)."""
    return np.concatenate([np.asarray(b) for b in blocks], axis=0)


def env_rank():
    """(rank, local_rank, world) from the launcher's environment (1 process if absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv(sock):
    def exact(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("control plane: peer closed the connection")
            buf += chunk
        return bytes(buf)

    (n,) = struct.unpack("<Q", exact(8))
    return pickle.loads(exact(n))


class ControlPlane:
    """Rendezvous / barrier / scalar reductions / host gather over local sockets (no-op when world == 1).

    Rank 0 listens on an abstract-namespace Unix socket named after MASTER_ADDR, MASTER_PORT and the launcher's run id (one node: no
    port to collide with the launcher's own store, nothing left on disk); set ``EFTB_CP_TCP_PORT`` to use TCP on MASTER_ADDR instead.
    Every collective is one round trip to the hub; all ranks must call the same collectives in the same order."""

    def __init__(self, timeout=300.0):
        self.rank, self.local_rank, self.world = env_rank()
        self.peers = []      # hub: one socket per rank 1..world-1, in rank order
        self.hub = None      # spokes: the socket to rank 0
        self._listener = None
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = os.environ.get("MASTER_PORT", "29511")
        tcp_port = os.environ.get("EFTB_CP_TCP_PORT")
        if tcp_port:
            family, target = socket.AF_INET, (addr, int(tcp_port))
        else:
            run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "-" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
            family, target = socket.AF_UNIX, "\0eftb-cp-%s-%s-%s" % (addr, port, run)
        if self.rank == 0:
            srv = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(target)
            srv.listen(self.world)
            srv.settimeout(timeout)
            self._listener = srv
            got = {}
            while len(got) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                r = _recv(conn)
                got[int(r)] = conn
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.monotonic()
            while True:
                s = socket.socket(family, socket.SOCK_STREAM)
                try:
                    s.connect(target)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.monotonic() - t0 > timeout:
                        raise TimeoutError("control plane: rank 0 did not open its socket")
                    time.sleep(0.05)
            s.settimeout(timeout)
            _send(s, self.rank)
            self.hub = s

    # every collective: spokes send their contribution, the hub combines and answers
    def _collective(self, value, combine):
        if self.world == 1:
            return combine([value])
        if self.rank == 0:
            vals = [value] + [_recv(p) for p in self.peers]
            out = combine(vals)
            for r, p in enumerate(self.peers, start=1):
                _send(p, out(r) if callable(out) else out)
            return out(0) if callable(out) else out
        _send(self.hub, value)
        return _recv(self.hub)

    def barrier(self):
        self._collective(None, lambda v: None)

    def broadcast_bytes(self, payload, src=0):
        return self._collective(payload, lambda v: v[src])

    def max(self, value):
        return float(self._collective(float(value), lambda v: max(v)))

    def gather_host(self, arr, dst=0):
        """Host-side gather of equally shaped arrays to `dst` (rank-major) -- CPU tests only, never a data-path measurement."""
        if self.world == 1:
            return np.asarray(arr)[None]
        stacked = self._collective(np.ascontiguousarray(arr), lambda v: (lambda r: np.stack(v) if r == dst else None))
        return stacked

    def gather_ragged(self, arr, dst=0):
        """Like gather_host for shards of different lengths: -> list of arrays in rank order on `dst`, None elsewhere."""
        if self.world == 1:
            return [np.asarray(arr)]
        return self._collective(np.ascontiguousarray(arr), lambda v: (lambda r: list(v) if r == dst else None))

    def close(self):
        for s in self.peers + [self.hub, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.hub, self._listener = [], None, None
