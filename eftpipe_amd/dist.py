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

import hashlib
import hmac
import ipaddress
import os
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


# ---------------------------------------------------------------------------------------------------------------------------------
# Wire format: typed frames, no pickle -- nothing a peer sends is ever executed or unpickled.
#   frame  = MAGIC(4) kind(u8) + body
#   kind 0 = None | 1 = float64 | 2 = int64 | 3 = bytes (u64 length) | 4 = ndarray (dtype code u8, ndim u8, shape u64[ndim], raw C-order
#   bytes) | 5 = list (u32 count, then that many frames; one level, arrays / None only)
# Limits (a bogus peer cannot make the hub allocate at will): 2 GiB per frame, 8 dimensions, 4096 list entries.
# ---------------------------------------------------------------------------------------------------------------------------------
_MAGIC = b"EFTB"
_MAX_BYTES = 1 << 31
_DTYPES = ["<f8", "<f4", "<i8", "<i4", "<u8", "<u4", "|u1", "|b1"]   # the only array types the control plane carries


class ProtocolError(ConnectionError):
    """A peer sent something that is not a control-plane frame (or exceeded a limit); the connection is dropped."""


def _exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _encode(obj, depth=0):
    if obj is None:
        return _MAGIC + b"\x00"
    if isinstance(obj, (bool, np.bool_)):
        obj = int(obj)
    if isinstance(obj, (int, np.integer)):
        return _MAGIC + b"\x02" + struct.pack("<q", int(obj))
    if isinstance(obj, (float, np.floating)):
        return _MAGIC + b"\x01" + struct.pack("<d", float(obj))
    if isinstance(obj, (bytes, bytearray)):
        return _MAGIC + b"\x03" + struct.pack("<Q", len(obj)) + bytes(obj)
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        code = a.dtype.newbyteorder("<").str if a.dtype.byteorder == "=" else a.dtype.str
        if code not in _DTYPES or a.ndim > 8:
            raise TypeError(f"control plane: arrays of dtype {a.dtype} / {a.ndim} dimensions are not carried")
        return _MAGIC + b"\x04" + struct.pack("<BB", _DTYPES.index(code), a.ndim) + struct.pack("<%dQ" % a.ndim, *a.shape) + a.tobytes()
    if isinstance(obj, (list, tuple)) and depth == 0:
        return _MAGIC + b"\x05" + struct.pack("<I", len(obj)) + b"".join(_encode(x, 1) for x in obj)
    raise TypeError(f"control plane: cannot send {type(obj).__name__}")


def _send(sock, obj):
    sock.sendall(_encode(obj))


def _recv(sock, depth=0):
    head = _exact(sock, 5)
    if head[:4] != _MAGIC:
        raise ProtocolError("control plane: not a control-plane frame")
    kind = head[4]
    if kind == 0:
        return None
    if kind == 1:
        return struct.unpack("<d", _exact(sock, 8))[0]
    if kind == 2:
        return struct.unpack("<q", _exact(sock, 8))[0]
    if kind == 3:
        (n,) = struct.unpack("<Q", _exact(sock, 8))
        if n > _MAX_BYTES:
            raise ProtocolError("control plane: frame too large")
        return _exact(sock, n)
    if kind == 4:
        code, ndim = struct.unpack("<BB", _exact(sock, 2))
        if code >= len(_DTYPES) or ndim > 8:
            raise ProtocolError("control plane: bad array header")
        shape = struct.unpack("<%dQ" % ndim, _exact(sock, 8 * ndim))
        dt = np.dtype(_DTYPES[code])
        count = 1
        for d in shape:
            count *= d
            if count * dt.itemsize > _MAX_BYTES:
                raise ProtocolError("control plane: frame too large")
        return np.frombuffer(_exact(sock, count * dt.itemsize), dtype=dt).reshape(shape).copy()
    if kind == 5 and depth == 0:
        (n,) = struct.unpack("<I", _exact(sock, 4))
        if n > 4096:
            raise ProtocolError("control plane: list too long")
        return [_recv(sock, 1) for _ in range(n)]
    raise ProtocolError("control plane: unknown frame kind %d" % kind)


def _peer_uid(sock):
    """uid of the process at the other end of an AF_UNIX stream socket (SO_PEERCRED: pid, uid, gid)."""
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    return struct.unpack("3i", cred)[1]


def _is_loopback(addr):
    try:
        return all(ipaddress.ip_address(info[4][0]).is_loopback for info in socket.getaddrinfo(addr, None))
    except (OSError, ValueError):
        return False


def _tag(token, nonce, rank):
    return hmac.new(token, nonce + struct.pack("<I", rank), hashlib.sha256).digest()


class ControlPlane:
    """Rendezvous / barrier / scalar reductions / host gather over local sockets (no-op when world == 1).

    Rank 0 listens on an abstract-namespace Unix socket named after MASTER_ADDR, MASTER_PORT and the launcher's run id (one node: no
    port to collide with the launcher's own store, nothing left on disk); set ``EFTB_CP_TCP_PORT`` to use TCP on MASTER_ADDR instead.
    Every collective is one round trip to the hub; all ranks must call the same collectives in the same order.

    Who may join.  Frames are typed (never unpickled), and a connection becomes a peer only after a handshake:
    * Unix socket: the kernel's SO_PEERCRED uid of the other end must be this process's uid (both directions);
    * TCP: the hub binds to a loopback MASTER_ADDR only (``EFTB_CP_TCP_ANY=1`` opts in to another interface) and both sides must hold
      the shared secret ``EFTB_CP_TOKEN`` from the launcher's environment (HMAC-SHA256 over fresh nonces, both directions);
      the token is also checked on the Unix socket when it is set.
    A connection that fails the handshake is closed and the hub keeps waiting for the real ranks."""

    HANDSHAKE_TIMEOUT = 10.0   # a silent foreign connection cannot hold the accept loop longer than this

    def __init__(self, timeout=300.0):
        self.rank, self.local_rank, self.world = env_rank()
        self.peers = []      # hub: one socket per rank 1..world-1, in rank order
        self.hub = None      # spokes: the socket to rank 0
        self._listener = None
        self.rejected = 0    # hub: connections turned away during the rendezvous
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = os.environ.get("MASTER_PORT", "29511")
        tcp_port = os.environ.get("EFTB_CP_TCP_PORT")
        token = os.environ.get("EFTB_CP_TOKEN", "").encode()
        if tcp_port:
            family, target = socket.AF_INET, (addr, int(tcp_port))
            if not token:
                raise RuntimeError("control plane: the TCP transport needs the shared secret EFTB_CP_TOKEN in every rank's environment")
            if self.rank == 0 and os.environ.get("EFTB_CP_TCP_ANY") != "1" and not _is_loopback(addr):
                raise RuntimeError(f"control plane: MASTER_ADDR {addr} is not a loopback address; set EFTB_CP_TCP_ANY=1 to listen on it")
        else:
            run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "-" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
            family, target = socket.AF_UNIX, "\0eftb-cp-%s-%s-%s" % (addr, port, run)
        unix = family == socket.AF_UNIX
        if self.rank == 0:
            srv = socket.socket(family, socket.SOCK_STREAM)
            if not unix:
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(target)
            srv.listen(self.world + 8)
            srv.settimeout(timeout)
            self._listener = srv
            got = {}
            while len(got) < self.world - 1:
                conn, _ = srv.accept()
                r = self._admit(conn, unix, token, got)
                if r is None:
                    self.rejected += 1
                    conn.close()
                    continue
                conn.settimeout(timeout)
                got[r] = conn
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
            if unix and _peer_uid(s) != os.getuid():
                s.close()
                raise ConnectionError("control plane: the hub socket belongs to another user")
            hub_nonce = _exact(s, 16)
            my_nonce = os.urandom(16)
            s.sendall(struct.pack("<I", self.rank) + my_nonce + _tag(token, hub_nonce, self.rank))
            try:
                answer = _exact(s, 32)
            except ConnectionError:
                s.close()
                raise ConnectionError("control plane: rank 0 refused this rank (EFTB_CP_TOKEN mismatch, duplicate rank or foreign user)") from None
            if not hmac.compare_digest(answer, _tag(token, my_nonce, 0)):
                s.close()
                raise ConnectionError("control plane: the hub does not hold EFTB_CP_TOKEN")
            self.hub = s

    def _admit(self, conn, unix, token, got):
        """Hub side of the handshake -> the peer's rank, or None (connection to be dropped)."""
        try:
            conn.settimeout(self.HANDSHAKE_TIMEOUT)
            if unix and _peer_uid(conn) != os.getuid():
                return None
            nonce = os.urandom(16)
            conn.sendall(nonce)
            msg = _exact(conn, 4 + 16 + 32)
            (r,) = struct.unpack("<I", msg[:4])
            if not (1 <= r < self.world) or r in got:
                return None
            if not hmac.compare_digest(msg[20:], _tag(token, nonce, r)):
                return None
            conn.sendall(_tag(token, msg[4:20], 0))
            return r
        except (OSError, struct.error):
            return None

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
