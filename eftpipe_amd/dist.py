"""Sharding of an MCMC batch of cosmologies over the GPUs of one node.

Each cosmology is an independent evaluation, so the batch axis is split contiguously over ranks (one
process per GPU), constant tables are replicated, and the only exchange is the gather of the
per-cosmology P_l(k) to rank 0 (RCCL over xGMI inside libeftbird; see include/eftbird.h).  The
reference has no counterpart: cobaya chains are independent MPI processes (reference README.md:29-34).

The control plane (rendezvous of the 128-byte RCCL id, barriers, max-over-ranks timing) uses
``torch.distributed`` with the gloo backend -- plumbing only; no tensor of the data path goes through it.
"""
from __future__ import annotations

import os

import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous, balanced [start, stop) of `total` items for `rank` of `world`."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def env_rank():
    """(rank, local_rank, world) from the torchrun environment (1 process if absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


class ControlPlane:
    """gloo process group for rendezvous / barrier / scalar reductions (no-op when world == 1)."""

    def __init__(self):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        if self.dist is None:
            return payload
        obj = [payload if self.rank == src else None]
        self.dist.broadcast_object_list(obj, src=src)
        return obj[0]

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def gather_host(self, arr, dst=0):
        """Host-side gather of equally shaped arrays (CPU tests and the no-RCCL fallback)."""
        if self.dist is None:
            return arr[None]
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(np.ascontiguousarray(arr), out, dst=dst)
        return np.stack(out) if self.rank == dst else None

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
