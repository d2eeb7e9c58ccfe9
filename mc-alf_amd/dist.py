"""Live-point sharding across the GPUs of one node.

The likelihood of every live point is independent, so a batch shards by contiguous row
blocks with no data-path collective; the only exchange is the gather of the per-sample logL
back to rank 0 (`torch.distributed.gather`: RCCL over xGMI with backend "nccl", gloo on
CPU for the tests).  Per-sample arithmetic does not depend on the shard, so the gathered
vector equals the single-GPU result bit for bit.

Reference: the reference has no collective of its own; its
data-parallelism is PolyChord's MPI master/worker (cli.py:110) or jaxns' vmap (cli.py:275-280).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Row range [lo, hi) of `rank`: contiguous blocks whose sizes differ by at most one."""
    if batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard request")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(batch: int, world: int) -> List[int]:
    return [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]


class LogLGather:
    """Pre-allocated gather of per-rank logL shards to `dst` (one collective per batch).

    Shards may be ragged; they are padded to the largest shard so that every rank sends the
    same number of elements (a requirement of gather on both RCCL and gloo)."""

    def __init__(self, batch: int, device, dst: int = 0, group=None, dtype=torch.float64):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst = dst
        self.batch = batch
        self.counts = shard_counts(batch, self.world)
        self.lo, self.hi = shard_bounds(batch, self.world, self.rank)
        self.width = max(self.counts) if self.counts else 0
        self.send = torch.zeros(self.width, dtype=dtype, device=device)
        self.recv = ([torch.empty(self.width, dtype=dtype, device=device) for _ in range(self.world)]
                     if self.rank == dst else None)

    @property
    def local(self) -> torch.Tensor:
        """View the local evaluator should write its `hi - lo` logL values into."""
        return self.send[: self.hi - self.lo]

    def gather(self) -> Optional[torch.Tensor]:
        """Run the collective; returns the full [batch] vector on `dst`, None elsewhere."""
        if self.world == 1:
            return self.local
        dist.gather(self.send, self.recv, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return torch.cat([buf[:c] for buf, c in zip(self.recv, self.counts)])


def sharded_loglike(evaluate: Callable[[int, int, torch.Tensor], None], batch: int, device, dst: int = 0,
                    group=None) -> Optional[torch.Tensor]:
    """One-shot helper: `evaluate(lo, hi, out)` fills `out` with logL of rows [lo, hi)."""
    g = LogLGather(batch, device, dst=dst, group=group)
    evaluate(g.lo, g.hi, g.local)
    return g.gather()
