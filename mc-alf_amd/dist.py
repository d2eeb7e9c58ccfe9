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

import ctypes as C
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib


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
    same number of elements (a requirement of gather on both RCCL and gloo).

    `depth` > 1 gives a ring of send/receive buffers so that the gather of batch k (issued with
    `gather_async`) overlaps the evaluation of batch k+1 into the next buffer: the collective is
    latency-bound (8 KB per rank at BASELINE config B) and would otherwise serialise with a
    ~0.1 ms kernel.  `local` always points at the buffer the next evaluation should fill and waits
    for that buffer's previous collective first."""

    def __init__(self, batch: int, device, dst: int = 0, group=None, dtype=torch.float64, depth: int = 1,
                 always_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst = dst
        # a single rank normally skips the collective; `always_collective` issues it anyway (a one-rank
        # gather), which lets one GPU exercise the RCCL plumbing of the multi-GPU path
        self.collective = dist.is_initialized() and (self.world > 1 or always_collective)
        self.batch = batch
        self.depth = max(1, int(depth))
        self.counts = shard_counts(batch, self.world)
        self.lo, self.hi = shard_bounds(batch, self.world, self.rank)
        self.width = max(self.counts) if self.counts else 0
        self._send = [torch.zeros(self.width, dtype=dtype, device=device) for _ in range(self.depth)]
        self._recv = ([[torch.empty(self.width, dtype=dtype, device=device) for _ in range(self.world)]
                       for _ in range(self.depth)] if self.rank == dst else [None] * self.depth)
        self._work = [None] * self.depth
        self._slot = 0
        self._last = None

    # kept for callers that index the single-buffer form
    @property
    def send(self) -> torch.Tensor:
        return self._send[self._slot]

    @property
    def local(self) -> torch.Tensor:
        """View the local evaluator should write its `hi - lo` logL values into (current slot)."""
        w = self._work[self._slot]
        if w is not None:                      # the slot's previous collective must have consumed it
            # A finished collective needs nothing; only an unfinished one costs a stream-level wait (on
            # RCCL that is a barrier packet in the launch stream, ~5 us of queue time per step).
            if not w.is_completed():
                w.wait()
            self._work[self._slot] = None
        return self._send[self._slot][: self.hi - self.lo]

    def _assemble(self, slot) -> Optional[torch.Tensor]:
        if not self.collective:
            return self._send[slot][: self.hi - self.lo]
        if self.rank != self.dst:
            return None
        return torch.cat([buf[:c] for buf, c in zip(self._recv[slot], self.counts)])

    def gather(self) -> Optional[torch.Tensor]:
        """Blocking form: run the collective on the current slot; full [batch] vector on `dst`."""
        slot = self._slot
        if self.collective:
            dist.gather(self._send[slot], self._recv[slot], dst=self.dst, group=self.group)
        self._slot = (slot + 1) % self.depth
        return self._assemble(slot)

    def gather_async(self) -> None:
        """Issue the collective for the current slot and move on to the next slot."""
        slot = self._slot
        if self.collective:
            self._work[slot] = dist.gather(self._send[slot], self._recv[slot], dst=self.dst, group=self.group,
                                           async_op=True)
        self._last = slot
        self._slot = (slot + 1) % self.depth

    def finish(self) -> Optional[torch.Tensor]:
        """Wait for every outstanding collective; returns the most recently gathered vector on `dst`."""
        for i, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[i] = None
        return None if self._last is None else self._assemble(self._last)


def sharded_loglike(evaluate: Callable[[int, int, torch.Tensor], None], batch: int, device, dst: int = 0,
                    group=None) -> Optional[torch.Tensor]:
    """One-shot helper: `evaluate(lo, hi, out)` fills `out` with logL of rows [lo, hi)."""
    g = LogLGather(batch, device, dst=dst, group=group)
    evaluate(g.lo, g.hi, g.local)
    return g.gather()


class InLibGather:
    """The same exchange done INSIDE the library: the context owns an RCCL communicator and
    `mcalf_loglike_gatherv_device` enqueues the kernels on the launch stream and one grouped send / receive on a
    context-owned stream behind them -- no Python and no host round trip in the step.  torch.distributed (any
    backend) is used once, to hand the root's 128-byte RCCL id to the other ranks; without an initialised process
    group this is a one-rank communicator.

    `batch` is the JOB-WIDE number of rows; shards are the contiguous blocks of `shard_bounds` (ragged allowed).
    Two (local, all) buffer pairs alternate so that the exchange of step k overlaps the kernels of step k+1
    (`mcalf_comm_set_overlap`); `finish()` joins the outstanding exchanges and returns the last gathered vector
    on `root` (None elsewhere).  `root` is a rank of `group`."""

    def __init__(self, fit, batch: int, device, root: int = 0, group=None, depth: int = 2):
        self.fit, self.root = fit, int(root)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if not (0 <= self.root < self.world):
            raise ValueError(f"root {root} is not a rank of a {self.world}-rank group")
        self.batch = int(batch)
        self.counts = shard_counts(self.batch, self.world)
        self.lo, self.hi = shard_bounds(self.batch, self.world, self.rank)
        self.n = self.hi - self.lo
        if self.world > 1:
            # every rank must pass the SAME count vector to mcalf_loglike_gatherv_device (a receive posted for a count
            # the sender does not send is a hang nothing can detect locally): compare them once, here
            seen = [None] * self.world
            dist.all_gather_object(seen, (self.batch, tuple(self.counts)), group=group)
            if any(v != seen[0] for v in seen):
                raise ValueError(f"InLibGather: the ranks disagree on the shard counts: {seen}")
        ident = [None]
        if self.rank == self.root:
            buf = C.create_string_buffer(_lib.MCALF_COMM_ID_BYTES)
            _lib.check(fit._lib.mcalf_comm_unique_id(buf))
            ident[0] = buf.raw
        if self.world > 1:
            # broadcast_object_list takes the GLOBAL rank of the source, `root` is a rank of `group`
            src = dist.get_global_rank(group, self.root) if group is not None else self.root
            dist.broadcast_object_list(ident, src=src, group=group)
        self._id = C.create_string_buffer(ident[0], _lib.MCALF_COMM_ID_BYTES)
        _lib.check(fit._lib.mcalf_comm_init(fit._ctx, self._id, self.world, self.rank), fit._ctx)
        self.depth = 2 if depth >= 2 else 1
        _lib.check(fit._lib.mcalf_comm_set_overlap(fit._ctx, 1 if self.depth == 2 else 0), fit._ctx)
        self._counts = (C.c_int64 * self.world)(*self.counts)
        self._local = [torch.empty(max(self.n, 1), dtype=torch.float64, device=device)[: self.n] for _ in range(self.depth)]
        self._all = ([torch.empty(self.batch, dtype=torch.float64, device=device) for _ in range(self.depth)]
                     if self.rank == self.root else [None] * self.depth)
        self._slot, self._last = 0, None

    @property
    def local(self) -> torch.Tensor:
        """This rank's logL block of the most recent step."""
        return self._local[self._last if self._last is not None else 0]

    @property
    def all(self) -> Optional[torch.Tensor]:
        """The gathered vector of the most recent step on `root` (valid after `finish()` or a stream sync)."""
        return self._all[self._last if self._last is not None else 0]

    def step(self, dP: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Evaluate this rank's rows `dP` [hi - lo][ndim] and enqueue the gather (asynchronous)."""
        st = stream if stream is not None else torch.cuda.current_stream()
        k = self._slot
        rc = self.fit._lib.mcalf_loglike_gatherv_device(
            self.fit._ctx, dP.data_ptr() if self.n else None, self.n, self._local[k].data_ptr() if self.n else None,
            self._all[k].data_ptr() if self._all[k] is not None else None, self._counts, self.root,
            C.c_void_p(st.cuda_stream))
        self._last, self._slot = k, (k + 1) % self.depth
        if rc:
            _lib.check(rc, self.fit._ctx)

    def finish(self, stream: Optional[torch.cuda.Stream] = None) -> Optional[torch.Tensor]:
        """Make `stream` wait for every outstanding exchange, synchronise it, return the last gathered vector."""
        st = stream if stream is not None else torch.cuda.current_stream()
        _lib.check(self.fit._lib.mcalf_comm_join(self.fit._ctx, C.c_void_p(st.cuda_stream)), self.fit._ctx)
        st.synchronize()
        return self.all

    def close(self):
        _lib.check(self.fit._lib.mcalf_comm_destroy(self.fit._ctx), self.fit._ctx)
