"""Data parallelism for the head: shard by image, one collective for the loss.

Every image's loss, gradient and detections are independent (model.py:296-317
sums over the batch), so the batch splits into contiguous per-rank slices and
no gradient or detection ever crosses a GPU.  With ``global_batch`` passed to
``YOLOv1Head.get_loss`` each rank's gradient is already the global one for its
slice.  The only exchange is the scalar loss: per-step sums sit in a device-side
``LossLog`` ring and ONE all-reduce (RCCL over xGMI when the backend is "nccl")
covers up to ``capacity`` steps at a time.  The payload is 16 bytes per step, so
the collective is latency-bound; batching steps amortises that latency and
nothing on the step's critical path waits for it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the batch owned by ``rank``; the first
    ``global_batch % world_size`` ranks hold one image more."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(global_batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class LossAllReducer:
    """Collects per-step loss sums and reduces them across ranks in batches.

    ``ring`` is a (capacity, 4) tensor the loss kernel appends to (``LossLog.ring``
    on the GPU; any tensor with that shape on the CPU under gloo).  ``flush(n)``
    all-reduces the first ``n`` rows asynchronously into a private copy and
    returns at once; ``results()`` waits and returns every reduced row so far as
    one (steps, 4) tensor of global sums (divide by the global batch for the
    value get_loss reports, model.py:317)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, always: bool = False):
        self.group = group
        # always: issue the collective even in a group of one rank (a rehearsal of the
        # multi-rank path on one GPU); needs an initialised process group
        self.always = bool(always)
        self._pending: List[Tuple[object, torch.Tensor]] = []

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def _collective(self) -> bool:
        if self.world_size > 1:
            return True
        return self.always and dist.is_available() and dist.is_initialized()

    def flush(self, ring: torch.Tensor, n: int) -> None:
        if n <= 0:
            return
        if n > ring.shape[0]:
            raise ValueError(f"{n} rows requested from a ring of {ring.shape[0]}")
        rows = ring[:n].clone()        # the ring is free for the next steps
        work = None
        if self._collective:
            if rows.is_cuda and dist.get_backend(self.group) == "gloo":
                # gloo moves host memory: stage through the CPU (a rehearsal path; the
                # GPU backend is "nccl" = RCCL and reduces the device buffer in place)
                host = rows.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                rows = host.to(rows.device)
            else:
                work = dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, rows))

    def flush_rows(self, rows: torch.Tensor) -> None:
        """As ``flush`` for a ring that holds every step of the run: ``rows`` (a view of
        the steps just logged) is all-reduced in place, no copy is made, and the view
        is kept for ``results()``; the caller must not let the ring wrap onto it."""
        if rows.shape[0] == 0:
            return
        work = None
        if self._collective:
            if rows.is_cuda and dist.get_backend(self.group) == "gloo":
                host = rows.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                rows.copy_(host)
            else:
                work = dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, rows))

    def results(self) -> torch.Tensor:
        out = []
        for work, rows in self._pending:
            if work is not None:
                work.wait()
            out.append(rows)
        self._pending = [(None, r) for r in out]
        if not out:
            return torch.zeros(0, 4)
        return out[0] if len(out) == 1 else torch.cat(out, dim=0)
