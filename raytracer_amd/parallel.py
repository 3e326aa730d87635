"""Partition of the work over the GPUs of one node, one process per GPU, over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Both partitions rest on one property of the reference shader: a pixel's result depends
only on (x, y, time, frameCount) and the replicated scene (shaders/raytracer.comp:959-961),
and frames couple only through the additive accumulation buffers (990-992).

* Row strips (one frame split over ranks): rank r renders the rows y with
  (y // strip_rows) % world == r, using global pixel coordinates, so every pixel gets the
  bits it would get on one GPU.  The frame is completed by an all-gather of each rank's
  packed rows — the only data-path collective.
* Samples (frames split over ranks): rank r renders frames f with f % world == r into a
  private zero-initialised accumulator; one all-reduce(sum) of accumulated_colors and
  sample_counts per batch of frames combines them.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def strip_rows_of(height: int, strip_rows: int, world: int, rank: int) -> torch.Tensor:
    """Row indices (ascending) owned by `rank`."""
    y = torch.arange(height, dtype=torch.long)
    if strip_rows <= 0 or world <= 1:
        return y if rank == 0 or strip_rows <= 0 else y[:0]
    return y[((y // strip_rows) % world) == rank]


class StripExchange:
    """Completes full-frame buffers (each shaped [height, ...], only this rank's rows valid)
    with the rows of every other rank: pack own rows, one all-gather per buffer, unpack the
    other ranks' rows.  Index tensors and staging buffers are built once.

    The two halves can be issued separately: after `pack()` the caller may already start
    rendering the next frame into its own rows (the all-gather reads the packed copy, and
    `exchange()` writes only rows this rank does not own), so the collective overlaps the
    next frame's kernel."""

    def __init__(self, buffers: List[torch.Tensor], height: int, strip_rows: int, world: int, rank: int,
                 group: Optional[dist.ProcessGroup] = None, include_self: bool = False):
        self.buffers = buffers
        self.world, self.rank, self.group = world, rank, group
        rows = [strip_rows_of(height, strip_rows, world, r) for r in range(world)]
        self.max_rows = max(int(r.numel()) for r in rows)
        dev = buffers[0].device
        self.mine = rows[rank].to(dev)
        # rows of the ranks before / after this one, and where they sit in the gathered array
        self.parts = []
        for lo, hi in ((0, rank), (rank + 1, world)) if not include_self else ((0, world),):
            if hi <= lo:
                continue
            dst = torch.cat([rows[r] for r in range(lo, hi)]).to(dev)
            src = torch.cat([torch.arange(rows[r].numel()) + r * self.max_rows for r in range(lo, hi)]).to(dev)
            self.parts.append((dst, src))
        self.send, self.recv = [], []
        for buf in buffers:
            assert buf.shape[0] == height
            tail = tuple(buf.shape[1:])
            self.send.append(torch.zeros((self.max_rows,) + tail, dtype=buf.dtype, device=dev))
            self.recv.append(torch.empty((world * self.max_rows,) + tail, dtype=buf.dtype, device=dev))

    def pack(self) -> None:
        n = int(self.mine.numel())
        for buf, send in zip(self.buffers, self.send):
            torch.index_select(buf, 0, self.mine, out=send[:n])

    def exchange(self) -> None:
        for buf, send, recv in zip(self.buffers, self.send, self.recv):
            dist.all_gather_into_tensor(recv, send, group=self.group)
            for dst, src in self.parts:
                buf.index_copy_(0, dst, recv.index_select(0, src))


_EXCHANGES: dict = {}


def gather_strips(buffers: List[torch.Tensor], height: int, strip_rows: int, world: int, rank: int,
                  group: Optional[dist.ProcessGroup] = None, force: bool = False) -> None:
    """Complete full-frame `buffers` in place with the rows of every other rank (see
    StripExchange).  `force` runs the pack / all-gather / unpack even for a single rank and
    unpacks this rank's own rows too (a self-test of the collective path)."""
    if world <= 1 and not force:
        return
    key = (tuple((b.data_ptr(), tuple(b.shape), b.dtype) for b in buffers), height, strip_rows, world, rank, id(group), force)
    ex = _EXCHANGES.get(key)
    if ex is None:
        _EXCHANGES.clear()  # one live configuration at a time; keeps stale device buffers from piling up
        ex = _EXCHANGES[key] = StripExchange(buffers, height, strip_rows, world, rank, group, include_self=force)
    ex.pack()
    ex.exchange()


def frames_of(n_frames: int, world: int, rank: int) -> List[int]:
    """Frame indices (= frameCount values) rendered by `rank` when samples are sharded."""
    return [f for f in range(n_frames) if f % world == rank]


def reduce_samples(accum: torch.Tensor, counts: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """Sum the per-rank accumulation buffers in place on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
