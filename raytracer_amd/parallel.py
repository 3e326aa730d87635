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


def gather_strips(buffers: List[torch.Tensor], height: int, strip_rows: int, world: int, rank: int,
                  group: Optional[dist.ProcessGroup] = None, force: bool = False) -> None:
    """Complete full-frame `buffers` (each shaped [height, ...], only this rank's rows
    valid) in place with the rows of every other rank.  `force` runs the pack / all-gather
    / unpack even for a single rank (a self-test of the collective path)."""
    if world <= 1 and not force:
        return
    rows = [strip_rows_of(height, strip_rows, world, r) for r in range(world)]
    max_rows = max(int(r.numel()) for r in rows)
    for buf in buffers:
        assert buf.shape[0] == height
        mine = rows[rank].to(buf.device)
        send = torch.zeros((max_rows,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
        send[: mine.numel()] = buf.index_select(0, mine)
        recv = torch.empty((world, max_rows) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
        dist.all_gather_into_tensor(recv.view(world * max_rows, *buf.shape[1:]), send, group=group)
        for r in range(world):
            if r == rank and not force:
                continue
            idx = rows[r].to(buf.device)
            buf.index_copy_(0, idx, recv[r, : idx.numel()])


def frames_of(n_frames: int, world: int, rank: int) -> List[int]:
    """Frame indices (= frameCount values) rendered by `rank` when samples are sharded."""
    return [f for f in range(n_frames) if f % world == rank]


def reduce_samples(accum: torch.Tensor, counts: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """Sum the per-rank accumulation buffers in place on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
