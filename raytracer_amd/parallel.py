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


def staging_rows(height: int, strip_rows: int, world: int) -> int:
    """Rows a rank's packed block has room for: rank 0's share, the largest."""
    return int(strip_rows_of(height, strip_rows, world, 0).numel())


def staging_bytes(height: int, width: int, strip_rows: int, world: int) -> int:
    """rt_strip_staging_bytes: 20 bytes per pixel of the largest share, rounded up to 16."""
    return (staging_rows(height, strip_rows, world) * width * 20 + 15) // 16 * 16


def pack_rows_torch(acc: torch.Tensor, cnt: torch.Tensor, strip_rows: int, world: int, rank: int, out: torch.Tensor) -> None:
    """The byte layout of rt_strip_pack, in torch (CPU rehearsals and as a check of the kernel): this
    rank's rows in ascending y, all float4 colours first, then all int32 counts."""
    height, width = cnt.shape
    rows = strip_rows_of(height, strip_rows, world, rank).to(acc.device)
    cap = staging_rows(height, strip_rows, world)
    a = out[:cap * width * 16].view(torch.float32).view(cap, width, 4)
    c = out[cap * width * 16:cap * width * 20].view(torch.int32).view(cap, width)
    a[:rows.numel()] = acc.index_select(0, rows)
    c[:rows.numel()] = cnt.index_select(0, rows)


def resolve_torch(acc: torch.Tensor, cnt: torch.Tensor) -> torch.Tensor:
    """acc / max(1, count) as B,G,R,A bytes in one int32 per pixel (raytracer.comp:992-995; rgba8 store =
    clamp, x 255, round to nearest even)."""
    d = cnt.clamp(min=1).to(torch.float32).unsqueeze(-1)
    q = torch.round(torch.nan_to_num(acc / d, nan=0.0).clamp(0.0, 1.0) * 255.0).to(torch.int32)
    packed = q[..., 2] | (q[..., 1] << 8) | (q[..., 0] << 16) | (q[..., 3] << 24)
    return packed.to(torch.int32)


def unpack_rows_torch(gathered: torch.Tensor, acc: torch.Tensor, cnt: torch.Tensor, bgra: torch.Tensor, strip_rows: int,
                      world: int, skip: int) -> None:
    """rt_strip_unpack in torch: every rank's rows except `skip`'s into the frame, BGRA8 rebuilt."""
    height, width = cnt.shape
    cap = staging_rows(height, strip_rows, world)
    per = staging_bytes(height, width, strip_rows, world)
    for r in range(world):
        if r == skip:
            continue
        rows = strip_rows_of(height, strip_rows, world, r).to(acc.device)
        blk = gathered[r * per:(r + 1) * per]
        a = blk[:cap * width * 16].view(torch.float32).view(cap, width, 4)[:rows.numel()]
        c = blk[cap * width * 16:cap * width * 20].view(torch.int32).view(cap, width)[:rows.numel()]
        acc.index_copy_(0, rows, a)
        cnt.index_copy_(0, rows, c)
        bgra.index_copy_(0, rows, resolve_torch(a, c))


class PackedStripExchange:
    """The strip exchange with ONE all-gather per frame: each rank packs its rows' colours and counts into
    one staging buffer (20 bytes per pixel; the BGRA8 bytes are not sent but rebuilt from colours and
    counts on arrival), the staging buffers are gathered, and every other rank's rows are unpacked.
    With a Context (`ctx`) the packing and unpacking are the library's kernels on its stream
    (rt_strip_pack / rt_strip_unpack on the framebuffer bound to it); without one they are the torch
    restatements above (CPU rehearsals with gloo)."""

    def __init__(self, acc: torch.Tensor, cnt: torch.Tensor, bgra: torch.Tensor, strip_rows: int, world: int, rank: int,
                 ctx=None, group: Optional[dist.ProcessGroup] = None, include_self: bool = False, via_host: bool = False):
        self.acc, self.cnt, self.bgra = acc, cnt, bgra
        self.strip_rows, self.world, self.rank, self.ctx, self.group = strip_rows, world, rank, ctx, group
        self.skip = -1 if include_self else rank
        self.via_host = via_host  # device buffers, but a backend that gathers host memory (rehearsals over gloo)
        height, width = cnt.shape
        self.bytes = staging_bytes(height, width, strip_rows, world)
        self.send = torch.zeros(self.bytes, dtype=torch.uint8, device=acc.device)
        self.recv = torch.empty(self.bytes * world, dtype=torch.uint8, device=acc.device)

    def pack(self) -> None:
        """Own rows -> staging (on the context's stream when there is one)."""
        if self.ctx is not None:
            self.ctx.strip_pack(self.strip_rows, self.world, self.rank, self.send.data_ptr())
        else:
            pack_rows_torch(self.acc, self.cnt, self.strip_rows, self.world, self.rank, self.send)

    def gather(self) -> None:
        """The one collective (torch's current stream)."""
        if self.via_host:
            torch.cuda.current_stream(self.send.device).synchronize()  # (the caller made this stream wait for the packing)
            parts = [torch.empty(self.bytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, self.send.cpu(), group=self.group)
            self.recv.copy_(torch.cat(parts))
            return
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)

    def unpack(self) -> None:
        """Other ranks' rows -> frame, BGRA8 rebuilt (on the context's stream when there is one)."""
        if self.ctx is not None:
            self.ctx.strip_unpack(self.strip_rows, self.world, self.skip, self.recv.data_ptr())
        else:
            unpack_rows_torch(self.recv, self.acc, self.cnt, self.bgra, self.strip_rows, self.world, self.skip)


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


def reduce_samples(accum: torch.Tensor, counts: torch.Tensor, group: Optional[dist.ProcessGroup] = None, via_host: bool = False) -> None:
    """Sum the per-rank accumulation buffers in place on every rank.  `via_host`: device buffers, but a backend that
    reduces host memory (rehearsals over gloo)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if via_host:
            for t in (accum, counts):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
            return
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
