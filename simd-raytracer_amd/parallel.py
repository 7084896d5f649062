"""Multi-GPU frame sharding: one process per GPU, buckets dealt round-robin, one all-gather over RCCL/xGMI.

The reference's only parallelism is threads pulling bucket tiles from a mutex queue
(render/tile/bucket.hpp:7-21, render/tile/queue.hpp:30-41, render/render.hpp:93-101).  Here bucket i belongs to
rank i % world (interleaved, because coverage is very uneven: most of scene5 is background); every rank renders
its buckets into a compact [buckets_per_rank, bucket, bucket, 3] buffer of equal length, the buffers are
all-gathered, and every rank assembles the frame.  No other collective is needed: pixels are independent and
the RNG is keyed by absolute pixel, so the frame does not depend on the number of ranks.

torch / torch.distributed are plumbing here (buffers, streams, RCCL).  The CPU branch of `gather_frame` only
re-orders bytes (it mirrors the k_assemble kernel) so that the collective path can be tested with gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class BucketLayout:
    width: int
    height: int
    bucket: int
    world: int

    @property
    def tiles_x(self) -> int:
        return (self.width + self.bucket - 1) // self.bucket

    @property
    def tiles_y(self) -> int:
        return (self.height + self.bucket - 1) // self.bucket

    @property
    def n_buckets(self) -> int:
        return self.tiles_x * self.tiles_y

    @property
    def buckets_per_rank(self) -> int:
        return (self.n_buckets + self.world - 1) // self.world

    @property
    def floats_per_rank(self) -> int:
        return self.buckets_per_rank * self.bucket * self.bucket * 3

    def buckets_of(self, rank: int) -> range:
        return range(rank, self.n_buckets, self.world)

    def pixel_sources(self) -> torch.Tensor:
        """For every frame pixel, its index in the gathered [world, buckets_per_rank, bucket, bucket] array."""
        ys = torch.arange(self.height).view(-1, 1)
        xs = torch.arange(self.width).view(1, -1)
        b = (ys // self.bucket) * self.tiles_x + (xs // self.bucket)
        rank, local = b % self.world, b // self.world
        return ((rank * self.buckets_per_rank + local) * self.bucket + ys % self.bucket) * self.bucket + xs % self.bucket


def extract_rank_buckets(frame: torch.Tensor, layout: BucketLayout, rank: int) -> torch.Tensor:
    """The compact buffer rank `rank` would produce for `frame` ([h, w, 3]); padding stays zero."""
    out = torch.zeros((layout.buckets_per_rank, layout.bucket, layout.bucket, 3), dtype=frame.dtype, device=frame.device)
    for j, b in enumerate(layout.buckets_of(rank)):
        x0, y0 = (b % layout.tiles_x) * layout.bucket, (b // layout.tiles_x) * layout.bucket
        tile = frame[y0:y0 + layout.bucket, x0:x0 + layout.bucket]
        out[j, : tile.shape[0], : tile.shape[1]] = tile
    return out.view(-1)


def assemble_host(gathered: torch.Tensor, layout: BucketLayout) -> torch.Tensor:
    """[world * floats_per_rank] -> [h, w, 3]; byte re-ordering only (CPU mirror of k_assemble)."""
    flat = gathered.reshape(-1, 3)
    return flat[layout.pixel_sources().reshape(-1)].reshape(layout.height, layout.width, 3)


def gather_frame(local: torch.Tensor, layout: BucketLayout, accel=None, cfg=None, frame: torch.Tensor | None = None,
                 gathered: torch.Tensor | None = None, group=None) -> torch.Tensor:
    """All-gather the rank-local bucket buffers and assemble the frame on every rank."""
    import torch.distributed as dist

    if gathered is None:
        gathered = torch.empty((layout.world * layout.floats_per_rank,), dtype=local.dtype, device=local.device)
    if layout.world > 1:
        dist.all_gather_into_tensor(gathered, local, group=group)
    else:
        gathered.copy_(local)
    if local.is_cuda:
        if accel is None or cfg is None:
            raise ValueError("device buffers are assembled by the k_assemble kernel: pass accel and cfg")
        if frame is None:
            frame = torch.empty((layout.height, layout.width, 3), dtype=local.dtype, device=local.device)
        accel.assemble_device(cfg, gathered.data_ptr(), frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return frame
    return assemble_host(gathered, layout)


def render_sharded(accel, cfg, local: torch.Tensor, frame: torch.Tensor, gathered: torch.Tensor, group=None) -> torch.Tensor:
    """One sharded frame on the current CUDA stream: render this rank's buckets, all-gather, assemble."""
    stream = torch.cuda.current_stream().cuda_stream
    accel.render_frame_device(cfg, local.data_ptr(), stream)
    if cfg.world_size <= 1:
        return local.view(frame.shape)
    layout = BucketLayout(frame.shape[1], frame.shape[0], accel.scene.info.bucket_size, cfg.world_size)
    return gather_frame(local, layout, accel, cfg, frame, gathered, group)
