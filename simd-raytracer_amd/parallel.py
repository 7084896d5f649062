"""Multi-GPU frame sharding: one process per GPU, buckets dealt round-robin, one all-gather over RCCL/xGMI.

The reference's only parallelism is threads pulling bucket tiles from a mutex queue
(render/tile/bucket.hpp:7-21, render/tile/queue.hpp:30-41, render/render.hpp:93-101).  Here bucket i belongs to
rank i % world (interleaved, because coverage is very uneven: most of scene5 is background; diagonally, bucket (bx, by) ->
rank (bx + by) % world, when a row of buckets is a whole number of rounds -- BucketLayout.skew_q); every rank renders
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

    @property
    def skew_q(self) -> int:
        """csrc/kernels.hpp rank_bucket(): when a row of buckets is a whole number of rounds (tiles_x % world == 0) round robin
        would give every rank the same columns of every row; those frames are dealt diagonally, bucket (bx, by) -> rank
        (bx + by) % world, tiles_x / world buckets per rank and row.  0 = plain round robin (bucket i -> rank i % world)."""
        return self.tiles_x // self.world if (self.world > 1 and self.tiles_x % self.world == 0) else 0

    def buckets_of(self, rank: int) -> list:
        """The buckets (row-major indices) of `rank`, in the order of its compact buffer."""
        q = self.skew_q
        if q == 0:
            return list(range(rank, self.n_buckets, self.world))
        out = []
        for local in range(self.buckets_per_rank):
            by, m = divmod(local, q)
            out.append(by * self.tiles_x + (rank - by) % self.world + m * self.world)
        return out

    def pixel_sources(self) -> torch.Tensor:
        """For every frame pixel, its index in the gathered [world, buckets_per_rank, bucket, bucket] array."""
        ys = torch.arange(self.height).view(-1, 1)
        xs = torch.arange(self.width).view(1, -1)
        bx, by = xs // self.bucket, ys // self.bucket
        q = self.skew_q
        if q == 0:
            b = by * self.tiles_x + bx
            rank, local = b % self.world, b // self.world
        else:
            rank, local = (bx + by) % self.world, by * q + bx // self.world
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


def ray_range(n: int, rank: int, world: int) -> tuple:
    """A batch of n independent rays cut into contiguous ranges, one per rank: [lo, hi) of `rank`.  Rank order == ray order, so an
    all-gather of the ranks' hit arrays IS the batch's hit array -- when the ranges are equally long (n % world == 0), which is what
    all_gather_into_tensor needs; otherwise gather_hits pads."""
    return rank * n // world, (rank + 1) * n // world


def gather_hits(mine: torch.Tensor, n: int, rank: int, world: int, out: torch.Tensor | None = None, group=None) -> torch.Tensor:
    """All-gather of the ranks' hit records ([hi - lo, 32] uint8 each, the ranges of ray_range) into the batch's [n, 32] array."""
    import torch.distributed as dist

    if out is None:
        out = torch.empty((n, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    if world == 1:
        out.copy_(mine)
        return out
    if n % world == 0:
        dist.all_gather_into_tensor(out.view(-1), mine.reshape(-1), group=group)
        return out
    longest = -(-n // world)                                   # ragged ranges: equal-length padded pieces, cut back afterwards
    padded = torch.zeros((longest, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    padded[: mine.shape[0]] = mine
    pieces = torch.empty((world, longest, mine.shape[1]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(pieces.view(-1), padded.view(-1), group=group)
    for r in range(world):
        lo, hi = ray_range(n, r, world)
        out[lo:hi] = pieces[r, : hi - lo]
    return out


def render_sharded(accel, cfg, local: torch.Tensor, frame: torch.Tensor, gathered: torch.Tensor, group=None) -> torch.Tensor:
    """One sharded frame on the current CUDA stream: render this rank's buckets, all-gather, assemble."""
    stream = torch.cuda.current_stream().cuda_stream
    accel.render_frame_device(cfg, local.data_ptr(), stream)
    if cfg.world_size <= 1:
        return local.view(frame.shape)
    layout = BucketLayout(frame.shape[1], frame.shape[0], accel.scene.info.bucket_size, cfg.world_size)
    return gather_frame(local, layout, accel, cfg, frame, gathered, group)


class FramePipeline:
    """A sequence of sharded frames with the all-gather of frame k overlapped with the rendering of frame k+1.

    At config-2 sizes a sharded frame is ~0.1 ms of rendering per GPU and the 25 MB all-gather costs about as much,
    so running them back to back halves the frame rate.  Here every frame owns one of `depth` buffer sets
    (rank-local buckets, gathered buffer, assembled frame); `submit()` enqueues render(k) on the compute stream and
    all_gather(k) as an asynchronous collective on RCCL's stream, then retires frame k-depth+1 (waits for ITS
    gather on the compute stream, assembles it).  Hazards: a buffer set is reused only after its assemble was
    enqueued on the compute stream, which is ordered after the wait on its gather, which read the local buffer.

    `render(local, k)` and `assemble(gathered, frame)` default to the accel's device entry points; tests pass
    host callables so that the same ordering logic runs over gloo on CPU tensors.
    """

    def __init__(self, layout: BucketLayout, accel=None, cfg=None, depth: int = 2, group=None, device="cuda",
                 render=None, assemble=None, floats_per_rank: int | None = None):
        import collections

        if depth < 1:
            raise ValueError("depth must be >= 1")
        if (render is None or assemble is None) and (accel is None or cfg is None):
            raise ValueError("pass accel and cfg, or render and assemble callables")
        self.layout, self.accel, self.cfg, self.depth, self.group = layout, accel, cfg, depth, group
        n = layout.floats_per_rank if floats_per_rank is None else floats_per_rank
        self.local = [torch.zeros((n,), dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((layout.world * n,), dtype=torch.float32, device=device) for _ in range(depth)]
        self.frames = [torch.empty((layout.height, layout.width, 3), dtype=torch.float32, device=device) for _ in range(depth)]
        self._render = render or self._render_device
        self._assemble = assemble or self._assemble_device
        self._pending = collections.deque()
        self._last = None
        self.submitted = 0

    def _render_device(self, local: torch.Tensor, k: int) -> None:
        self.accel.render_frame_device(self.cfg, local.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def _assemble_device(self, gathered: torch.Tensor, frame: torch.Tensor) -> None:
        self.accel.assemble_device(self.cfg, gathered.data_ptr(), frame.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def submit(self):
        """Starts frame k = self.submitted; returns the frame retired by this call ((k_retired, tensor)) or None."""
        import torch.distributed as dist

        k = self.submitted
        slot = k % self.depth
        self.submitted += 1
        self._render(self.local[slot], k)
        work = dist.all_gather_into_tensor(self.gathered[slot], self.local[slot], group=self.group, async_op=True)
        self._pending.append((k, slot, work))
        return self._retire() if len(self._pending) >= self.depth else None

    def _retire(self):
        k, slot, work = self._pending.popleft()
        work.wait()                       # device tensors: the compute stream waits, the host does not
        self._assemble(self.gathered[slot], self.frames[slot])
        self._last = self.frames[slot]
        return k, self.frames[slot]

    def last_frame(self) -> torch.Tensor:
        """The frame retired last (assembled on this rank)."""
        return self._last

    def drain(self):
        """Retires every frame still in flight, oldest first."""
        out = []
        while self._pending:
            out.append(self._retire())
        return out
