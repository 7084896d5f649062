"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run the product's bucket layout + all-gather + assembly.
The rank-local buffers are cut from an oracle frame (no GPU here); the GPU box runs the same code with RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENE2, SCENE5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frame_np, bucket, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = importlib.import_module("simd-raytracer_amd.parallel")
        frame = torch.from_numpy(frame_np)
        h, w, _ = frame.shape
        layout = par.BucketLayout(w, h, bucket, world)
        local = par.extract_rank_buckets(frame, layout, rank)
        assert local.numel() == layout.floats_per_rank
        out = par.gather_frame(local, layout)
        ok = torch.equal(out.view(torch.int32), frame.view(torch.int32))
        np.save(os.path.join(result_dir, f"ok_{rank}.npy"), np.array([ok, local.numel()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene,w,h,world", [(SCENE5, 203, 117, 2), (SCENE2, 100, 100, 3),
                                             (SCENE2, 96, 60, 2)])     # 96 / 24 = 4 buckets per row on 2 ranks: the diagonal deal
def test_sharded_gather_reassembles_the_frame(ora, tmp_path, scene, w, h, world):
    flat = ora.load_crtscene(scene)
    frame, _ = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD).render(w, h, 1, 5, 0, n_threads=2)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, frame, flat.bucket_size, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, n = np.load(tmp_path / f"ok_{r}.npy")
        assert ok == 1, f"rank {r} assembled a different frame"


def test_layout_matches_the_c_abi(rtk):
    """BucketLayout (host python) and rtk_render_output_floats (C-ABI) agree on the rank-local buffer size."""
    par = importlib.import_module("simd-raytracer_amd.parallel")
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE2))
    for (w, h, world) in [(1920, 1920, 8), (100, 37, 3), (24, 24, 2), (3840, 2160, 8), (96, 60, 2), (48, 48, 2)]:
        lay = par.BucketLayout(w, h, acc.scene.info.bucket_size, world)
        assert (lay.skew_q != 0) == (world > 1 and lay.tiles_x % world == 0)
        for r in range(world):
            n = acc.output_floats(rtk.RenderConfig(width=w, height=h, rank=r, world_size=world))
            assert n == lay.floats_per_rank
        covered = sorted(b for r in range(world) for b in lay.buckets_of(r))
        assert covered == list(range(lay.n_buckets))
    assert acc.output_floats(rtk.RenderConfig(width=50, height=20)) == 50 * 20 * 3


def test_diagonal_deal_spreads_every_rank_over_all_columns():
    """Config 5's frame: 3840 / 24 = 160 buckets per row on 8 ranks.  Round robin would give a rank the same 20 columns in every
    row; the diagonal deal (bucket (bx, by) -> rank (bx + by) % 8) gives it every column, each equally often, and every rank
    exactly n_buckets / world buckets.  pixel_sources is the matching gather map: a bijection onto the gathered array."""
    par = importlib.import_module("simd-raytracer_amd.parallel")
    lay = par.BucketLayout(3840, 2160, 24, 8)
    assert lay.tiles_x == 160 and lay.skew_q == 20 and lay.buckets_per_rank * 8 == lay.n_buckets
    for r in range(8):
        cols = np.bincount([b % lay.tiles_x for b in lay.buckets_of(r)], minlength=lay.tiles_x)
        assert cols.min() >= lay.tiles_y // 8 and cols.max() <= -(-lay.tiles_y // 8)
    small = par.BucketLayout(96, 72, 24, 2)
    src = small.pixel_sources().reshape(-1).numpy()
    assert len(np.unique(src)) == 96 * 72 and src.max() < 2 * small.buckets_per_rank * 24 * 24
    frame = torch.arange(72 * 96 * 3, dtype=torch.float32).reshape(72, 96, 3)
    gathered = torch.cat([par.extract_rank_buckets(frame, small, r) for r in range(2)])
    assert torch.equal(par.assemble_host(gathered, small), frame)
    # 1920x1080 / 64 = 30 buckets per row on 8 ranks is not a whole number of rounds: plain round robin stays
    assert par.BucketLayout(1920, 1080, 64, 8).skew_q == 0


def _hits_worker(rank, world, port, n, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = importlib.import_module("simd-raytracer_amd.parallel")
        whole = (torch.arange(n * 32, dtype=torch.int64) * 2654435761 % 251).to(torch.uint8).view(n, 32)   # stands in for the hit records
        lo, hi = par.ray_range(n, rank, world)
        got = par.gather_hits(whole[lo:hi].clone(), n, rank, world)
        np.save(os.path.join(result_dir, f"hits_{rank}.npy"), np.array([torch.equal(got, whole), hi - lo]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(1 << 12, 2), (1000, 3)])
def test_sharded_ray_batch_gathers_into_ray_order(tmp_path, n, world):
    """bench.py's synthetic workload at N > 1: contiguous ray ranges per rank, hits all-gathered (rank order == ray order); the
    ragged case pads.  Same code path as on the GPUs (RCCL there, gloo here)."""
    par = importlib.import_module("simd-raytracer_amd.parallel")
    covered = [par.ray_range(n, r, world) for r in range(world)]
    assert covered[0][0] == 0 and covered[-1][1] == n and all(covered[r][1] == covered[r + 1][0] for r in range(world - 1))
    port = _free_port()
    mp.spawn(_hits_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, m = np.load(tmp_path / f"hits_{r}.npy")
        assert ok == 1 and m == covered[r][1] - covered[r][0]


def _pipeline_worker(rank, world, port, frame_np, bucket, depth, n_frames, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = importlib.import_module("simd-raytracer_amd.parallel")
        base = torch.from_numpy(frame_np)
        h, w, _ = base.shape
        layout = par.BucketLayout(w, h, bucket, world)

        def frame_k(k):                                     # a frame sequence: frame k = base + k
            return base + float(k)

        def render(local, k):                               # stands in for rtk_render_frame_device of this rank's buckets
            local.copy_(par.extract_rank_buckets(frame_k(k), layout, rank))

        def assemble(gathered, frame):                      # CPU mirror of k_assemble
            frame.copy_(par.assemble_host(gathered, layout))

        pipe = par.FramePipeline(layout, depth=depth, device="cpu", render=render, assemble=assemble)
        done = []
        for _ in range(n_frames):
            r = pipe.submit()
            if r is not None:
                done.append((r[0], r[1].clone()))
        in_flight = n_frames - len(done)
        done += [(k, f.clone()) for k, f in pipe.drain()]
        ok = [k for k, _ in done] == list(range(n_frames)) and in_flight == min(depth - 1, n_frames)
        ok = ok and all(torch.equal(f, frame_k(k)) for k, f in done)
        np.save(os.path.join(result_dir, f"pipe_{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth,n_frames", [(1, 3), (2, 5), (3, 2)])
def test_frame_pipeline_retires_every_frame_in_order(ora, tmp_path, depth, n_frames):
    """bench.py's N>1 step: gather(k) is asynchronous and overlaps render(k+1); every frame must still come out,
    in order, bit-identical, on every rank, and `depth-1` frames are in flight until the drain."""
    flat = ora.load_crtscene(SCENE5)
    frame, _ = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD).render(150, 70, 1, 5, 0, n_threads=2)
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(2, port, frame, flat.bucket_size, depth, n_frames, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(tmp_path / f"pipe_{r}.npy")[0] == 1, f"rank {r}"
