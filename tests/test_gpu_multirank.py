"""Two ranks sharing the one GPU of the test box: each renders its buckets through the C-ABI (rank / world_size),
the bucket buffers are all-gathered (gloo over CPU staging here — RCCL needs one GPU per rank) and assembled by
k_assemble on the device.  Covers the rank plumbing end to end except the RCCL transport itself."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, SCENE2, SCENE5

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scene, w, h, depth, diffuse, spp, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rtk = importlib.import_module("simd-raytracer_amd")
        acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(scene), device=0)
        cfg = rtk.RenderConfig(width=w, height=h, max_ray_depth=depth, diffuse_rays=diffuse, spp=spp, rank=rank, world_size=world)
        n = acc.output_floats(cfg)
        local = torch.empty((n,), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        acc.render_frame_device(cfg, local.data_ptr(), st)
        rays = acc.last_counters()["rays"]
        gathered_cpu = torch.empty((world * n,), dtype=torch.float32)
        dist.all_gather_into_tensor(gathered_cpu, local.cpu())
        gathered = gathered_cpu.cuda()
        frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        acc.assemble_device(cfg, gathered.data_ptr(), frame.data_ptr(), st)
        torch.cuda.synchronize()
        t = torch.tensor([float(rays)], dtype=torch.float64)
        dist.all_reduce(t)
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.cpu().numpy())
        np.save(os.path.join(out_dir, f"rays_{rank}.npy"), np.array([t.item()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene,w,h,depth,diffuse,spp", [(SCENE5, 640, 360, 5, 0, 1), (SCENE2, 192, 192, 5, 1, 2)])
def test_two_ranks_render_one_frame(ora, tmp_path, scene, w, h, depth, diffuse, spp):
    world = 2
    ref, ocn = ora.Accel(ora.Scene(ora.load_crtscene(scene)), ora.ACCEL_KD_SIMD).render(w, h, spp, depth, diffuse)
    mp.spawn(_worker, args=(world, _free_port(), scene, w, h, depth, diffuse, spp, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        frame = np.load(tmp_path / f"frame_{r}.npy")
        assert np.array_equal(frame.view(np.uint32), ref.view(np.uint32)), f"rank {r}"
        assert int(np.load(tmp_path / f"rays_{r}.npy")[0]) == ocn["rays"]
