"""Diagnostic: where the time of a repacked batch goes (run under rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene")))
n = 1 << 22
cfg = rtk.RenderConfig(width=1920, height=1080)
cam = torch.empty((1920 * 1080, 6), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream()
acc.camera_rays_device(cfg, cam.data_ptr(), 0, st.cuda_stream)
coh = cam.repeat(3, 1)[:n].contiguous()
g = torch.Generator(device="cpu"); g.manual_seed(42)
shuf = coh[torch.randperm(n, generator=g).cuda()].contiguous()
hits = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
import numpy as np
rng = np.random.default_rng(43)
o = rng.uniform([-15, -5, -15], [15, 8.82, 15], size=(n, 3)).astype(np.float32)
v = rng.normal(size=(n, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
sec = torch.from_numpy(np.concatenate([o, v], axis=1).astype(np.float32)).cuda()
for name, rays in (("coherent", coh), ("shuffled", shuf), ("secondary", sec)):
    for mode in (2, 8, 0):
        for _ in range(2): acc.intersect_device(rays.data_ptr(), n, name != "secondary", hits.data_ptr(), mode, st.cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): acc.intersect_device(rays.data_ptr(), n, name != "secondary", hits.data_ptr(), mode, st.cuda_stream)
        torch.cuda.synchronize()
        print(name, "mode", mode, "%.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
