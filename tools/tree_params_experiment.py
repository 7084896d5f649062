"""Frame times of configs 2 and 3 when the tree is built with other template parameters of the reference's accelerator
(kd_tree_simd_accel<F, eps, max_depth, max_leaf_size>, kd_tree_simd.hpp:63-67; the CLI instantiates the defaults 8 / 64)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
CASES = {"cfg2": (f"{S}/hw09/scene5.crtscene", dict(width=1920, height=1080, spp=1, max_ray_depth=5)),
         "cfg3": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=4, max_ray_depth=10))}
st = torch.cuda.current_stream().cuda_stream
for cname, (path, kw) in CASES.items():
    base = None
    for (d, l) in [(8, 64), (10, 64), (12, 32), (14, 16), (16, 16)]:
        acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path), max_depth=d, max_leaf_size=l)
        ti = acc.tree_info()
        cfg = rtk.RenderConfig(**kw)
        out = torch.empty((acc.output_floats(cfg),), dtype=torch.float32, device="cuda")
        for _ in range(8): acc.render_frame_device(cfg, out.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): acc.render_frame_device(cfg, out.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        rays = acc.last_counters()["rays"]
        q = (255.999 * out.clamp(0, 1).double()).to(torch.uint8)
        if base is None: base = q.clone()
        diff = int((q.view(-1, 3) != base.view(-1, 3)).any(dim=1).sum())
        print(f"{cname} max_depth {d:2d} max_leaf {l:2d}: nodes {ti.n_nodes:6d} leaf refs {ti.n_leaf_refs:7d}  {dt*1e3:8.3f} ms  {rays/dt/1e6:8.1f} Mrays/s  8-bit pixels differing from the 8/64 tree: {diff}", flush=True)
