"""RTK_TRAVERSAL_FAST beside the parity mode on the streaming pipeline: frame time and how many pixels differ (floats / 8-bit).
RTK_FAST_OCCLUDERS=0 keeps is_occluded's stepping loop in the fast mode."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
CASES = {"cfg3 scene8 1080p spp4 d10": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=4, max_ray_depth=10)),
         "scene8 1080p spp1 d5 (the reference's refractive_dragon.png)": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=1, max_ray_depth=5)),
         "cfg4 shape hw15s2 960 spp8 gi1": (f"{S}/hw15/scene2.crtscene", dict(width=960, height=960, spp=8, max_ray_depth=5, diffuse_rays=1))}
stream = torch.cuda.current_stream()
q = lambda t: (255.999 * t.clamp(0, 1).double()).to(torch.uint8)
for name, (path, kw) in CASES.items():
    res = {}
    for mode, trav in (("parity", rtk.TRAVERSAL_REFERENCE), ("fast", rtk.TRAVERSAL_FAST)):
        a = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path), traversal=trav)
        c = rtk.RenderConfig(trace_mode=6, **kw)
        b = torch.empty((kw["height"], kw["width"], 3), dtype=torch.float32, device="cuda")
        for _ in range(3): a.render_frame_device(c, b.data_ptr(), stream.cuda_stream)
        ms = min(bench.event_ms(torch, stream, lambda: a.render_frame_device(c, b.data_ptr(), stream.cuda_stream), 1) for _ in range(3))
        res[mode] = (ms, a.last_counters()["rays"], b)
    d = res["fast"][2] - res["parity"][2]
    print(f"{name}: parity {res['parity'][0]:.2f} ms ({res['parity'][1]} rays)  fast {res['fast'][0]:.2f} ms ({res['fast'][1]} rays)  "
          f"pixels differing {int((d != 0).any(dim=2).sum())}, in 8 bit {int((q(res['fast'][2]) != q(res['parity'][2])).any(dim=2).sum())}, "
          f"max |diff| {float(d.abs().max()):.4f}", flush=True)
