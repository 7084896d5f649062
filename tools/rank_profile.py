"""Diagnostic: one rank of a sharded frame under rocprofv3 --kernel-trace --stats (TC_* as in rank_times.py, TC_RANK, TC_WORLD)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(f"{S}/{os.environ.get('TC_SCENE', 'hw15/scene2.crtscene')}"))
W, H = int(os.environ.get("TC_W", "1920")), int(os.environ.get("TC_H", "1920"))
cfg = rtk.RenderConfig(width=W, height=H, spp=int(os.environ.get("TC_SPP", "8")), max_ray_depth=int(os.environ.get("TC_DEPTH", "5")),
                       diffuse_rays=int(os.environ.get("TC_GI", "1")), rank=int(os.environ.get("TC_RANK", "0")), world_size=int(os.environ.get("TC_WORLD", "8")))
out = torch.empty((acc.output_floats(cfg),), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(4): acc.render_frame_device(cfg, out.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n): acc.render_frame_device(cfg, out.data_ptr(), st)
torch.cuda.synchronize()
print("frame %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
