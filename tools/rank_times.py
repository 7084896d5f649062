"""What each rank of an N-GPU run would spend in its render kernel, measured on ONE GPU: renders rank r of world N
(bucket i -> rank i % N) for every r and reports the slowest rank (the one the frame waits for)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
# default: BASELINE config 2; TC_SCENE / TC_W / TC_H / TC_SPP / TC_DEPTH / TC_GI select another frame (e.g. config 5's shape)
scene_rel = os.environ.get("TC_SCENE", "hw09/scene5.crtscene")
W, H = int(os.environ.get("TC_W", "1920")), int(os.environ.get("TC_H", "1080"))
SPP, DEPTH, GI = int(os.environ.get("TC_SPP", "1")), int(os.environ.get("TC_DEPTH", "5")), int(os.environ.get("TC_GI", "0"))
N_TIMED = int(os.environ.get("TC_FRAMES", "40"))
acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(f"{S}/{scene_rel}"))
modes = [int(m) for m in os.environ.get("TC_MODES", "0 7").split()]
st = torch.cuda.current_stream()
base = {}
for world in [int(w) for w in os.environ.get("TC_WORLDS", "1 2 4 8").split()]:
    for mode in modes:
        times, rays = [], []
        for rank in range(world):
            cfg = rtk.RenderConfig(width=W, height=H, spp=SPP, max_ray_depth=DEPTH, diffuse_rays=GI, trace_mode=mode, rank=rank, world_size=world)
            out = torch.empty((acc.output_floats(cfg),), dtype=torch.float32, device="cuda")
            for _ in range(5): acc.render_frame_device(cfg, out.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            n = N_TIMED
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(n): acc.render_frame_device(cfg, out.data_ptr(), st.cuda_stream)
            b.record(st)
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / n)
            rays.append(acc.last_counters()["rays"])
        if world == 1: base[mode] = max(times)
        print(f"world {world} mode {mode}: slowest rank {max(times):.3f} ms, fastest {min(times):.3f} ms, rays/rank {min(rays)}..{max(rays)}, "
              f"kernel-only scaling {base.get(mode, max(times)) / max(times):.2f}x", flush=True)
