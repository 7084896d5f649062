"""Times the frame modes on the BASELINE config shapes (device-resident output, no CPU work)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
CASES = {
    "cfg2 scene5 1080p spp1 d5": (f"{S}/hw09/scene5.crtscene", dict(width=1920, height=1080, spp=1, max_ray_depth=5)),
    "cfg3 scene8 1080p spp1 d10": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=1, max_ray_depth=10)),
    "cfg3 scene8 1080p spp4 d10": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=4, max_ray_depth=10)),
    "cfg4 hw15s2 960 spp8 d5 gi1": (f"{S}/hw15/scene2.crtscene", dict(width=960, height=960, spp=8, max_ray_depth=5, diffuse_rays=1)),
    "cfg4f hw15s2 1920 spp16 d5 gi1": (f"{S}/hw15/scene2.crtscene", dict(width=1920, height=1920, spp=16, max_ray_depth=5, diffuse_rays=1)),
    "cfg5f hw15s2 4K spp8 d10 gi1": (f"{S}/hw15/scene2.crtscene", dict(width=3840, height=2160, spp=8, max_ray_depth=10, diffuse_rays=1)),
    "cfg5 hw15s2 4K spp4 d10 gi1": (f"{S}/hw15/scene2.crtscene", dict(width=3840, height=2160, spp=4, max_ray_depth=10, diffuse_rays=1)),
    "hw15s2 1920 spp1 d5": (f"{S}/hw15/scene2.crtscene", dict(width=1920, height=1920, spp=1, max_ray_depth=5)),
}
names = {0: "auto", 1: "lane", 2: "wave", 3: "group4", 6: "stream", 7: "twopass"}
only = sys.argv[1:] 
for cname, (path, kw) in CASES.items():
    if only and not any(o in cname for o in only): continue
    time.sleep(float(os.environ.get("TC_SLEEP", "0")))
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path))
    for mode in [int(m) for m in os.environ.get("TC_MODES", "0 1 2 3 6 7").split()]:
        cfg = rtk.RenderConfig(trace_mode=mode, **kw)
        out = torch.empty((acc.output_floats(cfg),), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        try:
            acc.render_frame_device(cfg, out.data_ptr(), st)
        except rtk.RtkError as e:
            print(f"{cname:32s} {names[mode]:8s} n/a ({e.code})"); continue
        for _ in range(int(os.environ.get("TC_WARM", "7"))): acc.render_frame_device(cfg, out.data_ptr(), st)   # cost feedback / engine trials settle on the first frames
        torch.cuda.synchronize()
        n = int(os.environ.get("TC_REPS", "8"))
        t0 = time.perf_counter()
        for _ in range(n): acc.render_frame_device(cfg, out.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        rays = acc.last_counters()["rays"]
        print(f"{cname:32s} {names[mode]:8s} {dt*1e3:9.3f} ms  {rays/dt/1e6:9.1f} Mrays/s  ({rays} rays)")
