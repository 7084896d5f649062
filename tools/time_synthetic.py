"""The three 2^24-ray sets of bench.py's `extras.synthetic_2p24` through rtk_accel_intersect_device, per strategy (best of 5)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
rtk = importlib.import_module("simd-raytracer_amd")
stream = torch.cuda.current_stream()
n = 1 << 24
acc, sets = bench.synthetic_rays(rtk, torch, stream, n)
hits = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
only = sys.argv[1:]
for name, rays, cull in sets:
    if only and not any(o in name for o in only): continue
    for mode in [int(m) for m in os.environ.get("TS_MODES", "2 0").split()]:
        f = lambda: acc.intersect_device(rays.data_ptr(), n, cull, hits.data_ptr(), mode, stream.cuda_stream)
        f(); f()
        ms = min(bench.event_ms(torch, stream, f, 1) for _ in range(5))
        print(f"{name:20s} mode {mode}: {ms:8.3f} ms  {n / ms / 1e3:9.1f} Mrays/s", flush=True)
