"""What a sorted batch costs by coherence alone: the 2^24 camera rays laid out in memory cell by cell (cells of c x c pixels in
Morton-ish row order of cells, rays of a cell in random order), walked in place by the wave walk (no index gather / scatter)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
rtk = importlib.import_module("simd-raytracer_amd")
stream = torch.cuda.current_stream()
n = 1 << 24
acc, sets = bench.synthetic_rays(rtk, torch, stream, n)
coherent = sets[0][1]
hits = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
W, H = bench.WIDTH, bench.HEIGHT
idx = torch.arange(n, device="cuda")
pix = idx % (W * H)
y, x = pix // W, pix % W
g = torch.Generator(device="cuda"); g.manual_seed(1)
for c in (8, 16, 32, 64):
    cell = (y // c) * ((W + c - 1) // c) + (x // c)
    key = cell.to(torch.float64) + torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 0.999
    order = torch.argsort(key)
    rays = coherent[order].contiguous()
    for mode in (2,):
        f = lambda: acc.intersect_device(rays.data_ptr(), n, True, hits.data_ptr(), mode, stream.cuda_stream)
        f(); f()
        ms = min(bench.event_ms(torch, stream, f, 1) for _ in range(5))
        print(f"cells {c:3d}x{c:<3d} random inside, in place, mode {mode}: {ms:8.3f} ms  {n / ms / 1e3:9.1f} Mrays/s", flush=True)
    del rays, order, key
