"""SURVEY §8(d) fixed-size synthetic workload for the batched C-ABI entry point rtk_accel_intersect_device:
N = 2^24 rays on scene5's tree — coherent (camera pixel-centre directions tiled), incoherent (same set shuffled with a
fixed seed) and secondary (origins uniform in the scene box, directions uniform on the sphere, no culling).
5 warm-up + 20 timed launches, median; algorithmic bytes from the per-ray work counters."""
import importlib, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
scene = rtk.parse_scene_file(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene"))
acc = rtk.KdTreeSimdAccel(scene)
N = 1 << 24
W, H = 1920, 1080
arr = scene.arrays()
cam, M = arr["cam_pos"], arr["cam_mat"].reshape(3, 3)
ys, xs = np.meshgrid(np.arange(H, dtype=np.float32) + 0.5, np.arange(W, dtype=np.float32) + 0.5, indexing="ij")
sx = ((2 * xs / W) - 1) * np.float32(W / H); sy = 1 - (2 * ys / H)
d = np.stack([sx, sy, -np.ones_like(sx)], axis=-1).reshape(-1, 3) @ M          # transpose(M) * v == v @ M
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
reps = -(-N // d.shape[0])
dirs = np.tile(d, (reps, 1))[:N]
coherent = np.concatenate([np.broadcast_to(cam, (N, 3)), dirs], axis=1).astype(np.float32)
rng = np.random.default_rng(42)
incoherent = coherent[rng.permutation(N)]
rng = np.random.default_rng(43)
o = rng.uniform([-15, -5, -15], [15, 8.82, 15], size=(N, 3)).astype(np.float32)
v = rng.normal(size=(N, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
secondary = np.concatenate([o, v], axis=1).astype(np.float32)
names = {0: "auto", 1: "lane", 2: "wave"}
results = []
for wname, rays, cull in (("coherent primary", coherent, True), ("incoherent primary", incoherent, True), ("secondary", secondary, False)):
    d_r = torch.from_numpy(np.ascontiguousarray(rays)).cuda(); d_h = torch.empty((N, 32), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    cn = acc.intersect_stats(d_r.data_ptr(), N, cull, d_h.data_ptr(), 2)
    bytes_alg = 32 * cn["nodes"] + 36 * cn["tris"] + 64 * N
    for mode in (2, 0, 1):
        for _ in range(5): acc.intersect_device(d_r.data_ptr(), N, cull, d_h.data_ptr(), mode, st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); acc.intersect_device(d_r.data_ptr(), N, cull, d_h.data_ptr(), mode, st); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        r = dict(workload=wname, mode=names[mode], ms=ms, mrays_s=N / ms / 1e3, hit_fraction=cn["hits"] / N,
                 nodes_per_ray=cn["nodes"] / N, tris_per_ray=cn["tris"] / N, algorithmic_GBps=bytes_alg / ms / 1e6,
                 frac_of_8TBps=bytes_alg / ms / 1e6 / 8000)
        results.append(r)
        print(f"{wname:20s} {names[mode]:5s} {ms:9.2f} ms {r['mrays_s']:9.1f} Mrays/s  alg {r['algorithmic_GBps']:8.0f} GB/s  frac {r['frac_of_8TBps']:.3f}"
              f"  ({r['nodes_per_ray']:.1f} nodes, {r['tris_per_ray']:.1f} tris per ray, {100*r['hit_fraction']:.1f}% hit)")
    del d_r, d_h
json.dump(results, open(os.path.join(ROOT, "gpurun_out", "bench_intersect.json"), "w"), indent=1)
