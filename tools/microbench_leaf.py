"""Single-wave cost of the wave-cooperative leaf loop: one 64-ray wave against a tree that is ONE leaf (max_depth=0)."""
import importlib, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
sc = rtk.parse_scene_file(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene"))
acc = rtk.KdTreeSimdAccel(sc, max_depth=0)
ntri = acc.tree_info().n_leaf_refs
rng = np.random.default_rng(1)
def run(name, rays, cull, mode, nwaves=1):
    rays = np.tile(rays, (nwaves, 1)).astype(np.float32)
    d_r = torch.from_numpy(rays).cuda(); d_h = torch.empty((rays.shape[0], 32), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): acc.intersect_device(d_r.data_ptr(), rays.shape[0], cull, d_h.data_ptr(), mode, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(10): acc.intersect_device(d_r.data_ptr(), rays.shape[0], cull, d_h.data_ptr(), mode, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    hits = (d_h.cpu().numpy().view(rtk.HIT_DTYPE)["tri"] != 0xFFFFFFFF).sum()
    print(f"{name:34s} mode {mode} waves {nwaves:5d}: {us:9.1f} us/launch  = {us*1e3/ntri:7.1f} ns/triangle-iteration ({us*2.4e3/ntri:6.0f} cyc @2.4GHz), hits {hits}")
o = np.array([0, 14, 26], np.float32)
# (a) rays from the camera towards the dragon: real mix of early outs
tgt = rng.uniform([-3, -3, -3], [3, 3, 3], size=(64, 3)).astype(np.float32)
d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
prim = np.concatenate([np.broadcast_to(o, (64, 3)), d], axis=1)
# (b) rays pointing away from everything: |det| passes (no cull) but u fails; with cull half fail at det
away = np.concatenate([np.broadcast_to(o, (64, 3)), np.broadcast_to(np.array([0, 1, 0.2], np.float32), (64, 3))], axis=1)
for mode in (2, 1):
    run("camera->dragon, cull", prim, True, mode)
    run("camera->dragon, no cull", prim, False, mode)
    run("away, cull", away, True, mode)
    run("away, no cull", away, False, mode)
run("camera->dragon, no cull", prim, False, 2, nwaves=1024)
run("camera->dragon, no cull", prim, False, 2, nwaves=4096)
run("camera->dragon, no cull", prim, False, 2, nwaves=8192)
run("camera->dragon, no cull", prim, False, 2, nwaves=16384)
print("leaf refs", ntri)
