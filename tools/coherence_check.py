"""How much does ray ordering matter for the wave-cooperative walk?  Random secondary-like rays, unsorted vs sorted by
(origin cell, direction octant) at several grid resolutions."""
import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(os.path.join(ROOT, "tests/golden/scenes/hw11/scene8.crtscene")))
rng = np.random.default_rng(43)
N = 1 << 20
lo, hi = np.array([-8, -5, -6], np.float32), np.array([8, 6, 6], np.float32)      # around the dragon
o = rng.uniform(lo, hi, size=(N, 3)).astype(np.float32)
d = rng.normal(size=(N, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d], axis=1)
def run(name, r, mode):
    d_r = torch.from_numpy(np.ascontiguousarray(r)).cuda(); d_h = torch.empty((r.shape[0], 32), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    acc.intersect_device(d_r.data_ptr(), r.shape[0], False, d_h.data_ptr(), mode, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): acc.intersect_device(d_r.data_ptr(), r.shape[0], False, d_h.data_ptr(), mode, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name:40s} mode {mode}: {ms:8.2f} ms  {r.shape[0]/ms/1e3:8.1f} Mrays/s")
for mode in (2, 1, 0):
    run("random order", rays, mode)
octant = ((d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2))
for g in (4, 8, 16, 32):
    cell = np.clip(((o - lo) / (hi - lo) * g).astype(np.int64), 0, g - 1)
    ckey = (cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2]
    order = np.argsort(ckey * 8 + octant, kind="stable")
    run(f"sorted cell {g}^3 then octant", rays[order], 2)
    order = np.argsort(octant * g**3 + ckey, kind="stable")
    run(f"sorted octant then cell {g}^3", rays[order], 2)
# finer direction key: 6 bits (quantised direction on a cube map face grid)
g = 16
cell = np.clip(((o - lo) / (hi - lo) * g).astype(np.int64), 0, g - 1)
ckey = (cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2]
dq = np.clip(((d + 1) * 2).astype(np.int64), 0, 3); dkey = (dq[:, 0] * 4 + dq[:, 1]) * 4 + dq[:, 2]
run("sorted cell 16^3 then dir 4^3", rays[np.argsort(ckey * 64 + dkey, kind="stable")], 2)
run("sorted dir 4^3 then cell 16^3", rays[np.argsort(dkey * g**3 + ckey, kind="stable")], 2)
