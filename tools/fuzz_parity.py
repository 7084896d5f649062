"""Randomised parity run (not part of the test suite: minutes of oracle time): random frame shapes / settings / engines against the CPU
oracle, bit for bit, and random ray batches through the repacking strategies against the plain wave walk.
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests/golden/scenes")
SCENES = [f"{S}/hw09/scene5.crtscene", f"{S}/hw11/scene8.crtscene", f"{S}/hw15/scene2.crtscene", f"{S}/hw12/scene4.crtscene"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pairs = {}
def pair(path):
    if path not in pairs:
        pairs[path] = (rtk.KdTreeSimdAccel(rtk.parse_scene_file(path)), oracle.Accel(oracle.Scene(oracle.load_crtscene(path)), oracle.ACCEL_KD_SIMD))
    return pairs[path]
t0 = time.time(); n_frames = n_batches = 0; last = t0
st = torch.cuda.current_stream().cuda_stream
while time.time() - t0 < budget:
    if time.time() - last > 30.0:
        last = time.time(); print(f"... {n_frames} frames, {n_batches} batches after {last - t0:.0f} s", flush=True)
    path = SCENES[rng.integers(len(SCENES))]
    acc, oacc = pair(path)
    if rng.random() < 0.6:
        w, h = int(rng.integers(17, 700)), int(rng.integers(17, 500))
        spp, depth = int(rng.integers(1, 4)), int(rng.integers(0, 8))
        diffuse = int(rng.integers(0, 3)) if rng.random() < 0.3 else 0
        mode = int(rng.choice([0, 1, 2, 3, 4, 6, 7]))
        fov = float(rng.choice([90.0, 60.0, 37.5]))
        cfg = rtk.RenderConfig(width=w, height=h, spp=spp, max_ray_depth=depth, diffuse_rays=diffuse, trace_mode=mode, fov_degrees=fov)
        try:
            got, cn = acc.render_frame(cfg)
        except rtk.RtkError as e:
            if e.code == rtk.RTK_ERR_UNSUPPORTED: continue
            raise
        ref, ocn = oacc.render(w, h, spp, depth, diffuse, fov_degrees=fov, count_work=False)
        ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and cn["rays"] == ocn["rays"]
        n_frames += 1
        if not ok:
            print("FRAME MISMATCH", path, w, h, spp, depth, diffuse, mode, fov, int((got != ref).any(axis=2).sum()), cn["rays"], ocn["rays"], flush=True)
            sys.exit(1)
    else:
        n = int(rng.integers(1 << 18, 1 << 20)) + int(rng.integers(0, 300))
        kind = rng.integers(4)
        flat = oacc.scene.flat
        lo, hi = flat.vertices.min(axis=0) - 2.0, flat.vertices.max(axis=0) + 2.0
        if kind == 0:                                                # one origin, directions everywhere
            o = np.tile(rng.uniform(lo, hi, size=(1, 3)), (n, 1)); d = rng.normal(size=(n, 3))
        elif kind == 1:                                              # everything everywhere
            o = rng.uniform(lo, hi, size=(n, 3)); d = rng.normal(size=(n, 3))
        elif kind == 2:                                              # origins on a plane, one direction (orthographic), shuffled
            o = rng.uniform(lo, hi, size=(n, 3)); o[:, 1] = hi[1]; d = np.tile([[0.1, -1.0, 0.05]], (n, 1))
        else:                                                        # camera rays of a random frame shape, shuffled or not
            w = int(rng.integers(64, 160)) * 8; h = -(-n // w)
            cam = acc.camera_rays(rtk.RenderConfig(width=w, height=h)).reshape(-1, 6)[:n]
            if rng.random() < 0.5: cam = cam[rng.permutation(n)]
            o, d = cam[:, :3], cam[:, 3:]
        rays = np.ascontiguousarray(np.concatenate([o, d], axis=1), dtype=np.float32)
        if rng.random() < 0.2: rays[rng.integers(0, n, 50), rng.integers(0, 6, 50)] = np.nan
        cull = bool(rng.integers(2))
        want = acc.intersect(rays, cull, rtk.TRACE_WAVE)
        for m in (rtk.TRACE_AUTO, 8):
            got = acc.intersect(rays, cull, m)
            if got.tobytes() != want.tobytes():
                print("BATCH MISMATCH", path, n, int(kind), cull, m, flush=True); sys.exit(1)
        k = 20000
        oh = oacc.intersect(rays[:k], cull)
        if not (np.array_equal(want["tri"][:k], oh["tri"]) and np.array_equal(want["t"][:k].view(np.uint32), oh["t"].view(np.uint32))):
            print("BATCH vs ORACLE MISMATCH", path, n, int(kind), cull, flush=True); sys.exit(1)
        n_batches += 1
print(f"fuzz ok: {n_frames} frames and {n_batches} ray batches in {time.time() - t0:.0f} s, all bit-equal", flush=True)
