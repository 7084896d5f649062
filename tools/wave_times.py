"""Diagnostic: per-wave start/end times of k_render from a -DRTK_DEBUG_WAVE_TIME build (not part of the product)."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
dbg = C.CDLL(sys.argv[1])
for name in ("rtk_render_frame",):
    getattr(dbg, name).argtypes = getattr(rtk.lib(), name).argtypes
dbg.rtk_scene_load_crtscene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
dbg.rtk_accel_build.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
sc = C.c_void_p(); dbg.rtk_scene_load_crtscene(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene").encode(), C.byref(sc))
ac = C.c_void_p(); dbg.rtk_accel_build(sc, None, C.byref(ac))
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h = 1920, 1080
world, rank = int(os.environ.get("WT_WORLD", "1")), int(os.environ.get("WT_RANK", "0"))
p = rtk.RenderConfig(width=w, height=h, trace_mode=mode, rank=rank, world_size=world).to_c()
if world > 1:                      # compact bucket buffer [buckets_per_rank][64][64][3]; blocks outside the frame stay zero
    nn = C.c_size_t(); dbg.rtk_render_output_floats.argtypes = rtk.lib().rtk_render_output_floats.argtypes
    assert dbg.rtk_render_output_floats(ac, C.byref(p), C.byref(nn)) == 0
    n = nn.value
    rgb = np.zeros((n // (64 * 3), 64, 3), np.float32)
else:
    rgb = np.zeros((h, w, 3), np.float32)
cn = rtk.Counters()
for _ in range(3):
    rc = dbg.rtk_render_frame(ac, C.byref(p), rgb.ctypes.data, C.byref(cn))
assert rc == 0
# one sample per 8x8 block (all lanes of a wave wrote the same values)
t0 = rgb[::8, ::8, 0].astype(np.float64); t1 = rgb[::8, ::8, 1].astype(np.float64); it = rgb[::8, ::8, 2]
live = t0 > 0
t0, t1, it = t0[live].reshape(1, -1), t1[live].reshape(1, -1), it[live].reshape(1, -1)
t1 = np.where(t1 < t0, t1 + 2**24, t1)
med = np.median(t0)
wrap = t0 < med - 2**23
t0 = np.where(wrap, t0 + 2**24, t0); t1 = np.where(wrap, t1 + 2**24, t1)
wrap = t0 > med + 2**23
t0 = np.where(wrap, t0 - 2**24, t0); t1 = np.where(wrap, t1 - 2**24, t1)
base = t0.min(); start = (t0 - base) / 100.0; end = (t1 - base) / 100.0; dur = end - start   # microseconds (100 MHz)
print("waves", dur.size, "kernel span us", end.max())
print("dur us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
heavy = it > 1
print("heavy waves", heavy.sum(), "mean dur %.1f" % dur[heavy].mean(), "sum of all durations (ms) %.2f" % (dur.sum() / 1e3))
print("start us percentiles of heavy:", np.percentile(start[heavy], [0, 10, 50, 90, 100]).round(1))
print("end us percentiles:", np.percentile(end, [50, 90, 99, 100]).round(1))
for lo, hi in [(0, 1), (1, 2), (2, 6), (6, 11), (11, 100)]:
    m = (it >= lo) & (it < hi)
    if m.any(): print(f"traces [{lo},{hi}): waves {m.sum()}, mean dur {dur[m].mean():.1f} us, max {dur[m].max():.1f}")
order = np.argsort(-dur.ravel())[:5]
print("longest:", [(int(i // dur.shape[1]), int(i % dur.shape[1]), round(float(dur.ravel()[i]), 1), int(it.ravel()[i])) for i in order])
# concurrency over time
ts = np.linspace(0, end.max(), 24)
print("active waves over time:", [int(((start <= t) & (end > t)).sum()) for t in ts])
