"""Diagnostic (-DRTK_DEBUG_KHIST build): leaf visits of the wave walk by leaf size and number of participating rays."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
dbg = C.CDLL(sys.argv[1])
dbg.rtk_render_frame.argtypes = rtk.lib().rtk_render_frame.argtypes
dbg.rtk_scene_load_crtscene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
dbg.rtk_accel_build.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
scene = sys.argv[3] if len(sys.argv) > 3 else "hw09/scene5.crtscene"
sc = C.c_void_p(); assert dbg.rtk_scene_load_crtscene(os.path.join(ROOT, "tests/golden/scenes", scene).encode(), C.byref(sc)) == 0
ac = C.c_void_p(); assert dbg.rtk_accel_build(sc, None, C.byref(ac)) == 0
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 1920, 1080
cfg = rtk.RenderConfig(width=w, height=h, trace_mode=mode, collect_stats=True, max_ray_depth=int(os.environ.get("KH_DEPTH", "5")))
p = cfg.to_c()
rgb = np.zeros((h, w, 3), np.float32); cn = rtk.Counters()
assert dbg.rtk_render_frame(ac, C.byref(p), rgb.ctypes.data, C.byref(cn)) == 0
out = (C.c_ulonglong * 64)()
dbg.rtk_debug_khist(out)
a = np.array(out[:], dtype=np.float64)
tri = a[:32].reshape(4, 8)[:, :6]; vis = a[32:].reshape(4, 8)[:, :6]
print("rows: leaf size <12, 12-63, 64-191, >=192; cols: participating rays <=2, <=4, <=8, <=16, <=32, <=64")
print("triangle steps (sum of leaf sizes), % of total:")
print(np.round(100 * tri / tri.sum(), 1))
print("leaf visits, % of total:")
print(np.round(100 * vis / vis.sum(), 1))
print("total wave triangle steps %.3e, leaf visits %.3e" % (tri.sum(), vis.sum()))
