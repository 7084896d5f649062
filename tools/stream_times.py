"""Diagnostic: per-work-unit timing of the streaming pipeline from a -DRTK_DEBUG_WAVE_TIME build (prints to stderr)."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
dbg = C.CDLL(sys.argv[1])
dbg.rtk_render_frame.argtypes = rtk.lib().rtk_render_frame.argtypes
dbg.rtk_scene_load_crtscene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
dbg.rtk_accel_build.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
sc = C.c_void_p(); dbg.rtk_scene_load_crtscene(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene").encode(), C.byref(sc))
ac = C.c_void_p(); dbg.rtk_accel_build(sc, None, C.byref(ac))
p = rtk.RenderConfig(width=1920, height=1080, trace_mode=6).to_c()
rgb = np.zeros((1080, 1920, 3), np.float32); cn = rtk.Counters()
for i in range(2):
    print("frame", i, file=sys.stderr)
    assert dbg.rtk_render_frame(ac, C.byref(p), rgb.ctypes.data, C.byref(cn)) == 0
