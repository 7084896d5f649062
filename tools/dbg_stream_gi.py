import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd"); import oracle as O
S2 = os.path.join(ROOT, "tests/golden/scenes/hw15/scene2.crtscene")
oa = O.Accel(O.Scene(O.load_crtscene(S2)), O.ACCEL_KD_SIMD)
for (w,h,spp,depth,diff) in [(96,96,1,4,3)]:
    ref, ocn = oa.render(w,h,spp,depth,diff)
    for fac in sys.argv[1:]:
        os.environ["RTK_STREAM_NODE_FACTOR"]=fac
        acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(S2))
        rgb, cn = acc.render_frame(rtk.RenderConfig(width=w,height=h,spp=spp,max_ray_depth=depth,diffuse_rays=diff,trace_mode=6))
        print((w,h,spp,depth,diff),"factor",fac,"rays",cn["rays"],"oracle",ocn["rays"],"maxdiff",float(np.abs(rgb-ref).max()))
