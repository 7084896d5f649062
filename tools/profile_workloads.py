#!/usr/bin/env python3
"""One `extras` workload of bench.py, run a fixed number of times, for rocprofv3 (kernel trace or a --pmc pass).

usage (on the GPU box, after `--`):  python3 tools/profile_workloads.py <workload> [units]
workloads and their unit (the thing bench.py times, so counters per unit / bench ms give rates):
  config3                 one frame of hw11/scene8 1920x1080, 4 spp, depth 10
  config4                 one 16-sample pass of hw15/scene2 1920x1920, spp 128, depth 5, 1 diffuse ray
  config5                 one 8-sample pass of hw15/scene2 3840x2160, spp 512, depth 10, 1 diffuse ray
  synthetic_<set>         one rtk_accel_intersect_device launch of 2^24 rays, set = coherent_primary | shuffled_primary |
                          uniform_secondary, RTK_TRACE_AUTO
Warm-up units (engine trials, cost feedback, workspace growth) run first; a marker file tells the summariser how many
launches belong to them: every kernel launch is attributed by order (warm-up launches first).
Prints one JSON line: {"workload", "units", "warmup_units"}.
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

rtk = importlib.import_module("simd-raytracer_amd")
S = os.path.join(ROOT, "tests", "golden", "scenes")


def main():
    name = sys.argv[1]
    units = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    if name.startswith("synthetic_"):
        n = 1 << 24
        acc, sets = bench.synthetic_rays(rtk, torch, stream, n)
        rays, cull = next((r, c) for nm, r, c in sets if "synthetic_" + nm == name)
        hits = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
        warm = 2

        def unit(_):
            acc.intersect_device(rays.data_ptr(), n, cull, hits.data_ptr(), 0, st)
    else:
        path, kw, per, warm = {
            "config3": (f"{S}/hw11/scene8.crtscene", dict(width=1920, height=1080, spp=4, max_ray_depth=10), 0, 5),
            "config4": (f"{S}/hw15/scene2.crtscene", dict(spp=128, max_ray_depth=5, diffuse_rays=1), 16, 5),
            "config5": (f"{S}/hw15/scene2.crtscene", dict(width=3840, height=2160, spp=512, max_ray_depth=10, diffuse_rays=1), 8, 5),
        }[name]
        acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path))
        c0 = rtk.RenderConfig(**kw)
        buf = torch.zeros((acc.output_floats(c0),), dtype=torch.float32, device="cuda")

        def unit(k):
            cfg = rtk.RenderConfig(**kw, sample_begin=(k * per) % kw["spp"], sample_count=per) if per else c0
            acc.render_frame_device(cfg, buf.data_ptr(), st)
    for k in range(warm):
        unit(k)
    torch.cuda.synchronize()
    # a marker kernel between warm-up and measured units: the summariser counts launches after the LAST k_to_rgb8 launch
    mark = torch.zeros((64,), dtype=torch.float32, device="cuda")
    out8 = torch.zeros((64,), dtype=torch.uint8, device="cuda")
    rtk.frame_to_rgb8_device(mark.data_ptr(), 64, out8.data_ptr(), st)
    torch.cuda.synchronize()
    for k in range(units):
        unit(warm + k)
    torch.cuda.synchronize()
    print(json.dumps({"workload": name, "units": units, "warmup_units": warm}))


if __name__ == "__main__":
    main()
