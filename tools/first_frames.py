"""Frame times of the first frames of a shape on a fresh accelerator, per trace mode (what a one-shot CLI render pays)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
SCENE = os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene")
st = torch.cuda.current_stream()
frame = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
names = {0: "auto", 3: "group4", 7: "twopass", 4: "group8"}
for mode in [int(m) for m in os.environ.get("FF_MODES", "0 3 7").split()]:
    for rep in range(2):
        acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE))
        acc.render_frame_device(rtk.RenderConfig(width=64, height=64, trace_mode=3 if mode == 0 else mode), frame.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        cfg = rtk.RenderConfig(width=1920, height=1080, trace_mode=mode)
        ts = []
        for k in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); acc.render_frame_device(cfg, frame.data_ptr(), st.cuda_stream); b.record(st)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"{names.get(mode, mode):8s} frames 1-4: " + "  ".join(f"{t:.3f}" for t in ts) + " ms")
