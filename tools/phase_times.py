"""Diagnostic (-DRTK_DEBUG_PHASES build): where an owner wave's cycles go (node walk / small leaves / sliced leaves / shading)."""
import ctypes as C, importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtk = importlib.import_module("simd-raytracer_amd")
dbg = C.CDLL(sys.argv[1])
dbg.rtk_render_frame.argtypes = rtk.lib().rtk_render_frame.argtypes
dbg.rtk_scene_load_crtscene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
dbg.rtk_accel_build.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
sc = C.c_void_p(); assert dbg.rtk_scene_load_crtscene(os.path.join(ROOT, "tests/golden/scenes/hw09/scene5.crtscene").encode(), C.byref(sc)) == 0
ac = C.c_void_p(); assert dbg.rtk_accel_build(sc, None, C.byref(ac)) == 0
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 1920, 1080
world, rank = int(os.environ.get("TC_WORLD", "1")), int(os.environ.get("TC_RANK", "0"))
p = rtk.RenderConfig(width=w, height=h, trace_mode=mode, rank=rank, world_size=world).to_c()
cn = rtk.Counters()
if world == 1:
    rgb = np.zeros((h, w, 3), np.float32)
    for _ in range(3): assert dbg.rtk_render_frame(ac, C.byref(p), rgb.ctypes.data, C.byref(cn)) == 0
    # lane i of block (by, bx) wrote value i at pixel (by*8 + i//8, bx*8 + i%8)
    r = rgb[:, :, 0].reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)[:, :60].astype(np.float64)
else:
    # one rank of a sharded frame: the compact [buckets_per_rank, B, B, 3] buffer; blocks keep their place inside a bucket
    B = int(os.environ.get("TC_BUCKET", "64"))
    dbg.rtk_render_output_floats.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    nf = C.c_size_t(); assert dbg.rtk_render_output_floats(ac, C.byref(p), C.byref(nf)) == 0
    dbg.rtk_render_frame_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dbuf = torch.zeros(nf.value, dtype=torch.float32, device="cuda")
    for _ in range(3):
        assert dbg.rtk_render_frame_device(ac, C.byref(p), dbuf.data_ptr(), None) == 0
        torch.cuda.synchronize()
    buf = dbuf.cpu().numpy()
    bpr = nf.value // (B * B * 3)
    r = buf.reshape(bpr, B, B, 3)[:, :, :, 0].reshape(bpr, B // 8, 8, B // 8, 8).transpose(0, 1, 3, 2, 4).reshape(-1, 64)[:, :60].astype(np.float64)
    r = r[r[:, 0] > 0]                                       # (blocks of the padding buckets never ran)
    print(f"rank {rank} of {world}: {len(r)} blocks")
names = ["total", "trace", "n_trace", "steps", "n_small", "t_small", "c_small", "n_big", "t_big", "c_big", "prologue", "to_first_trace", "first_trace", "after_first_trace", "chunks", "surv", "ctris", "rt0", "rt1", "wg"]
def show(tag, m):
    s = r[m].sum(0); d = dict(zip(names, s)); n = m.sum()
    c_nodes = d["trace"] - d["c_small"] - d["c_big"]
    print(f"--- {tag}: {n} blocks, mean total {d['total']/n:.0f} cycles; trace {100*d['trace']/d['total']:.0f}% "
          f"(nodes {100*c_nodes/d['total']:.0f}%, small leaves {100*d['c_small']/d['total']:.0f}%, sliced leaves {100*d['c_big']/d['total']:.0f}%), rest {100*(1-d['trace']/d['total']):.0f}%")
    print(f"    per block: traces {d['n_trace']/n:.1f}, node steps {d['steps']/n:.0f} ({c_nodes/max(d['steps'],1):.0f} cyc/step), "
          f"small leaves {d['n_small']/n:.1f} with {d['t_small']/n:.0f} tris ({d['c_small']/max(d['t_small'],1):.0f} cyc/tri), "
          f"sliced leaves {d['n_big']/n:.1f} with {d['t_big']/n:.0f} tris ({d['c_big']/max(d['n_big'],1):.0f} cyc/leaf, {d['c_big']/max(d['t_big'],1):.0f} cyc/tri)")
    print(f"    of the total: waiting for the helpers of a light burst {100*r[m][:,20].sum()/d['total']:.0f}%, making bundles {100*r[m][:,33].sum()/d['total']:.0f}%, "
          f"culling the leaf list {100*r[m][:,34].sum()/d['total']:.0f}%")
    g = r[m][:, 36:44].sum(0)
    if g[0] > 0:
        print(f"    owner's exact tests: {g[0]/n:.0f} surviving triangles per block; the wave goes on after det + u estimate for {100*g[1]/g[0]:.0f}% "
              f"({g[2]/max(g[1],1):.1f} lanes alive), after u for {100*g[3]/g[0]:.0f}% ({g[4]/max(g[3],1):.1f}), after v for {100*g[5]/g[0]:.0f}% ({g[6]/max(g[5],1):.1f}); "
              f"{g[7]/g[0]:.2f} accepted lanes per triangle")
    print(f"    owner's bundle culling per block: {d['chunks']/n:.1f} chunks, {d['ctris']/n:.0f} triangles in, {d['surv']/n:.1f} survivors ({100*d['surv']/max(d['ctris'],1):.1f}%)")
tot = r[:, 0]
bg = (r[:, 2] == 1) & (r[:, 4] + r[:, 7] == 0)
print("background blocks (one trace, no leaf):", bg.sum(), "mean cycles: prologue %.0f, ray setup until first trace %.0f, first trace %.0f, after it %.0f; total %.0f"
      % tuple(r[bg][:, [10, 11, 12, 13, 0]].mean(0)))
show("all", tot >= 0)
show("traced >1", r[:, 2] > 1)
order = np.argsort(-tot)
top = np.zeros(len(tot), bool); top[order[:100]] = True
show("100 longest", top)
top = np.zeros(len(tot), bool); top[order[:1000]] = True
show("1000 longest", top)

# ---- timeline (s_memrealtime, 100 MHz): when blocks start / end, how many owners are running
t0 = r[:, 17].copy(); t1 = r[:, 18].copy()
t1 = np.where(t1 < t0, t1 + 2**24, t1)
base = np.median(t0)
t0 = np.where(t0 < base - 2**23, t0 + 2**24, t0); t1 = np.where(t1 < base - 2**23, t1 + 2**24, t1)
t1 -= t0.min(); t0 -= t0.min()
start, end = t0 / 100.0, t1 / 100.0                      # microseconds
print("kernel span %.1f us; block durations us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (end.max(), (end - start).mean(), *np.percentile(end - start, [50, 90, 99]), (end - start).max()))
ts = np.linspace(0, end.max(), 25)
print("owners running over time:", [int(((start <= t) & (end > t)).sum()) for t in ts])
last = np.argsort(-end)[:8]
print("last to finish (end us, start us, duration us, traces, dispatch index):", [(round(float(end[i]), 1), round(float(start[i]), 1), round(float(end[i] - start[i]), 1), int(r[i, 2]), int(r[i, 19])) for i in last])
longest = np.argsort(-(end - start))[:8]
for i in np.argsort(-(end - start))[:6]:
    print("  block %d: %.1f us; burst wait %.1f us; traces (us, kind: 100+log2(parts) = light burst, else rays in the root box): %s" % (
        i, end[i] - start[i], r[i, 20] / 2400.0, [(round(float(r[i, 21 + k]) / 2400.0, 1), int(r[i, 27 + k])) for k in range(6) if r[i, 21 + k] > 0]))
print("longest (duration us, start us, traces, dispatch index, block y, block x):", [(round(float(end[i] - start[i]), 1), round(float(start[i]), 1), int(r[i, 2]), int(r[i, 19]), int(i // (w // 8)), int(i % (w // 8))) for i in longest])
