"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Bar: bit-exact for indices and — because both sides run the same IEEE operations in the same order with
fp-contract off — also for every float (t, u, v, normals, pixels).  The stated tolerance of the north star is
per-channel |delta| < 1e-4; the tests assert that and report when the result is not exactly 0.
"""
import numpy as np
import pytest

from conftest import CONFIG_SCENES, SCENE2, SCENE5, SCENE8

pytestmark = pytest.mark.gpu

TOL = 1e-4
MODES = {"auto": 0, "lane": 1, "wave": 2, "repack": 8}          # batched intersect (repack: rays sorted by cell first, csrc/repack.hip)
FRAME_MODES = {"auto": 0, "lane": 1, "wave": 2, "group4": 3, "group8": 4, "group16": 5, "stream": 6, "twopass": 7}   # frames: + workgroup-cooperative leaves, streaming pipeline


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _scene_pair(rtk, ora, path, normalize=True, okind=None):
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path), normalize_hit_normal=normalize)
    osc = ora.Scene(ora.load_crtscene(path))
    oacc = ora.Accel(osc, ora.ACCEL_KD_SIMD if okind is None else okind)
    return acc, oacc


def _mixed_rays(flat, n, seed):
    """Camera-like rays, random rays inside the scene box, axis-parallel rays (0 components -> inf/NaN slabs)."""
    rng = np.random.default_rng(seed)
    lo = flat.vertices.min(axis=0) - 1.0
    hi = flat.vertices.max(axis=0) + 1.0
    k = n // 4
    # 1) from the camera towards points in the scene box
    tgt = rng.uniform(lo, hi, size=(k, 3)).astype(np.float32)
    o1 = np.broadcast_to(flat.cam_pos, (k, 3)).astype(np.float32)
    d1 = tgt - o1
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True).astype(np.float32)
    # 2) random origins in the box, random directions (un-normalised on purpose)
    o2 = rng.uniform(lo, hi, size=(k, 3)).astype(np.float32)
    d2 = rng.normal(size=(k, 3)).astype(np.float32)
    # 3) axis-parallel and plane-parallel directions
    o3 = rng.uniform(lo, hi, size=(k, 3)).astype(np.float32)
    d3 = rng.normal(size=(k, 3)).astype(np.float32)
    zero = rng.integers(0, 3, size=k)
    d3[np.arange(k), zero] = 0.0
    half = k // 2
    d3[np.arange(half), (zero[:half] + 1) % 3] = 0.0
    d3[np.arange(0, k, 7), zero[::7]] = -0.0
    # 4) origins exactly on vertices / box planes, pointing at other vertices (edge and vertex grazing)
    vi = rng.integers(0, flat.vertices.shape[0], size=(n - 3 * k, 2))
    o4 = flat.vertices[vi[:, 0]].copy()
    d4 = flat.vertices[vi[:, 1]] - o4 + np.float32(1e-3) * rng.normal(size=o4.shape).astype(np.float32)
    o4 = o4 - d4  # start one segment length before the first vertex
    rays = np.concatenate([np.concatenate([o, d], axis=1) for o, d in ((o1, d1), (o2, d2), (o3, d3), (o4, d4))])
    return np.ascontiguousarray(rays.astype(np.float32))


@pytest.mark.parametrize("scene", list(CONFIG_SCENES))
@pytest.mark.parametrize("cull", [True, False])
@pytest.mark.parametrize("mode", list(MODES))
def test_intersect_matches_oracle(rtk, ora, scene, cull, mode):
    acc, oacc = _scene_pair(rtk, ora, CONFIG_SCENES[scene])
    rays = _mixed_rays(oacc.scene.flat, 60_000, seed=hash((scene, cull)) & 0xFFFF)
    got = acc.intersect(rays, cull, MODES[mode])
    ref = oacc.intersect(rays, cull)
    assert np.array_equal(got["tri"], ref["tri"])
    assert np.array_equal(got["mesh"], ref["mesh"])
    for f in ("t", "u", "v"):
        assert np.array_equal(_bits(got[f]), _bits(ref[f])), f
    hit = ref["tri"] != 0xFFFFFFFF
    assert hit.sum() > 1000
    # hit_normal may be NaN for degenerate vertex normals on both sides; compare bit patterns of finite ones and NaN-ness
    gn, rn = got["normal"][hit], ref["normal"][hit]
    assert np.array_equal(np.isnan(gn), np.isnan(rn))
    assert np.array_equal(_bits(np.nan_to_num(gn)), _bits(np.nan_to_num(rn)))


def _boundary_rays(flat, n, seed):
    """Rays aimed exactly (to float rounding) at triangle vertices, edge points and a hair outside edges: the barycentric
    tests 0 <= u, u <= 1, 0 <= v, u + v <= 1 are decided by the last bits.  The wave path settles clear misses with a
    reciprocal estimate before it runs the IEEE division (trace.hip.hpp); this is where an unsafe shortcut would show."""
    rng = np.random.default_rng(seed)
    starts = np.concatenate([[0], np.cumsum(flat.mesh_nverts)[:-1]])
    tri_mesh = np.repeat(np.arange(len(flat.mesh_ntris)), flat.mesh_ntris)
    gidx = flat.indices.astype(np.int64) + starts[tri_mesh][:, None]
    t = rng.integers(0, gidx.shape[0], size=n)
    v0, v1, v2 = (flat.vertices[gidx[t, k]].astype(np.float64) for k in range(3))
    kind = rng.integers(0, 5, size=n)
    a = rng.uniform(0, 1, size=n)
    eps = np.where(kind == 4, rng.choice([-1.0, 1.0], size=n) * 10.0 ** rng.uniform(-9, -5, size=n), 0.0)
    bu = np.select([kind == 0, kind == 1, kind == 2, kind >= 3], [np.zeros(n), a, np.zeros(n), a])           # vertex v0 | edge v0v1 | edge v0v2 | edge v1v2
    bv = np.select([kind == 0, kind == 1, kind == 2, kind >= 3], [np.zeros(n), np.zeros(n) + eps, a, 1.0 - a + eps])
    target = v0 + bu[:, None] * (v1 - v0) + bv[:, None] * (v2 - v0)
    origin = target + rng.normal(size=(n, 3)) * rng.uniform(0.5, 30.0, size=(n, 1))
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True) * rng.choice([1.0, 0.37, 4.0], size=(n, 1))
    return np.ascontiguousarray(np.concatenate([origin, d], axis=1).astype(np.float32))


@pytest.mark.parametrize("scene", list(CONFIG_SCENES))
@pytest.mark.parametrize("cull", [True, False])
def test_intersect_rays_through_vertices_and_edges(rtk, ora, scene, cull):
    acc, oacc = _scene_pair(rtk, ora, CONFIG_SCENES[scene])
    rays = _boundary_rays(oacc.scene.flat, 50_000, seed=len(scene) + int(cull))
    ref = oacc.intersect(rays, cull)
    assert (ref["tri"] != 0xFFFFFFFF).sum() > 5000
    for mode in MODES.values():
        got = acc.intersect(rays, cull, mode)
        assert np.array_equal(got["tri"], ref["tri"]), mode
        for f in ("t", "u", "v"):
            assert np.array_equal(_bits(got[f]), _bits(ref[f])), (mode, f)
    # how sharp the set is: a fair share of the hits sit within a few ulps of a barycentric boundary
    hit = ref["tri"] != 0xFFFFFFFF
    u, v = ref["u"][hit].astype(np.float64), ref["v"][hit].astype(np.float64)
    assert (np.minimum(np.minimum(u, v), 1.0 - u - v) < 1e-6).sum() > 500


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 1000])
def test_intersect_ragged_sizes(rtk, ora, n):
    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    rays = _mixed_rays(oacc.scene.flat, max(n, 8), seed=n)[:n]
    for mode in MODES.values():
        got = acc.intersect(rays, True, mode)
        ref = oacc.intersect(rays, True)
        assert got.shape == (n,)
        assert np.array_equal(got["tri"], ref["tri"])
        assert np.array_equal(_bits(got["t"]), _bits(ref["t"]))


def test_large_incoherent_batches_are_repacked_without_changing_a_bit(rtk, ora):
    """2^19 rays in no useful order: RTK_TRACE_AUTO probes the batch, finds it incoherent and sorts it (csrc/repack.hip);
    RTK_TRACE_REPACK sorts unconditionally.  Hits must land in the caller's order with the bits of the unsorted walk, and
    of the oracle on a sample.  A coherent batch of the same size (AUTO's probe says no) and degenerate batches (all rays
    equal; NaN rays) go through the same entry points."""
    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    n = 1 << 19
    rng = np.random.default_rng(7)
    cam = acc.camera_rays(rtk.RenderConfig(width=1024, height=512), 0).reshape(-1, 6)            # 2^19 pixel-centre rays, row-major
    assert cam.shape[0] == n
    sets = {"coherent": cam, "shuffled": cam[rng.permutation(n)],
            "mixed": _mixed_rays(oacc.scene.flat, n, seed=11)[rng.permutation(n)]}
    same = np.repeat(cam[12345:12346], 70_000, axis=0)
    bad = cam[:70_000].copy(); bad[::97, 3] = np.nan; bad[5::131, 0] = np.inf
    sets["all_equal"] = same
    sets["nan_inf"] = bad
    for name, rays in sets.items():
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        ref = acc.intersect(rays, True, MODES["wave"])
        for mode in ("auto", "repack"):
            got = acc.intersect(rays, True, MODES[mode])
            assert got.tobytes() == ref.tobytes(), (name, mode)
        k = 40_000
        o = oacc.intersect(rays[:k], True)
        assert np.array_equal(ref["tri"][:k], o["tri"]) and np.array_equal(_bits(ref["t"][:k]), _bits(o["t"])), name
    assert (acc.intersect(np.ascontiguousarray(sets["shuffled"]), True, MODES["repack"])["tri"] != 0xFFFFFFFF).sum() > 10_000


def test_intersect_unnormalized_normal_matches_kd_tree_accel(rtk, ora):
    """normalize_hit_normal=0 reproduces kd_tree.hpp:140: compare with the scalar accel's normals."""
    acc, oacc = _scene_pair(rtk, ora, SCENE5, normalize=False, okind=ora.ACCEL_KD_SCALAR)
    rays = _mixed_rays(oacc.scene.flat, 40_000, seed=7)
    got = acc.intersect(rays, False)
    ref = oacc.intersect(rays, False)
    # The two reference accels are not equivalent on exact ties (different trees -> different tie order) nor at
    # det == eps / t == eps (triangle.hpp:38-62 vs kd_tree_simd.hpp:33-57); the vertex-grazing rays of the mixed
    # set provoke such ties, so allow a handful of different winners, but they must be ties in t.
    same = got["tri"] == ref["tri"]
    assert same.mean() > 0.999
    diff = ~same & (got["tri"] != 0xFFFFFFFF) & (ref["tri"] != 0xFFFFFFFF)
    assert np.allclose(got["t"][diff], ref["t"][diff], rtol=1e-5, atol=1e-6)
    hit = same & (ref["tri"] != 0xFFFFFFFF)
    assert hit.sum() > 1000
    assert np.array_equal(_bits(np.nan_to_num(got["normal"][hit])), _bits(np.nan_to_num(ref["normal"][hit])))


RENDER_CASES = [
    # name, scene, w, h, spp, depth, diffuse
    ("scene5_640x480", SCENE5, 640, 480, 1, 5, 0),            # BASELINE config 1 shape
    ("scene5_ragged_203x117", SCENE5, 203, 117, 1, 5, 0),     # not a multiple of 8 or of the bucket
    ("scene8_480x270_d10", SCENE8, 480, 270, 1, 10, 0),       # refractive forks, config 3 at spp 1
    ("hw15_scene2_384", SCENE2, 384, 384, 1, 5, 0),           # bucket 24
    ("scene5_spp4", SCENE5, 320, 180, 4, 5, 0),               # jittered samples (counter-based RNG)
    ("scene8_spp2_d10", SCENE8, 240, 136, 2, 10, 0),
    ("hw15_scene2_gi", SCENE2, 160, 160, 8, 5, 1),            # config 4 shape: GI + refraction + reflection
    ("hw15_scene2_gi3", SCENE2, 96, 96, 2, 4, 3),             # several diffuse rays per hit
    ("scene5_depth0", SCENE5, 160, 90, 1, 0, 0),              # max_ray_depth 0 -> background everywhere hit
    ("scene8_depth1", SCENE8, 160, 90, 1, 1, 0),
]


@pytest.mark.parametrize("case", RENDER_CASES, ids=[c[0] for c in RENDER_CASES])
@pytest.mark.parametrize("mode", list(FRAME_MODES))
def test_render_gate_a(rtk, ora, case, mode):
    """Gate A: HIP frame vs the kd_tree_simd_accel restatement — expect max|delta| = 0, require < 1e-4."""
    _, path, w, h, spp, depth, diffuse = case
    acc, oacc = _scene_pair(rtk, ora, path)
    cfg = rtk.RenderConfig(width=w, height=h, spp=spp, max_ray_depth=depth, diffuse_rays=diffuse, trace_mode=FRAME_MODES[mode])
    if mode == "twopass" and spp != 1:
        with pytest.raises(rtk.RtkError) as e:
            acc.render_frame(cfg)
        assert e.value.code == rtk.RTK_ERR_UNSUPPORTED
        return
    rgb, cn = acc.render_frame(cfg)
    ref, ocn = oacc.render(w, h, spp, depth, diffuse)
    assert cn["rays"] == ocn["rays"]
    assert cn["primary"] == ocn["primary"] == w * h * spp
    delta = float(np.max(np.abs(rgb - ref)))
    assert delta < TOL
    assert delta == 0.0, f"within tolerance but not bit-exact: {delta}"


@pytest.mark.parametrize("n_lights", [1, 2, 3, 5, 9])
def test_light_bursts_with_any_number_of_lights(rtk, ora, n_lights, tmp_path):
    """The light loop (render.hpp:184-206) runs as bursts of jobs (light, part of the lanes) over the waves of a workgroup:
    one light -> four parts, two -> two each, more lights than waves -> several bursts.  scene5 with its lights replaced by
    1 .. 9 lights around the dragon, in every frame mode that has helpers and in the sequential ones; bit-exact vs the oracle,
    also sharded eight ways (few blocks per rank -> RTK_TRACE_AUTO takes the 8-wave workgroups)."""
    import json
    doc = json.load(open(SCENE5))
    base = doc["lights"]
    doc["lights"] = [{"intensity": base[i % len(base)]["intensity"] * (0.5 + 0.25 * (i % 3)),
                      "position": [float(9 - 4 * i), float(7 + (i * 5) % 11), float(-3 + 2 * (i % 4))]} for i in range(n_lights)]
    path = str(tmp_path / f"scene5_{n_lights}_lights.crtscene")
    json.dump(doc, open(path, "w"))
    acc, oacc = _scene_pair(rtk, ora, path)
    w, h = 320, 184
    ref, ocn = oacc.render(w, h, 1, 5, 0)
    for mode in ("auto", "group4", "group8", "group16", "wave", "stream"):
        rgb, cn = acc.render_frame(rtk.RenderConfig(width=w, height=h, max_ray_depth=5, trace_mode=FRAME_MODES[mode]))
        assert cn["rays"] == ocn["rays"], mode
        assert np.array_equal(_bits(rgb), _bits(ref)), (mode, float(np.max(np.abs(rgb - ref))))
    # a repeated frame (cost feedback: packed and single-block workgroups side by side) ...
    for _ in range(3):
        rgb, _ = acc.render_frame(rtk.RenderConfig(width=w, height=h, max_ray_depth=5, trace_mode=FRAME_MODES["group4"]))
        assert np.array_equal(_bits(rgb), _bits(ref))
    # ... and the frame sharded eight ways, twice (the second time with the ranks' own block orders)
    import torch
    world = 8
    cfgs = [rtk.RenderConfig(width=w, height=h, max_ray_depth=5, rank=r, world_size=world) for r in range(world)]
    gathered = torch.full((world, acc.output_floats(cfgs[0])), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        for r in range(world):
            acc.render_frame_device(cfgs[r], gathered[r].data_ptr(), stream)
        acc.assemble_device(cfgs[0], gathered.data_ptr(), out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(out.cpu().numpy()), _bits(ref))


@pytest.mark.parametrize("w,h", [(8, 8), (9, 7), (16, 8), (24, 24), (64, 64), (200, 120)])
def test_tiny_frames_repeated_through_the_cost_feedback(rtk, ora, w, h):
    """One pixel block, a ragged one, a handful: the block order / workgroup list machinery (k_order_by_cost: single-block and
    packed workgroups merged by expected duration, refreshed every 16th frame) must cope with frames that have fewer blocks than a
    workgroup has waves, and every repetition must keep the bits."""
    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    ref, _ = oacc.render(w, h, 1, 5, 0)
    for mode in ("auto", "group4", "group8"):
        for i in range(36):
            rgb, _ = acc.render_frame(rtk.RenderConfig(width=w, height=h, max_ray_depth=5, trace_mode=FRAME_MODES[mode]))
            assert np.array_equal(_bits(rgb), _bits(ref)), (mode, i)


@pytest.mark.parametrize("scene", list(CONFIG_SCENES))
def test_render_gate_b_kd_tree_accel(rtk, ora, scene):
    """Gate B (BASELINE wording): normalize_hit_normal=0 vs the CPU kd_tree_accel render, |delta| < 1e-4."""
    acc, oacc = _scene_pair(rtk, ora, CONFIG_SCENES[scene], normalize=False, okind=ora.ACCEL_KD_SCALAR)
    depth = 10 if scene == "scene8" else 5
    rgb, cn = acc.render_frame(rtk.RenderConfig(width=480, height=272, max_ray_depth=depth))
    ref, ocn = oacc.render(480, 272, 1, depth, 0)
    assert cn["rays"] == ocn["rays"]
    assert float(np.max(np.abs(rgb - ref))) < TOL


def test_render_config2_full_size(rtk, ora):
    """BASELINE config 2 at full size: exact frame, and the ray count the reference itself produces."""
    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    rgb, cn = acc.render_frame(rtk.RenderConfig(width=1920, height=1080, spp=1, max_ray_depth=5))
    ref, ocn = oacc.render(1920, 1080, 1, 5, 0)
    assert cn["rays"] == ocn["rays"] == 2_726_485
    assert cn["primary"] == 2_073_600
    assert float(np.max(np.abs(rgb - ref))) == 0.0
    assert rtk.format_ppm(rgb) == ora.write_ppm(ref)


@pytest.mark.parametrize("scene,depth", [("scene5", 5), ("scene8", 10), ("hw15_scene2", 5)])
def test_work_counters_match_oracle(rtk, ora, scene, depth):
    """Per-ray work (nodes popped, boxes passed, leaves, triangles, W=16 packets) equals the CPU restatement's:
    the algorithmic-byte figure of the roofline is computed from these."""
    acc, oacc = _scene_pair(rtk, ora, CONFIG_SCENES[scene])
    for name, mode in FRAME_MODES.items():
        cfg = rtk.RenderConfig(width=320, height=184, max_ray_depth=depth, trace_mode=mode, collect_stats=True)
        _, cn = acc.render_frame(cfg)
        _, ocn = ora.Accel(oacc.scene, ora.ACCEL_KD_SIMD, W=16).render(320, 184, 1, depth, 0)
        assert cn["rays"] == ocn["rays"]
        assert cn["hits"] == ocn["hits"]
        assert cn["nodes"] == ocn["nodes"]
        assert cn["boxpass"] == ocn["boxpass"]
        assert cn["leaves"] == ocn["leaves"]
        assert cn["tris"] == ocn["tris"]
        assert cn["packets16"] == ocn["packets"]


# ---------------------------------------------------------------- size-independent properties at full size

def test_full_size_properties_4k(rtk, ora):
    """3840x2160 (config 5 resolution): run-to-run determinism, traversal-strategy independence and shard
    independence, plus the oracle's ray count — properties that do not need a full-size reference image."""
    import torch

    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    w, h = 3840, 2160
    base, cn = acc.render_frame(rtk.RenderConfig(width=w, height=h))
    again, _ = acc.render_frame(rtk.RenderConfig(width=w, height=h))
    assert np.array_equal(_bits(base), _bits(again))
    for mode in (1, 2, 3, 4, 5, 6, 7):
        other, cn2 = acc.render_frame(rtk.RenderConfig(width=w, height=h, trace_mode=mode))
        assert cn2["rays"] == cn["rays"]
        assert np.array_equal(_bits(base), _bits(other))
    _, ocn = oacc.render(w, h, 1, 5, 0)
    assert cn["rays"] == ocn["rays"]
    # sharded over 3 ranks on one device, gathered and assembled
    world = 3
    cfgs = [rtk.RenderConfig(width=w, height=h, rank=r, world_size=world) for r in range(world)]
    n = acc.output_floats(cfgs[0])
    gathered = torch.empty((world, n), dtype=torch.float32, device="cuda")
    for r in range(world):
        acc.render_frame_device(cfgs[r], gathered[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    acc.assemble_device(cfgs[0], gathered.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(base))


@pytest.mark.parametrize("scene,w,h,world", [("scene5", 640, 360, 2), ("hw15_scene2", 200, 200, 4), ("scene8", 333, 77, 8),
                                             # whole rounds of buckets per row: the diagonal deal (kernels.hpp rank_bucket)
                                             ("hw15_scene2", 192, 120, 8), ("scene5", 512, 300, 4), ("hw15_scene2", 96, 96, 2)])
@pytest.mark.parametrize("mode", [0, 6])
def test_sharded_frames_assemble_to_the_unsharded_frame(rtk, ora, scene, w, h, world, mode):
    import torch

    acc, _ = _scene_pair(rtk, ora, CONFIG_SCENES[scene])
    depth = 10 if scene == "scene8" else 5
    base, cn = acc.render_frame(rtk.RenderConfig(width=w, height=h, max_ray_depth=depth))
    cfgs = [rtk.RenderConfig(width=w, height=h, max_ray_depth=depth, rank=r, world_size=world, trace_mode=mode) for r in range(world)]
    n = acc.output_floats(cfgs[0])
    gathered = torch.full((world, n), float("nan"), dtype=torch.float32, device="cuda")
    rays = 0
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        acc.render_frame_device(cfgs[r], gathered[r].data_ptr(), stream)
        rays += acc.last_counters()["rays"]
    assert rays == cn["rays"]
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    acc.assemble_device(cfgs[0], gathered.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(out.cpu().numpy()), _bits(base))


@pytest.mark.parametrize("mode", ["group4", "wave", "lane", "group16"])
def test_cost_feedback_reorders_blocks_but_not_results(rtk, ora, mode):
    """From the second frame of a shape on, the megakernel starts its pixel blocks most-expensive-first, using the cycle
    counts the previous frame reported (api.hip "cost feedback").  Frames 1, 2, 3 must all be the oracle's frame, also when
    another shape is rendered in between, and for a rank of a sharded frame."""
    import torch

    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    ref, ocn = oacc.render(333, 190, 1, 5, 0)
    cfg = rtk.RenderConfig(width=333, height=190, max_ray_depth=5, trace_mode=FRAME_MODES[mode])
    other = rtk.RenderConfig(width=100, height=60, max_ray_depth=5, trace_mode=FRAME_MODES[mode])
    ref_other, _ = oacc.render(100, 60, 1, 5, 0)
    for i in range(11):                                             # the first frame's order comes from the camera-ray prior, the second's from measured costs, then every 16th is re-sorted
        rgb, cn = acc.render_frame(cfg)
        assert cn["rays"] == ocn["rays"], i
        assert np.array_equal(_bits(rgb), _bits(ref)), i
        if i == 2:                                                  # a different shape invalidates the recorded costs
            rgb2, _ = acc.render_frame(other)
            assert np.array_equal(_bits(rgb2), _bits(ref_other))
    # one rank of a 3-way sharded frame, three times: the compact bucket buffer must not change
    scfg = rtk.RenderConfig(width=333, height=190, max_ray_depth=5, trace_mode=FRAME_MODES[mode], rank=1, world_size=3)
    n = acc.output_floats(scfg)
    outs = []
    for i in range(3):
        buf = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
        acc.render_frame_device(scfg, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(buf.cpu().numpy())
    assert np.array_equal(_bits(outs[0]), _bits(outs[1])) and np.array_equal(_bits(outs[0]), _bits(outs[2]))


@pytest.mark.parametrize("mode", ["group4", "stream", "lane"])
def test_early_exit_occlusion_queries_change_the_work_not_the_frame(rtk, ora, mode):
    """Scenes without transmissive materials: is_occluded only needs "closest hit nearer than the light?", so the production
    path lets a shadow ray stop at the first hit that answers yes (a prefix of the reference's evaluation order).  The frame
    and the number of intersect() calls are the oracle's; collect_stats=1 still reports the reference's per-ray work,
    collect_stats=2 what was actually visited (never more)."""
    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    ref, ocn = oacc.render(320, 180, 1, 5, 0)
    out = {}
    for stats in (0, 1, 2):
        rgb, cn = acc.render_frame(rtk.RenderConfig(width=320, height=180, max_ray_depth=5, trace_mode=FRAME_MODES[mode], collect_stats=stats))
        assert np.array_equal(_bits(rgb), _bits(ref)), stats
        assert cn["rays"] == ocn["rays"], stats
        out[stats] = cn
    for k in ("nodes", "boxpass", "leaves", "tris", "hits"):
        assert out[1][k] == ocn[k], k                                   # the reference's work, exactly
    assert out[2]["tris"] < out[1]["tris"] and out[2]["nodes"] <= out[1]["nodes"]


@pytest.mark.parametrize("scene,depth,gi", [("scene8", 10, 0), ("hw15_scene2", 5, 1)])
def test_auto_engine_trials_on_forking_scenes_keep_the_frame(rtk, ora, scene, depth, gi):
    """RTK_TRACE_AUTO on a scene whose ray trees fork times the streaming pipeline and the megakernel on the first frames of
    a shape and keeps the faster (api.hip).  Whichever engine a frame goes through, it is the oracle's frame."""
    acc, oacc = _scene_pair(rtk, ora, CONFIG_SCENES[scene])
    ref, ocn = oacc.render(240, 136, 2, depth, gi)
    cfg = rtk.RenderConfig(width=240, height=136, spp=2, max_ray_depth=depth, diffuse_rays=gi)
    for i in range(8):
        rgb, cn = acc.render_frame(cfg)
        assert cn["rays"] == ocn["rays"], i
        assert np.array_equal(_bits(rgb), _bits(ref)), i


def test_streaming_pipeline_queue_overflow_falls_back_to_the_megakernel(rtk, ora, monkeypatch):
    """With room for the camera rays only, every refractive / GI child overflows the node queue: the frame must
    still be exact (redone by the megakernel) and the counters must describe one frame, not two."""
    _, oacc = _scene_pair(rtk, ora, SCENE8)
    monkeypatch.setenv("RTK_STREAM_NODE_FACTOR", "1")              # environment knobs are read when an accel is built
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE8))
    rgb, cn = acc.render_frame(rtk.RenderConfig(width=200, height=120, max_ray_depth=6, trace_mode=FRAME_MODES["stream"]))
    ref, ocn = oacc.render(200, 120, 1, 6, 0)
    assert cn["rays"] == ocn["rays"]
    assert np.array_equal(_bits(rgb), _bits(ref))
    monkeypatch.delenv("RTK_STREAM_NODE_FACTOR")
    acc2 = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE8))
    rgb2, cn2 = acc2.render_frame(rtk.RenderConfig(width=200, height=120, max_ray_depth=6, trace_mode=FRAME_MODES["stream"]))
    assert cn2["rays"] == ocn["rays"] and np.array_equal(_bits(rgb2), _bits(ref))


def test_device_buffers_and_stream(rtk, ora):
    """The device-pointer entry points on a non-default torch stream."""
    import torch

    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    rays = _mixed_rays(oacc.scene.flat, 10_000, seed=3)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d_rays = torch.from_numpy(rays).to("cuda", non_blocking=False)
        d_hits = torch.empty((rays.shape[0], 32), dtype=torch.uint8, device="cuda")
        acc.intersect_device(d_rays.data_ptr(), rays.shape[0], False, d_hits.data_ptr(), 0, s.cuda_stream)
    s.synchronize()
    got = d_hits.cpu().numpy().view(rtk.HIT_DTYPE).reshape(-1)
    ref = oacc.intersect(rays, False)
    assert np.array_equal(got["tri"], ref["tri"])
    assert np.array_equal(_bits(got["t"]), _bits(ref["t"]))


def test_bad_arguments_are_reported_not_thrown(rtk):
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE5))
    with pytest.raises(rtk.RtkError) as e:
        acc.render_frame(rtk.RenderConfig(max_ray_depth=99))
    assert e.value.code == rtk.RTK_ERR_INVALID
    with pytest.raises(rtk.RtkError):
        acc.render_frame(rtk.RenderConfig(spp=0))
    with pytest.raises(rtk.RtkError):
        acc.intersect(np.zeros((4, 6), np.float32), True, trace_mode=17)


def test_raster_batches_are_dealt_as_8x8_blocks_without_changing_a_bit(rtk, ora):
    """RTK_TRACE_AUTO on a large coherent batch that is rows of camera rays (k_raster_probe finds the row length) hands 8x8 pixel
    blocks to the waves instead of 64x1 strips.  Lane placement only: every hit is where the caller's order puts it, same bits
    as RTK_TRACE_WAVE on the plain order -- for whole frames tiled, a ragged tail, a width that is no multiple of 8 (no tiling),
    and a batch that only looks like a raster at its start."""
    import torch

    acc, _ = _scene_pair(rtk, ora, SCENE5)
    st = torch.cuda.current_stream().cuda_stream
    for (w, h, n) in [(1920, 1080, 1920 * 1080 + 1920 * 3 + 77), (640, 360, 640 * 360 * 2), (333, 400, 333 * 400 * 3)]:
        cfg = rtk.RenderConfig(width=w, height=h)
        cam = torch.empty((w * h, 6), dtype=torch.float32, device="cuda")
        acc.camera_rays_device(cfg, cam.data_ptr(), 0, st)
        rays = cam.repeat(-(-n // (w * h)), 1)[:n].contiguous()
        if n >= 1 << 20:                                            # a tail that stops looking like the raster
            rays[-50_000:] = rays[torch.randperm(n, device="cuda")[:50_000]]
        want = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
        got = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
        acc.intersect_device(rays.data_ptr(), n, True, want.data_ptr(), rtk.TRACE_WAVE, st)
        acc.intersect_device(rays.data_ptr(), n, True, got.data_ptr(), rtk.TRACE_AUTO, st)
        torch.cuda.synchronize()
        assert torch.equal(want, got), (w, h, n)


def test_sort_keys_of_the_repacking_cover_their_cases(rtk, ora):
    """The key paths of csrc/repack.hip, each against the unsorted wave walk (same bits, caller's order): one origin with directions all
    over the sphere (octahedral map incl. its folded hemisphere, a first ray that points along -x: negative pole), a few origins
    (three-component direction cells), a size that ends in a partial 256-ray tile, and ray buffers that are only 8- and 4-byte aligned
    (the float4 staging is skipped)."""
    import torch

    acc, oacc = _scene_pair(rtk, ora, SCENE5)
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(21)
    n = (1 << 18) + 77
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[0] = (-1.0, 0.05, 0.02)
    d[5::1001] = 0.0                                                 # zero directions: NaN keys, misses
    one = np.concatenate([np.tile(np.float32([0.3, 2.0, 1.5]), (n, 1)), d], axis=1)
    few = one.copy(); few[:, 0] += rng.integers(0, 3, n).astype(np.float32) * 4.0
    sets = {"one_origin_sphere": one, "few_origins": few}
    for name, rays in sets.items():
        for shift in (0, 2, 1):                                     # floats the ray buffer is shifted by: 16-, 8-, 4-byte aligned
            flat = torch.zeros((n * 6 + 4,), dtype=torch.float32, device="cuda")
            flat[shift:shift + n * 6] = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32).reshape(-1)).cuda()
            ptr = flat.data_ptr() + 4 * shift
            want = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
            acc.intersect_device(ptr, n, False, want.data_ptr(), rtk.TRACE_WAVE, st)
            for mode in (rtk.TRACE_AUTO, MODES["repack"]):
                got = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
                acc.intersect_device(ptr, n, False, got.data_ptr(), mode, st)
                torch.cuda.synchronize()
                assert torch.equal(want, got), (name, shift, mode)
        k = 30_000
        o = oacc.intersect(np.ascontiguousarray(rays[:k], dtype=np.float32), False)
        w = np.frombuffer(want.cpu().numpy().tobytes(), dtype=rtk.HIT_DTYPE)[:k]
        assert np.array_equal(w["tri"], o["tri"]) and np.array_equal(_bits(w["t"]), _bits(o["t"])), name
        assert (o["tri"] != 0xFFFFFFFF).sum() > 1000
