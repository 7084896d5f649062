"""Generated scenes through the C-ABI scene constructor (rtk_scene_create) against the oracle: triangle soups, a bumpy height
field, axis-aligned quads (rays parallel to box planes -> inf / NaN slabs), zero-area and duplicated triangles, all four
material kinds and textures, several lights, a light exactly on a surface.  Frames must be bit-identical through every engine."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("RTK_SOAK_SEEDS", "8"))      # a soak run sets this to a few hundred


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _same_frame(a, b):
    """Bit-identical, except that a NaN pixel (a light sitting exactly on a surface divides by a zero distance) only has to be a
    NaN on both sides: which payload and sign a NaN carries through an addition is a property of the hardware (x86 SSE keeps the
    first operand's, its default NaN is negative; the GPU's is positive), not of the algorithm."""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(_bits(np.where(na, 0.0, a).astype(np.float32)), _bits(np.where(nb, 0.0, b).astype(np.float32)))


def _make_scene(ora, seed):
    rng = np.random.default_rng(seed)
    verts, idx, nverts, ntris, mesh_mat = [], [], [], [], []

    def add_mesh(v, t, m):
        verts.append(np.asarray(v, np.float32).reshape(-1, 3)); idx.append(np.asarray(t, np.uint32).reshape(-1, 3))
        nverts.append(len(verts[-1])); ntris.append(len(idx[-1])); mesh_mat.append(m)

    n_mat = 5
    # 1) a bumpy height field (many small leaves, shared edges and vertices)
    g = int(rng.integers(6, 24))
    xs, zs = np.meshgrid(np.linspace(-4, 4, g), np.linspace(-4, 4, g), indexing="ij")
    ys = 0.4 * rng.normal(size=xs.shape) - 1.0
    v = np.stack([xs, ys, zs], axis=-1).reshape(-1, 3)
    t = []
    for i in range(g - 1):
        for j in range(g - 1):
            a = i * g + j
            t += [[a, a + 1, a + g], [a + 1, a + g + 1, a + g]]
    add_mesh(v, t, int(rng.integers(0, n_mat)))
    # 2) a triangle soup, including zero-area triangles and a duplicated one
    k = int(rng.integers(20, 300))
    v = rng.uniform(-3, 3, size=(3 * k, 3)) * np.array([1.0, 0.6, 1.0]) + np.array([0, 1.0, 0])
    t = np.arange(3 * k).reshape(k, 3).tolist()
    t += [[0, 0, 1], [2, 2, 2], t[0]]                                       # degenerate, point, duplicate
    add_mesh(v, t, int(rng.integers(0, n_mat)))
    # 3) axis-aligned quads: a floor and a wall (box planes coincide with triangle planes)
    add_mesh([[-6, -2, -6], [6, -2, -6], [6, -2, 6], [-6, -2, 6]], [[0, 2, 1], [0, 3, 2]], int(rng.integers(0, n_mat)))
    add_mesh([[-6, -2, -6], [6, -2, -6], [6, 5, -6], [-6, 5, -6]], [[0, 1, 2], [0, 2, 3]], int(rng.integers(0, n_mat)))
    kinds = np.array([ora.MAT_DIFFUSE, ora.MAT_REFLECTIVE, ora.MAT_REFRACTIVE if seed % 2 else ora.MAT_DIFFUSE, ora.MAT_CONSTANT,
                      ora.MAT_DIFFUSE], np.int32)
    n_l = int(rng.integers(1, 6))
    lights = rng.uniform(-5, 5, size=(n_l, 3)) + np.array([0, 6, 0])
    if seed % 3 == 0:
        lights[0] = [0.0, -2.0, 0.0]                                            # on the floor: zero-length and grazing shadow rays
    c, s_ = np.cos(0.35), np.sin(0.35)
    return ora.FlatScene(
        mesh_material=np.asarray(mesh_mat, np.int32), mesh_nverts=np.asarray(nverts, np.int32), mesh_ntris=np.asarray(ntris, np.int32),
        vertices=np.concatenate(verts).astype(np.float32), indices=np.concatenate(idx).astype(np.uint32),
        mat_kind=kinds, mat_albedo=rng.uniform(0.1, 1.0, size=(n_mat, 3)).astype(np.float32),
        mat_ior=np.full(n_mat, 1.5, np.float32), mat_smooth=rng.integers(0, 2, size=n_mat).astype(np.int32),
        light_pos=lights.astype(np.float32), light_intensity=rng.uniform(100, 2000, size=n_l).astype(np.float32),
        cam_pos=np.array([0.0 if seed % 4 else 0.5, 3.0, 11.0], np.float32),
        cam_mat=np.array([1, 0, 0, 0, c, -s_, 0, s_, c], np.float32), background=np.array([0.1, 0.3, 0.2], np.float32),
        width=96, height=64, bucket_size=int(rng.choice([16, 24, 64])))


def _rtk_scene(rtk, f):
    return rtk.Scene.from_arrays(f.mesh_material, f.mesh_nverts, f.mesh_ntris, f.vertices, f.indices, f.mat_kind, f.mat_albedo,
                                 f.mat_ior, f.mat_smooth, f.light_pos, f.light_intensity, f.cam_pos, f.cam_mat, f.background,
                                 f.width, f.height, f.bucket_size)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_generated_scenes_render_bit_exactly(rtk, ora, seed):
    flat = _make_scene(ora, seed)
    acc = rtk.KdTreeSimdAccel(_rtk_scene(rtk, flat))
    oacc = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD)
    gi = 1 if seed % 4 == 1 else 0
    spp = 1 + seed % 2
    ref, ocn = oacc.render(96, 64, spp, 4, gi)
    assert np.isfinite(ref).any()
    for mode in (rtk.TRACE_AUTO, rtk.TRACE_GROUP4, rtk.TRACE_STREAM, rtk.TRACE_LANE, rtk.TRACE_GROUP8):
        for rep in range(2):                                                   # the second frame runs in cost-feedback order
            rgb, cn = acc.render_frame(rtk.RenderConfig(width=96, height=64, spp=spp, max_ray_depth=4, diffuse_rays=gi, trace_mode=mode))
            assert cn["rays"] == ocn["rays"], (mode, rep)
            assert _same_frame(rgb, ref), (mode, rep)


@pytest.mark.parametrize("seed", range(3))
def test_generated_scenes_intersect_bit_exactly(rtk, ora, seed):
    flat = _make_scene(ora, 100 + seed)
    acc = rtk.KdTreeSimdAccel(_rtk_scene(rtk, flat))
    oacc = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD)
    rng = np.random.default_rng(seed)
    n = 20_000
    o = rng.uniform(-7, 7, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 4, rng.integers(0, 3)] = 0.0                                       # axis-parallel rays
    o[: n // 8, 1] = -2.0                                                       # starting on the floor plane
    rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    for cull in (True, False):
        ref = oacc.intersect(rays, cull)
        for mode in (0, 1, 2):
            got = acc.intersect(rays, cull, mode)
            assert np.array_equal(got["tri"], ref["tri"]), (cull, mode)
            for f in ("t", "u", "v"):
                assert np.array_equal(_bits(got[f]), _bits(ref[f])), (cull, mode, f)
