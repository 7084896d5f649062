"""Host-side Python mirror of the reference's accelerator / render interface over the rtk C-ABI.

Names follow the reference (paths relative to /root/reference/include/raytracer/):
  parse_scene_file      io/json/loader.hpp:235-265
  KdTreeSimdAccel       render/accel/kd_tree_simd.hpp:63-98 (ctor from a scene, intersect<cull>)
  render_frame          render/render.hpp:18-108
  write_ppm             io/image/ppm.hpp:7-25

Everything computes in librtk_hip.so (hand-written HIP for gfx950).  There is no CPU fallback:
if the library is missing this module raises at import, and compute calls raise RtkError
(RTK_ERR_NO_DEVICE) when no GPU is usable.  torch is used only as plumbing (device buffers,
streams, torch.distributed) by the *_device entry points.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RTK_LIB_OVERRIDE") or os.path.join(_HERE, "librtk_hip.so")     # override: A/B builds (tools/)

RTK_OK, RTK_ERR_INVALID, RTK_ERR_NO_DEVICE, RTK_ERR_HIP, RTK_ERR_IO, RTK_ERR_PARSE, RTK_ERR_UNSUPPORTED = range(7)
MAT_DIFFUSE, MAT_REFLECTIVE, MAT_REFRACTIVE, MAT_CONSTANT, MAT_TEXTURE = 0, 1, 2, 3, 4
TEX_ALBEDO, TEX_EDGES, TEX_CHECKER, TEX_BITMAP = 0, 1, 2, 3
TRACE_AUTO, TRACE_LANE, TRACE_WAVE, TRACE_GROUP4, TRACE_GROUP8, TRACE_GROUP16, TRACE_STREAM, TRACE_TWOPASS = 0, 1, 2, 3, 4, 5, 6, 7
TRAVERSAL_REFERENCE, TRAVERSAL_FAST = 0, 1     # rtk.h RTK_TRAVERSAL_*: leaf order of the wave-cooperative walks (FAST is not the parity mode)
TRACE_REPACK = 8        # batched intersect only: rays sorted by origin / direction cell before the trace (csrc/repack.hip)

# every symbol include/rtk.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "rtk_abi_version", "rtk_last_error", "rtk_device_count",
    "rtk_scene_create", "rtk_scene_load_crtscene", "rtk_scene_get_info", "rtk_scene_get_arrays", "rtk_scene_get_textures",
    "rtk_scene_get_bitmaps", "rtk_decode_jpeg",
    "rtk_scene_vertex_normals", "rtk_scene_destroy",
    "rtk_accel_build", "rtk_accel_tree_info", "rtk_accel_tree_dump", "rtk_accel_destroy",
    "rtk_accel_intersect", "rtk_accel_intersect_device", "rtk_accel_intersect_stats",
    "rtk_render_output_floats", "rtk_render_frame", "rtk_render_frame_device", "rtk_render_last_counters",
    "rtk_render_last_critical_path",
    "rtk_tiles_assemble_device", "rtk_camera_rays", "rtk_camera_rays_device",
    "rtk_frame_to_rgb8_device", "rtk_format_ppm_rgb8", "rtk_write_ppm", "rtk_format_ppm",
]


class RtkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rtk error {code}: {msg}")
        self.code = code


if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  This package has no CPU fallback."
    )

# One HIP runtime per process: PyTorch bundles its own libamdhip64/libhsa-runtime64 (same sonames as the system
# ROCm ones but different files).  If librtk_hip.so pulled in the system copies and torch its bundled ones, the
# second runtime to initialise finds no device.  Importing torch first makes the loader satisfy librtk_hip.so's
# libamdhip64.so.7 / libhsa-runtime64.so.1 from the copies torch already mapped.  Pure C/C++ hosts use system ROCm.
try:
    import torch  # noqa: F401  (plumbing only: device buffers, streams, torch.distributed)
except ImportError:  # pragma: no cover - torch is part of the image
    torch = None

_L = C.CDLL(_LIB_PATH)


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_int32), ("mesh_material", C.POINTER(C.c_int32)), ("mesh_nverts", C.POINTER(C.c_int32)),
        ("mesh_ntris", C.POINTER(C.c_int32)), ("vertices", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
        ("n_materials", C.c_int32), ("mat_kind", C.POINTER(C.c_int32)), ("mat_albedo", C.POINTER(C.c_float)),
        ("mat_ior", C.POINTER(C.c_float)), ("mat_smooth", C.POINTER(C.c_int32)),
        ("mat_texture", C.POINTER(C.c_int32)), ("uvs", C.POINTER(C.c_float)), ("mesh_has_uvs", C.POINTER(C.c_int32)),
        ("n_textures", C.c_int32), ("tex_kind", C.POINTER(C.c_int32)), ("tex_color_a", C.POINTER(C.c_float)),
        ("tex_color_b", C.POINTER(C.c_float)), ("tex_param", C.POINTER(C.c_float)),
        ("n_lights", C.c_int32), ("light_pos", C.POINTER(C.c_float)), ("light_intensity", C.POINTER(C.c_float)),
        ("cam_pos", C.c_float * 3), ("cam_mat", C.c_float * 9), ("background", C.c_float * 3),
        ("width", C.c_int32), ("height", C.c_int32), ("bucket_size", C.c_int32),
        ("tex_pixels", C.POINTER(C.c_uint8)), ("tex_bitmap", C.POINTER(C.c_int32)),
    ]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_meshes", "n_materials", "n_lights", "n_vertices", "n_triangles", "width", "height", "bucket_size",
                 "n_textures", "n_uv_vertices", "n_bitmap_bytes")]


class AccelParams(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("max_leaf_size", C.c_int32), ("eps", C.c_float),
                ("normalize_hit_normal", C.c_int32), ("device", C.c_int32), ("traversal", C.c_int32)]


class TreeInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_nodes", "n_inner", "n_leaves", "n_leaf_refs", "max_leaf_refs", "n_triangles", "tree_depth", "reserved")]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_ray_depth", C.c_int32),
                ("diffuse_rays", C.c_int32), ("seed", C.c_uint32), ("fov_degrees", C.c_double),
                ("shadow_bias", C.c_float), ("reflection_bias", C.c_float), ("refraction_bias", C.c_float),
                ("trace_mode", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("collect_stats", C.c_int32),
                ("sample_begin", C.c_int32), ("sample_count", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "primary", "hits", "nodes", "boxpass", "leaves", "tris", "packets16")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RAY_DTYPE = np.dtype([("origin", "<f4", (3,)), ("direction", "<f4", (3,))])
HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4"), ("mesh", "<u4"), ("normal", "<f4", (3,))])
assert RAY_DTYPE.itemsize == 24 and HIT_DTYPE.itemsize == 32

_vp = C.c_void_p
_L.rtk_abi_version.restype = C.c_int
_L.rtk_last_error.restype = C.c_char_p
_L.rtk_device_count.argtypes = [C.POINTER(C.c_int)]
_L.rtk_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(_vp)]
_L.rtk_scene_load_crtscene.argtypes = [C.c_char_p, C.POINTER(_vp)]
_L.rtk_scene_get_info.argtypes = [_vp, C.POINTER(SceneInfo)]
_L.rtk_scene_get_arrays.argtypes = [_vp] * 15
_L.rtk_scene_get_textures.argtypes = [_vp] * 8
_L.rtk_scene_get_bitmaps.argtypes = [_vp] * 3
_L.rtk_decode_jpeg.argtypes = [_vp, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, C.c_size_t]
_L.rtk_scene_vertex_normals.argtypes = [_vp, C.c_int32, _vp]
_L.rtk_scene_destroy.argtypes = [_vp]
_L.rtk_scene_destroy.restype = None
_L.rtk_accel_build.argtypes = [_vp, C.POINTER(AccelParams), C.POINTER(_vp)]
_L.rtk_accel_tree_info.argtypes = [_vp, C.POINTER(TreeInfo)]
_L.rtk_accel_tree_dump.argtypes = [_vp, _vp, _vp, _vp]
_L.rtk_accel_destroy.argtypes = [_vp]
_L.rtk_accel_destroy.restype = None
_L.rtk_accel_intersect.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp]
_L.rtk_accel_intersect_device.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, _vp]
_L.rtk_accel_intersect_stats.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, C.POINTER(Counters)]
_L.rtk_render_output_floats.argtypes = [_vp, C.POINTER(RenderParams), C.POINTER(C.c_size_t)]
_L.rtk_render_frame.argtypes = [_vp, C.POINTER(RenderParams), _vp, C.POINTER(Counters)]
_L.rtk_render_frame_device.argtypes = [_vp, C.POINTER(RenderParams), _vp, _vp]
_L.rtk_render_last_counters.argtypes = [_vp, C.POINTER(Counters)]
_L.rtk_render_last_critical_path.argtypes = [_vp, C.POINTER(C.c_double)]
_L.rtk_tiles_assemble_device.argtypes = [_vp, C.POINTER(RenderParams), _vp, _vp, _vp]
_L.rtk_camera_rays.argtypes = [_vp, C.POINTER(RenderParams), C.c_int32, _vp]
_L.rtk_camera_rays_device.argtypes = [_vp, C.POINTER(RenderParams), C.c_int32, _vp, _vp]
_L.rtk_frame_to_rgb8_device.argtypes = [_vp, C.c_size_t, _vp, _vp]
_L.rtk_format_ppm_rgb8.argtypes = [_vp, C.c_int32, C.c_int32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]
_L.rtk_write_ppm.argtypes = [_vp, C.c_int32, C.c_int32, C.c_char_p]
_L.rtk_format_ppm.argtypes = [_vp, C.c_int32, C.c_int32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]


def _check(rc: int) -> None:
    if rc != RTK_OK:
        raise RtkError(rc, (_L.rtk_last_error() or b"").decode("utf-8", "replace"))


def lib() -> C.CDLL:
    return _L


def lib_path() -> str:
    return _LIB_PATH


def abi_version() -> int:
    return _L.rtk_abi_version()


def device_count() -> int:
    n = C.c_int(0)
    _check(_L.rtk_device_count(C.byref(n)))
    return n.value


def _fp(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Scene:
    """scene<float> (scene/scene.hpp:14-22) held by the library."""

    def __init__(self, handle):
        self._h = handle
        info = SceneInfo()
        _check(_L.rtk_scene_get_info(self._h, C.byref(info)))
        self.info = info

    @classmethod
    def from_arrays(cls, mesh_material, mesh_nverts, mesh_ntris, vertices, indices, mat_kind, mat_albedo, mat_ior,
                    mat_smooth, light_pos, light_intensity, cam_pos, cam_mat, background, width, height, bucket_size=64,
                    mat_texture=None, uvs=None, mesh_has_uvs=None, tex_kind=None, tex_color_a=None, tex_color_b=None,
                    tex_param=None, tex_pixels=None, tex_bitmap=None):
        n_tex = 0 if tex_kind is None else len(tex_kind)
        keep = dict(
            mt=np.ascontiguousarray(mat_texture if mat_texture is not None else np.full(len(mat_kind), -1), np.int32),
            uv=np.ascontiguousarray(uvs if uvs is not None else np.zeros((0, 2)), np.float32),
            hu=np.ascontiguousarray(mesh_has_uvs if mesh_has_uvs is not None else np.zeros(len(mesh_material)), np.int32),
            tk=np.ascontiguousarray(tex_kind if tex_kind is not None else np.zeros(0), np.int32),
            ta=np.ascontiguousarray(tex_color_a if tex_color_a is not None else np.zeros((n_tex, 3)), np.float32),
            tb=np.ascontiguousarray(tex_color_b if tex_color_b is not None else np.zeros((n_tex, 3)), np.float32),
            tp=np.ascontiguousarray(tex_param if tex_param is not None else np.zeros(n_tex), np.float32),
            px=np.ascontiguousarray(tex_pixels if tex_pixels is not None else np.zeros(0), np.uint8),
            bm=np.ascontiguousarray(tex_bitmap if tex_bitmap is not None else np.zeros((n_tex, 3)), np.int32),
            mm=np.ascontiguousarray(mesh_material, np.int32), nv=np.ascontiguousarray(mesh_nverts, np.int32),
            nt=np.ascontiguousarray(mesh_ntris, np.int32), v=np.ascontiguousarray(vertices, np.float32),
            ix=np.ascontiguousarray(indices, np.uint32), mk=np.ascontiguousarray(mat_kind, np.int32),
            ma=np.ascontiguousarray(mat_albedo, np.float32), mi=np.ascontiguousarray(mat_ior, np.float32),
            ms=np.ascontiguousarray(mat_smooth, np.int32), lp=np.ascontiguousarray(light_pos, np.float32),
            li=np.ascontiguousarray(light_intensity, np.float32),
        )
        d = SceneDesc()
        d.n_meshes = len(keep["mm"])
        d.mesh_material, d.mesh_nverts, d.mesh_ntris = _fp(keep["mm"], C.c_int32), _fp(keep["nv"], C.c_int32), _fp(keep["nt"], C.c_int32)
        d.vertices, d.indices = _fp(keep["v"], C.c_float), _fp(keep["ix"], C.c_uint32)
        d.n_materials = len(keep["mk"])
        d.mat_kind, d.mat_albedo = _fp(keep["mk"], C.c_int32), _fp(keep["ma"], C.c_float)
        d.mat_ior, d.mat_smooth = _fp(keep["mi"], C.c_float), _fp(keep["ms"], C.c_int32)
        d.mat_texture, d.uvs, d.mesh_has_uvs = _fp(keep["mt"], C.c_int32), _fp(keep["uv"], C.c_float), _fp(keep["hu"], C.c_int32)
        d.n_textures = n_tex
        d.tex_kind, d.tex_color_a = _fp(keep["tk"], C.c_int32), _fp(keep["ta"], C.c_float)
        d.tex_color_b, d.tex_param = _fp(keep["tb"], C.c_float), _fp(keep["tp"], C.c_float)
        d.tex_pixels, d.tex_bitmap = _fp(keep["px"], C.c_uint8), _fp(keep["bm"], C.c_int32)
        d.n_lights = len(keep["li"])
        d.light_pos, d.light_intensity = _fp(keep["lp"], C.c_float), _fp(keep["li"], C.c_float)
        d.cam_pos[:] = [float(x) for x in np.asarray(cam_pos, np.float32)]
        d.cam_mat[:] = [float(x) for x in np.asarray(cam_mat, np.float32)]
        d.background[:] = [float(x) for x in np.asarray(background, np.float32)]
        d.width, d.height, d.bucket_size = int(width), int(height), int(bucket_size)
        h = _vp()
        _check(_L.rtk_scene_create(C.byref(d), C.byref(h)))
        return cls(h)

    def arrays(self) -> dict:
        i = self.info
        out = dict(
            mesh_material=np.zeros(i.n_meshes, np.int32), mesh_nverts=np.zeros(i.n_meshes, np.int32),
            mesh_ntris=np.zeros(i.n_meshes, np.int32), vertices=np.zeros((i.n_vertices, 3), np.float32),
            indices=np.zeros((i.n_triangles, 3), np.uint32), mat_kind=np.zeros(i.n_materials, np.int32),
            mat_albedo=np.zeros((i.n_materials, 3), np.float32), mat_ior=np.zeros(i.n_materials, np.float32),
            mat_smooth=np.zeros(i.n_materials, np.int32), light_pos=np.zeros((i.n_lights, 3), np.float32),
            light_intensity=np.zeros(i.n_lights, np.float32), cam_pos=np.zeros(3, np.float32),
            cam_mat=np.zeros(9, np.float32), background=np.zeros(3, np.float32),
        )
        _check(_L.rtk_scene_get_arrays(self._h, *[a.ctypes.data for a in out.values()]))
        tex = dict(
            mat_texture=np.zeros(i.n_materials, np.int32), mesh_has_uvs=np.zeros(i.n_meshes, np.int32),
            uvs=np.zeros((i.n_uv_vertices, 2), np.float32), tex_kind=np.zeros(i.n_textures, np.int32),
            tex_color_a=np.zeros((i.n_textures, 3), np.float32), tex_color_b=np.zeros((i.n_textures, 3), np.float32),
            tex_param=np.zeros(i.n_textures, np.float32),
        )
        _check(_L.rtk_scene_get_textures(self._h, *[a.ctypes.data for a in tex.values()]))
        out.update(tex)
        bmp = dict(tex_bitmap=np.zeros((i.n_textures, 3), np.int32), tex_pixels=np.zeros(i.n_bitmap_bytes, np.uint8))
        _check(_L.rtk_scene_get_bitmaps(self._h, bmp["tex_bitmap"].ctypes.data, bmp["tex_pixels"].ctypes.data))
        out.update(bmp)
        out.update(width=i.width, height=i.height, bucket_size=i.bucket_size)
        return out

    def vertex_normals(self, mesh: int) -> np.ndarray:
        nv = int(self.arrays()["mesh_nverts"][mesh])
        out = np.zeros((nv, 3), np.float32)
        _check(_L.rtk_scene_vertex_normals(self._h, mesh, out.ctypes.data))
        return out

    def __del__(self, _destroy=_L.rtk_scene_destroy):     # bound at definition: module globals may be gone at interpreter exit
        if getattr(self, "_h", None):
            _destroy(self._h)
            self._h = None


def decode_jpeg(data: bytes) -> np.ndarray:
    """uint8 [h][w][channels] as `stbi_load(path, &w, &h, &channels, 0)` (scene/texture/bitmap.hpp:15) returns a baseline JPEG."""
    buf = np.frombuffer(data, np.uint8)
    w, h, ch = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(_L.rtk_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), C.byref(ch), None, 0))
    out = np.zeros((h.value, w.value, ch.value), np.uint8)
    _check(_L.rtk_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), C.byref(ch), out.ctypes.data, out.size))
    return out


def parse_scene_file(path: str) -> Scene:
    """io/json/loader.hpp:235-265 (own JSON reader in csrc/crtscene.cpp)."""
    h = _vp()
    _check(_L.rtk_scene_load_crtscene(os.fsencode(path), C.byref(h)))
    return Scene(h)


@dataclass
class RenderConfig:
    """config.hpp:6-17 as runtime values (+ resolution override and multi-GPU sharding)."""
    width: int = 0
    height: int = 0
    spp: int = 1
    max_ray_depth: int = 5
    diffuse_rays: int = 0
    seed: int = 42
    fov_degrees: float = 90.0
    shadow_bias: float = 1e-4
    reflection_bias: float = 1e-4
    refraction_bias: float = 1e-4
    trace_mode: int = TRACE_AUTO
    rank: int = 0
    world_size: int = 1
    collect_stats: int = 0     # 1 (or True): per-ray work counters of the reference algorithm; 2: of the production path (early-exit occlusion queries)
    sample_begin: int = 0      # progressive accumulation: this call renders samples [sample_begin, sample_begin + sample_count)
    sample_count: int = 0      # 0 = all spp samples in one call

    def to_c(self) -> RenderParams:
        return RenderParams(self.width, self.height, self.spp, self.max_ray_depth, self.diffuse_rays, self.seed,
                            self.fov_degrees, np.float32(self.shadow_bias), np.float32(self.reflection_bias),
                            np.float32(self.refraction_bias), self.trace_mode, self.rank, self.world_size,
                            int(self.collect_stats), self.sample_begin, self.sample_count)


class KdTreeSimdAccel:
    """kd_tree_simd_accel<float, eps, max_depth, max_leaf_size> (render/accel/kd_tree_simd.hpp:63-98).

    normalize_hit_normal=False gives kd_tree_accel's un-normalised hit_normal (kd_tree.hpp:140)."""

    def __init__(self, scene: Scene, eps: float = 1e-6, max_depth: int = 8, max_leaf_size: int = 64,
                 normalize_hit_normal: bool = True, device: int = -1, traversal: int = TRAVERSAL_REFERENCE):
        self.scene = scene  # the reference keeps scene_ptr public (render.hpp:21)
        p = AccelParams(max_depth, max_leaf_size, np.float32(eps), 1 if normalize_hit_normal else 0, device, traversal)
        h = _vp()
        _check(_L.rtk_accel_build(scene._h, C.byref(p), C.byref(h)))
        self._h = h

    # ---- tree introspection
    def tree_info(self) -> TreeInfo:
        ti = TreeInfo()
        _check(_L.rtk_accel_tree_info(self._h, C.byref(ti)))
        return ti

    def tree_dump(self):
        ti = self.tree_info()
        box = np.zeros((ti.n_nodes, 6), np.float32)
        link = np.zeros((ti.n_nodes, 4), np.int32)
        refs = np.zeros((ti.n_leaf_refs,), np.int32)
        _check(_L.rtk_accel_tree_dump(self._h, box.ctypes.data, link.ctypes.data, refs.ctypes.data))
        return box, link, refs

    # ---- accel.intersect<cull>(ray), batched
    def intersect(self, rays: np.ndarray, cull: bool, trace_mode: int = TRACE_AUTO) -> np.ndarray:
        """rays: [n,6] float32 (origin xyz, direction xyz) in host memory -> HIT_DTYPE[n]."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0],), HIT_DTYPE)
        _check(_L.rtk_accel_intersect(self._h, rays.ctypes.data, rays.shape[0], 1 if cull else 0, trace_mode, out.ctypes.data))
        return out

    def intersect_device(self, d_rays_ptr: int, n: int, cull: bool, d_hits_ptr: int, trace_mode: int = TRACE_AUTO,
                         stream: int = 0) -> None:
        _check(_L.rtk_accel_intersect_device(self._h, d_rays_ptr, n, 1 if cull else 0, trace_mode, d_hits_ptr, stream))

    def intersect_stats(self, d_rays_ptr: int, n: int, cull: bool, d_hits_ptr: int, trace_mode: int = TRACE_AUTO) -> dict:
        c = Counters()
        _check(_L.rtk_accel_intersect_stats(self._h, d_rays_ptr, n, 1 if cull else 0, trace_mode, d_hits_ptr, C.byref(c)))
        return c.as_dict()

    # ---- frames
    def output_floats(self, cfg: RenderConfig) -> int:
        n = C.c_size_t(0)
        p = cfg.to_c()
        _check(_L.rtk_render_output_floats(self._h, C.byref(p), C.byref(n)))
        return n.value

    def render_frame(self, cfg: RenderConfig, rgb: np.ndarray | None = None):
        """Whole frame into host memory: ([h,w,3] float32, counters dict).  `rgb`: the buffer to render into -- for a later
        pass of a progressive frame (cfg.sample_begin > 0) it holds the running sums of the passes before."""
        w = cfg.width or self.scene.info.width
        h = cfg.height or self.scene.info.height
        p = cfg.to_c()
        n = self.output_floats(cfg)
        assert n == w * h * 3
        if rgb is None:
            if cfg.sample_begin > 0:
                raise ValueError("a pass with sample_begin > 0 needs the buffer the previous passes rendered into")
            rgb = np.zeros((h, w, 3), np.float32)
        assert rgb.dtype == np.float32 and rgb.shape == (h, w, 3) and rgb.flags.c_contiguous
        c = Counters()
        _check(_L.rtk_render_frame(self._h, C.byref(p), rgb.ctypes.data, C.byref(c)))
        return rgb, c.as_dict()

    def render_frame_device(self, cfg: RenderConfig, d_out_ptr: int, stream: int = 0) -> None:
        p = cfg.to_c()
        _check(_L.rtk_render_frame_device(self._h, C.byref(p), d_out_ptr, stream))

    def last_counters(self) -> dict:
        c = Counters()
        _check(_L.rtk_render_last_counters(self._h, C.byref(c)))
        return c.as_dict()

    def camera_rays(self, cfg: RenderConfig, sample: int = 0) -> np.ndarray:
        """[h, w, 6] float32: origin + direction of every pixel's camera ray for sample `sample` (render.hpp:35-62)."""
        w = cfg.width or self.scene.info.width
        h = cfg.height or self.scene.info.height
        rays = np.zeros((h, w, 6), np.float32)
        p = cfg.to_c()
        _check(_L.rtk_camera_rays(self._h, C.byref(p), sample, rays.ctypes.data))
        return rays

    def camera_rays_device(self, cfg: RenderConfig, d_rays_ptr: int, sample: int = 0, stream: int = 0) -> None:
        p = cfg.to_c()
        _check(_L.rtk_camera_rays_device(self._h, C.byref(p), sample, d_rays_ptr, stream))

    def last_critical_path_ms(self) -> float:
        """Longest 8x8 pixel block of the most recent megakernel frame (ms): the frame's critical path."""
        ms = C.c_double(0.0)
        _check(_L.rtk_render_last_critical_path(self._h, C.byref(ms)))
        return ms.value

    def assemble_device(self, cfg: RenderConfig, d_gathered_ptr: int, d_rgb_ptr: int, stream: int = 0) -> None:
        p = cfg.to_c()
        _check(_L.rtk_tiles_assemble_device(self._h, C.byref(p), d_gathered_ptr, d_rgb_ptr, stream))

    def __del__(self, _destroy=_L.rtk_accel_destroy):
        if getattr(self, "_h", None):
            _destroy(self._h)
            self._h = None


def render_frame(accel: KdTreeSimdAccel, cfg: RenderConfig | None = None):
    """render_frame<A,F>(accel, BUCKET_TILES) (render/render.hpp:18-108)."""
    return accel.render_frame(cfg or RenderConfig())


def format_ppm(rgb: np.ndarray) -> bytes:
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    n = C.c_size_t(0)
    _check(_L.rtk_format_ppm(rgb.ctypes.data, w, h, None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _check(_L.rtk_format_ppm(rgb.ctypes.data, w, h, buf, n.value, C.byref(n)))
    return buf.raw[: n.value]


def frame_to_rgb8_device(d_rgb_ptr: int, n_floats: int, d_out_ptr: int, stream: int = 0) -> None:
    """uint8(255.999 * clamp(c, 0, 1)) per channel on the device (io/image/ppm.hpp:17-19)."""
    _check(_L.rtk_frame_to_rgb8_device(d_rgb_ptr, n_floats, d_out_ptr, stream))


def format_ppm_rgb8(rgb8: np.ndarray) -> bytes:
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = rgb8.shape
    n = C.c_size_t(0)
    _check(_L.rtk_format_ppm_rgb8(rgb8.ctypes.data, w, h, None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _check(_L.rtk_format_ppm_rgb8(rgb8.ctypes.data, w, h, buf, n.value, C.byref(n)))
    return buf.raw[: n.value]


def write_ppm(rgb: np.ndarray, path: str) -> None:
    """write_ppm (io/image/ppm.hpp:7-25)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    _check(_L.rtk_write_ppm(rgb.ctypes.data, w, h, os.fsencode(path)))
