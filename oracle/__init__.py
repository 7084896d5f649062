"""ctypes front end for the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY.  Allowed importers: tests/, __graft_entry__.smoke(), and the
cpu_baseline leg of bench.py.  The product package (simd-raytracer_amd/) never imports this.

Parity-pin status: see oracle/rt_oracle.h — pinned by the reference's own committed renders
(tests/golden/ref_outputs/, every byte of refractive_dragon.png and textures.png) and by the
reference-measured counters recorded in SURVEY.md.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MAT_DIFFUSE, MAT_REFLECTIVE, MAT_REFRACTIVE, MAT_CONSTANT, MAT_TEXTURE = 0, 1, 2, 3, 4
TEX_ALBEDO, TEX_EDGES, TEX_CHECKER, TEX_BITMAP = 0, 1, 2, 3
ACCEL_KD_SIMD, ACCEL_KD_SCALAR = 0, 1
C_RAYS, C_HITS, C_NODES, C_BOXPASS, C_LEAVES, C_PACKETS, C_TRIS, C_PRIMARY, C_COUNT = range(9)
COUNTER_NAMES = ["rays", "hits", "nodes", "boxpass", "leaves", "packets", "tris", "primary"]


def _cpu_flags() -> set:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def native_width() -> int:
    """Packet width the reference would pick on this host (native_simd<float>::size())."""
    fl = _cpu_flags()
    if {"avx512f", "avx512vl", "avx512bw", "avx512dq"} <= fl:
        return 16
    return 8


def build(force: bool = False) -> None:
    """Compile the oracle flavours with gcc (no GPU needed)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, n))
        for n in ("liboracle_avx2.so", "liboracle_avx512.so", "liboracle_avx2_fast.so", "liboracle_avx512_fast.so")
    )
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


class _SceneDesc(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_int32),
        ("mesh_material", C.POINTER(C.c_int32)),
        ("mesh_nverts", C.POINTER(C.c_int32)),
        ("mesh_ntris", C.POINTER(C.c_int32)),
        ("vertices", C.POINTER(C.c_float)),
        ("indices", C.POINTER(C.c_uint32)),
        ("n_materials", C.c_int32),
        ("mat_kind", C.POINTER(C.c_int32)),
        ("mat_albedo", C.POINTER(C.c_float)),
        ("mat_ior", C.POINTER(C.c_float)),
        ("mat_smooth", C.POINTER(C.c_int32)),
        ("mat_texture", C.POINTER(C.c_int32)),
        ("uvs", C.POINTER(C.c_float)),
        ("mesh_has_uvs", C.POINTER(C.c_int32)),
        ("n_textures", C.c_int32),
        ("tex_kind", C.POINTER(C.c_int32)),
        ("tex_color_a", C.POINTER(C.c_float)),
        ("tex_color_b", C.POINTER(C.c_float)),
        ("tex_param", C.POINTER(C.c_float)),
        ("n_lights", C.c_int32),
        ("light_pos", C.POINTER(C.c_float)),
        ("light_intensity", C.POINTER(C.c_float)),
        ("cam_pos", C.c_float * 3),
        ("cam_mat", C.c_float * 9),
        ("background", C.c_float * 3),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("bucket_size", C.c_int32),
        ("tex_pixels", C.POINTER(C.c_uint8)),
        ("tex_bitmap", C.POINTER(C.c_int32)),
    ]


class _RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("spp", C.c_int32),
        ("max_depth", C.c_int32),
        ("diffuse_rays", C.c_int32),
        ("seed", C.c_uint32),
        ("fov_degrees", C.c_double),
        ("shadow_bias", C.c_float),
        ("reflection_bias", C.c_float),
        ("refraction_bias", C.c_float),
        ("n_threads", C.c_int32),
        ("count_work", C.c_int32),
    ]


HIT_DTYPE = np.dtype(
    [("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4"), ("mesh", "<u4"), ("normal", "<f4", (3,))]
)
assert HIT_DTYPE.itemsize == 32

_libs: dict = {}


def _lib(fast: bool = False, isa: str | None = None):
    isa = isa or ("avx512" if native_width() == 16 else "avx2")
    key = (isa, fast)
    if key in _libs:
        return _libs[key]
    build()
    path = os.path.join(_HERE, f"liboracle_{isa}{'_fast' if fast else ''}.so")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ora_scene_create.restype = vp
    L.ora_scene_create.argtypes = [C.POINTER(_SceneDesc)]
    L.ora_scene_destroy.argtypes = [vp]
    L.ora_accel_build.restype = vp
    L.ora_accel_build.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.ora_accel_destroy.argtypes = [vp]
    for n in ("ora_accel_num_nodes", "ora_accel_num_packets", "ora_accel_num_leaf_refs", "ora_accel_num_triangles"):
        getattr(L, n).restype = C.c_int64
        getattr(L, n).argtypes = [vp]
    L.ora_accel_dump.argtypes = [vp, vp, vp, vp]
    L.ora_scene_vertex_normals.argtypes = [vp, C.c_int, vp]
    L.ora_intersect.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.ora_render_frame.restype = C.c_int
    L.ora_render_frame.argtypes = [vp, C.POINTER(_RenderParams), vp, vp]
    L.ora_camera_rays.restype = C.c_int
    L.ora_camera_rays.argtypes = [vp, C.POINTER(_RenderParams), C.c_int, vp]
    L.ora_write_ppm.restype = C.c_size_t
    L.ora_write_ppm.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.ora_root_key.restype = C.c_uint32
    L.ora_root_key.argtypes = [C.c_uint32] * 3
    L.ora_child_key.restype = C.c_uint32
    L.ora_child_key.argtypes = [C.c_uint32] * 2
    L.ora_urand_key.restype = C.c_float
    L.ora_urand_key.argtypes = [C.c_uint32] * 2
    L.ora_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    _libs[key] = L
    return L


_KIND = {"diffuse": MAT_DIFFUSE, "reflective": MAT_REFLECTIVE, "refractive": MAT_REFRACTIVE, "constant": MAT_CONSTANT}


@dataclass
class FlatScene:
    """The reference's scene<float> flattened to arrays (io/json/loader.hpp:235-265).

    Every number goes double -> float32 exactly as loader.hpp:9-17 does."""

    mesh_material: np.ndarray
    mesh_nverts: np.ndarray
    mesh_ntris: np.ndarray
    vertices: np.ndarray  # [sum nverts, 3] f32
    indices: np.ndarray  # [sum ntris, 3] u32 (mesh-local)
    mat_kind: np.ndarray
    mat_albedo: np.ndarray
    mat_ior: np.ndarray
    mat_smooth: np.ndarray
    light_pos: np.ndarray
    light_intensity: np.ndarray
    cam_pos: np.ndarray
    cam_mat: np.ndarray
    background: np.ndarray
    width: int
    height: int
    bucket_size: int
    mat_texture: np.ndarray = None
    uvs: np.ndarray = None            # concatenated [., 2] of the meshes with uvs
    mesh_has_uvs: np.ndarray = None
    tex_kind: np.ndarray = None
    tex_color_a: np.ndarray = None
    tex_color_b: np.ndarray = None
    tex_param: np.ndarray = None
    tex_pixels: np.ndarray = None     # uint8, the decoded RGB bytes of all bitmap textures, concatenated
    tex_bitmap: np.ndarray = None     # [n_textures, 3] int32: byte offset into tex_pixels, width, height
    extra: dict = field(default_factory=dict)


def resolve_texture_path(scene_path: str, file_path: str) -> str:
    """The reference opens `file_path` relative to the process's working directory, which its README asks to be the project
    root (README.md:32-35).  Here: as given if it exists, else below the nearest ancestor of the scene file that holds it."""
    if os.path.exists(file_path):
        return file_path
    d = os.path.dirname(os.path.abspath(scene_path))
    while True:
        cand = os.path.join(d, file_path)
        if os.path.exists(cand):
            return cand
        # the fixtures keep the reference's scenes/ tree without its project root: scenes/hw12/textures/x.jpg -> hw12/textures/x.jpg
        parts = file_path.replace("\\", "/").split("/")
        for k in range(1, len(parts)):
            cand = os.path.join(d, *parts[k:])
            if os.path.exists(cand):
                return cand
        nd = os.path.dirname(d)
        if nd == d:
            raise FileNotFoundError(file_path)
        d = nd


def load_crtscene(path: str, bitmaps: bool = True) -> FlatScene:
    """Independent Python reader for .crtscene (used only to feed the oracle).

    bitmaps=False: a bitmap texture raises NotImplementedError instead of being decoded (oracle/stb_jpeg.py)."""
    with open(path) as f:
        doc = json.load(f)
    st = doc["settings"]
    img = st["image_settings"]
    tex_names, tkind, ta, tb, tp, tbitmap = [], [], [], [], [], []
    tpix, tbmp, tfail = [], [], []
    for t in doc.get("textures", []) if isinstance(doc.get("textures", []), list) else []:      # loader.hpp:78-106
        ty = t["type"]
        tex_names.append(t["name"])
        tbitmap.append(ty == "bitmap")
        if ty == "albedo":
            tkind.append(TEX_ALBEDO); ta.append(t["albedo"][:3]); tb.append([0, 0, 0]); tp.append(0.0)
        elif ty == "edges":
            tkind.append(TEX_EDGES); ta.append(t["edge_color"][:3]); tb.append(t["inner_color"][:3]); tp.append(t["edge_width"])
        elif ty == "checker":
            tkind.append(TEX_CHECKER); ta.append(t["color_A"][:3]); tb.append(t["color_B"][:3]); tp.append(t["square_size"])
        elif ty == "bitmap":                                              # loader.hpp:97-101, texture/bitmap.hpp:11-37
            tkind.append(TEX_BITMAP); ta.append([0, 0, 0]); tb.append([0, 0, 0]); tp.append(0.0)
        else:
            raise ValueError("texture type unknown")  # loader.hpp:104
        tfail.append(None)
        if ty == "bitmap" and bitmaps:
            from . import stb_jpeg
            try:                                  # a file that cannot be decoded is an error once a material uses the texture
                px = stb_jpeg.load(resolve_texture_path(path, t["file_path"]))
                if px.shape[2] != 3:
                    raise NotImplementedError("bitmap.hpp:26-28 reads three channels per pixel")
            except (OSError, ValueError, NotImplementedError) as e:
                tfail[-1] = e
                tkind[-1] = TEX_ALBEDO
                tbmp.append([0, 0, 0])
                continue
            tbmp.append([sum(len(x) for x in tpix), px.shape[1], px.shape[0]])
            tpix.append(px.reshape(-1))
        else:
            tbmp.append([0, 0, 0])
    mats = doc["materials"]
    kinds, alb, ior, smooth, mtex = [], [], [], [], []
    for m in mats:
        t = m["type"]
        if t not in _KIND:
            raise ValueError("material type unknown")  # loader.hpp:145
        tex = -1
        if t == "diffuse" and isinstance(m["albedo"], str):          # texture_material, loader.hpp:120-125
            tex = tex_names.index(m["albedo"])
            if tbitmap[tex] and not bitmaps:
                raise NotImplementedError("bitmap texture (decode with bitmaps=True)")
            if tfail[tex] is not None:
                raise tfail[tex]
            kinds.append(MAT_TEXTURE)
            alb.append([0, 0, 0])
        else:
            kinds.append(_KIND[t])
            alb.append(list(m.get("albedo", [0, 0, 0]))[:3] if t != "refractive" else [0, 0, 0])
        mtex.append(tex)
        ior.append(m.get("ior", 1.0) if t == "refractive" else 1.0)
        smooth.append(1 if m["smooth_shading"] else 0)
    mm, nv, nt, verts, idx, uvl, has_uv = [], [], [], [], [], [], []
    for o in doc["objects"]:
        v = np.asarray(o["vertices"], dtype=np.float64)
        if v.size % 3:
            raise ValueError("vertex coordinates not multiple of 3")  # loader.hpp:170
        t = np.asarray(o["triangles"], dtype=np.int64)
        if t.size % 3:
            raise ValueError("triangle indices not multiple of 3")  # loader.hpp:224
        mm.append(int(o["material_index"]))
        nv.append(v.size // 3)
        nt.append(t.size // 3)
        verts.append(v.reshape(-1, 3).astype(np.float32))
        idx.append(t.reshape(-1, 3).astype(np.uint32))
        uv = np.asarray(o.get("uvs", []), dtype=np.float64)
        if uv.size:                                                       # loader.hpp:173-192: (u, v, ignored) triples
            if uv.size % 3:
                raise ValueError("uv coordinates not multiple of 3")
            uvl.append(uv.reshape(-1, 3)[: v.size // 3, :2].astype(np.float32))
            has_uv.append(1)
        else:
            has_uv.append(0)
    lights = doc["lights"]
    return FlatScene(
        mesh_material=np.asarray(mm, np.int32),
        mesh_nverts=np.asarray(nv, np.int32),
        mesh_ntris=np.asarray(nt, np.int32),
        vertices=np.ascontiguousarray(np.concatenate(verts) if verts else np.zeros((0, 3), np.float32)),
        indices=np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros((0, 3), np.uint32)),
        mat_kind=np.asarray(kinds, np.int32),
        mat_albedo=np.asarray(alb, np.float64).astype(np.float32).reshape(-1, 3),
        mat_ior=np.asarray(ior, np.float64).astype(np.float32),
        mat_smooth=np.asarray(smooth, np.int32),
        light_pos=np.asarray([l["position"][:3] for l in lights], np.float64).astype(np.float32).reshape(-1, 3),  # load_vec3 reads at(0..2), loader.hpp:19-26 (hw15/scene1 has a 4-component position)
        light_intensity=np.asarray([l["intensity"] for l in lights], np.float64).astype(np.float32),
        cam_pos=np.asarray(doc["camera"]["position"][:3], np.float64).astype(np.float32),
        cam_mat=np.asarray(doc["camera"]["matrix"][:9], np.float64).astype(np.float32),
        background=np.asarray(st["background_color"][:3], np.float64).astype(np.float32),
        width=int(img["width"]),
        height=int(img["height"]),
        bucket_size=int(img.get("bucket_size", 64)),  # loader.hpp:48
        mat_texture=np.asarray(mtex, np.int32),
        uvs=np.ascontiguousarray(np.concatenate(uvl) if uvl else np.zeros((0, 2), np.float32)),
        mesh_has_uvs=np.asarray(has_uv, np.int32),
        tex_kind=np.asarray(tkind, np.int32),
        tex_color_a=np.asarray(ta, np.float64).astype(np.float32).reshape(-1, 3),
        tex_color_b=np.asarray(tb, np.float64).astype(np.float32).reshape(-1, 3),
        tex_param=np.asarray(tp, np.float64).astype(np.float32),
        tex_pixels=np.ascontiguousarray(np.concatenate(tpix) if tpix else np.zeros(0, np.uint8)),
        tex_bitmap=np.asarray(tbmp, np.int32).reshape(-1, 3),
    )


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Scene:
    def __init__(self, flat: FlatScene, fast: bool = False, isa: str | None = None):
        self.flat = flat
        self.L = _lib(fast, isa)
        d = _SceneDesc()
        self._keep = flat
        d.n_meshes = len(flat.mesh_material)
        d.mesh_material = _p(flat.mesh_material, C.c_int32)
        d.mesh_nverts = _p(flat.mesh_nverts, C.c_int32)
        d.mesh_ntris = _p(flat.mesh_ntris, C.c_int32)
        d.vertices = _p(flat.vertices, C.c_float)
        d.indices = _p(flat.indices, C.c_uint32)
        d.n_materials = len(flat.mat_kind)
        d.mat_kind = _p(flat.mat_kind, C.c_int32)
        d.mat_albedo = _p(flat.mat_albedo, C.c_float)
        d.mat_ior = _p(flat.mat_ior, C.c_float)
        d.mat_smooth = _p(flat.mat_smooth, C.c_int32)
        self._tex = dict(
            mt=flat.mat_texture if flat.mat_texture is not None else np.full(len(flat.mat_kind), -1, np.int32),
            uv=flat.uvs if flat.uvs is not None else np.zeros((0, 2), np.float32),
            hu=flat.mesh_has_uvs if flat.mesh_has_uvs is not None else np.zeros(len(flat.mesh_material), np.int32),
            tk=flat.tex_kind if flat.tex_kind is not None else np.zeros(0, np.int32),
            ta=flat.tex_color_a if flat.tex_color_a is not None else np.zeros((0, 3), np.float32),
            tb=flat.tex_color_b if flat.tex_color_b is not None else np.zeros((0, 3), np.float32),
            tp=flat.tex_param if flat.tex_param is not None else np.zeros(0, np.float32),
            px=flat.tex_pixels if flat.tex_pixels is not None else np.zeros(0, np.uint8),
            bm=flat.tex_bitmap if flat.tex_bitmap is not None else np.zeros((len(flat.tex_kind) if flat.tex_kind is not None else 0, 3), np.int32),
        )
        self._tex = {k: np.ascontiguousarray(v) for k, v in self._tex.items()}
        d.mat_texture, d.uvs, d.mesh_has_uvs = _p(self._tex["mt"], C.c_int32), _p(self._tex["uv"], C.c_float), _p(self._tex["hu"], C.c_int32)
        d.n_textures = len(self._tex["tk"])
        d.tex_kind, d.tex_color_a = _p(self._tex["tk"], C.c_int32), _p(self._tex["ta"], C.c_float)
        d.tex_color_b, d.tex_param = _p(self._tex["tb"], C.c_float), _p(self._tex["tp"], C.c_float)
        d.tex_pixels, d.tex_bitmap = _p(self._tex["px"], C.c_uint8), _p(self._tex["bm"], C.c_int32)
        d.n_lights = len(flat.light_intensity)
        d.light_pos = _p(flat.light_pos, C.c_float)
        d.light_intensity = _p(flat.light_intensity, C.c_float)
        d.cam_pos[:] = flat.cam_pos.tolist()
        d.cam_mat[:] = flat.cam_mat.tolist()
        d.background[:] = flat.background.tolist()
        d.width, d.height, d.bucket_size = flat.width, flat.height, flat.bucket_size
        self.h = self.L.ora_scene_create(C.byref(d))

    def vertex_normals(self, mesh: int) -> np.ndarray:
        out = np.zeros((int(self.flat.mesh_nverts[mesh]), 3), np.float32)
        self.L.ora_scene_vertex_normals(self.h, mesh, out.ctypes.data)
        return out

    def __del__(self):
        try:
            self.L.ora_scene_destroy(self.h)
        except Exception:
            pass


class Accel:
    """kind: ACCEL_KD_SIMD (kd_tree_simd.hpp) or ACCEL_KD_SCALAR (kd_tree.hpp)."""

    def __init__(self, scene: Scene, kind: int = ACCEL_KD_SIMD, eps: float = 1e-6, max_depth: int = 8,
                 max_leaf: int | None = None, W: int | None = None):
        self.scene = scene
        self.L = scene.L
        self.kind = kind
        if max_leaf is None:
            max_leaf = 64 if kind == ACCEL_KD_SIMD else 16  # kd_tree_simd.hpp:66 / kd_tree.hpp:13
        self.W = W or native_width()
        self.h = self.L.ora_accel_build(scene.h, kind, np.float32(eps), max_depth, max_leaf, self.W)
        if not self.h:
            raise ValueError("ora_accel_build failed")

    @property
    def num_nodes(self):
        return self.L.ora_accel_num_nodes(self.h)

    @property
    def num_packets(self):
        return self.L.ora_accel_num_packets(self.h)

    @property
    def num_leaf_refs(self):
        return self.L.ora_accel_num_leaf_refs(self.h)

    @property
    def num_triangles(self):
        return self.L.ora_accel_num_triangles(self.h)

    def dump(self):
        n = self.num_nodes
        box = np.zeros((n, 6), np.float32)
        link = np.zeros((n, 4), np.int32)
        refs = np.zeros((self.num_leaf_refs,), np.int32)
        self.L.ora_accel_dump(self.h, box.ctypes.data, link.ctypes.data, refs.ctypes.data)
        return box, link, refs

    def intersect(self, rays: np.ndarray, cull: bool, counters: np.ndarray | None = None) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((rays.shape[0],), HIT_DTYPE)
        cp = counters.ctypes.data if counters is not None else None
        self.L.ora_intersect(self.h, rays.ctypes.data, rays.shape[0], 1 if cull else 0, out.ctypes.data, cp)
        return out

    def render(self, width=0, height=0, spp=1, max_depth=5, diffuse_rays=0, seed=42, fov_degrees=90.0,
               shadow_bias=1e-4, reflection_bias=1e-4, refraction_bias=1e-4, n_threads=0, count_work=True):
        p = _RenderParams(width, height, spp, max_depth, diffuse_rays, seed, fov_degrees,
                          np.float32(shadow_bias), np.float32(reflection_bias), np.float32(refraction_bias), n_threads,
                          1 if count_work else 0)
        w = width or self.scene.flat.width
        h = height or self.scene.flat.height
        rgb = np.zeros((h, w, 3), np.float32)
        cn = np.zeros((C_COUNT,), np.uint64)
        rc = self.L.ora_render_frame(self.h, C.byref(p), rgb.ctypes.data, cn.ctypes.data)
        if rc != 0:
            raise ValueError("ora_render_frame: bad parameters")
        return rgb, dict(zip(COUNTER_NAMES, (int(x) for x in cn)))

    def camera_rays(self, width=0, height=0, spp=1, seed=42, fov_degrees=90.0, sample=0) -> np.ndarray:
        """[h, w, 6] origin + direction of every pixel's camera ray (render.hpp:35-62)."""
        p = _RenderParams(width, height, spp, 5, 0, seed, fov_degrees, np.float32(1e-4), np.float32(1e-4), np.float32(1e-4), 1, 0)
        w = width or self.scene.flat.width
        h = height or self.scene.flat.height
        rays = np.zeros((h, w, 6), np.float32)
        if self.L.ora_camera_rays(self.h, C.byref(p), sample, rays.ctypes.data) != 0:
            raise ValueError("ora_camera_rays: bad parameters")
        return rays

    def __del__(self):
        try:
            self.L.ora_accel_destroy(self.h)
        except Exception:
            pass


def write_ppm(rgb: np.ndarray) -> bytes:
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    L = _lib()
    n = L.ora_write_ppm(rgb.ctypes.data, w, h, None, 0)
    buf = C.create_string_buffer(n)
    L.ora_write_ppm(rgb.ctypes.data, w, h, buf, n)
    return buf.raw[:n]


def root_key(seed, pixel, sample) -> int:
    return int(_lib().ora_root_key(seed, pixel, sample))


def child_key(key, child) -> int:
    return int(_lib().ora_child_key(key, child))


def urand_key(key, j) -> float:
    return float(_lib().ora_urand_key(key, j))


def sincos(angle: float):
    s, c = C.c_float(), C.c_float()
    _lib().ora_sincos(np.float32(angle), C.byref(s), C.byref(c))
    return s.value, c.value
