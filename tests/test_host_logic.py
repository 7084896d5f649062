"""Host-side logic of the product against the oracle (no GPU): .crtscene reader, vertex normals, kd-tree build,
tree flattening, PPM writer, error reporting."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import CONFIG_SCENES, SCENE5


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("scene", list(CONFIG_SCENES))
def test_crtscene_reader_matches_python_json(rtk, ora, scene):
    """csrc/crtscene.cpp vs an independent json.load + float32 cast (loader.hpp:9-17 semantics)."""
    sc = rtk.parse_scene_file(CONFIG_SCENES[scene])
    flat = ora.load_crtscene(CONFIG_SCENES[scene])
    arr = sc.arrays()
    for k in ("mesh_material", "mesh_nverts", "mesh_ntris", "vertices", "indices", "mat_kind", "mat_albedo", "mat_ior",
              "mat_smooth", "light_pos", "light_intensity", "cam_pos", "cam_mat", "background"):
        assert np.array_equal(_bits(arr[k]), _bits(getattr(flat, k))), k
    assert (arr["width"], arr["height"], arr["bucket_size"]) == (flat.width, flat.height, flat.bucket_size)


@pytest.mark.parametrize("scene", list(CONFIG_SCENES))
def test_vertex_normals_and_tree_match_oracle(rtk, ora, scene):
    sc = rtk.parse_scene_file(CONFIG_SCENES[scene])
    flat = ora.load_crtscene(CONFIG_SCENES[scene])
    osc = ora.Scene(flat)
    for m in range(len(flat.mesh_material)):
        a, b = sc.vertex_normals(m), osc.vertex_normals(m)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(_bits(np.nan_to_num(a)), _bits(np.nan_to_num(b)))
    for (md, ml) in [(8, 64), (8, 16), (3, 64), (12, 8), (0, 64)]:
        acc = rtk.KdTreeSimdAccel(sc, max_depth=md, max_leaf_size=ml)
        oacc = ora.Accel(osc, ora.ACCEL_KD_SIMD, max_depth=md, max_leaf=ml, W=16)
        box, link, refs = acc.tree_dump()
        obox, olink, orefs = oacc.dump()
        assert np.array_equal(_bits(box), _bits(obox))
        assert np.array_equal(link, olink)
        assert np.array_equal(refs, orefs)
        ti = acc.tree_info()
        assert ti.n_nodes == oacc.num_nodes and ti.n_leaf_refs == oacc.num_leaf_refs
        assert ti.n_triangles == oacc.num_triangles
        assert ti.n_inner + ti.n_leaves == ti.n_nodes


def test_scene_from_arrays_equals_scene_from_file(rtk, ora):
    flat = ora.load_crtscene(SCENE5)
    a = rtk.Scene.from_arrays(flat.mesh_material, flat.mesh_nverts, flat.mesh_ntris, flat.vertices, flat.indices,
                              flat.mat_kind, flat.mat_albedo, flat.mat_ior, flat.mat_smooth, flat.light_pos,
                              flat.light_intensity, flat.cam_pos, flat.cam_mat, flat.background, flat.width, flat.height,
                              flat.bucket_size)
    b = rtk.parse_scene_file(SCENE5)
    ta, tb = rtk.KdTreeSimdAccel(a).tree_dump(), rtk.KdTreeSimdAccel(b).tree_dump()
    for x, y in zip(ta, tb):
        assert np.array_equal(_bits(x), _bits(y))


def test_ppm_bytes_match_oracle(rtk, ora):
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, size=(37, 53, 3)).astype(np.float32)
    img[0, 0] = [0.0, 1.0, 0.5]
    img[0, 1] = [0.999999, 0.00390624, 0.00390626]
    got, ref = rtk.format_ppm(img), ora.write_ppm(img)
    assert got == ref
    assert got.startswith(b"P3\n53 37\n255\n0 255 127\t")
    assert got.count(b"\n") == 3 + 37 and got.count(b"\t") == 37 * 53


def test_write_ppm_file(rtk, tmp_path):
    img = np.full((4, 5, 3), 0.5, np.float32)
    p = tmp_path / "image.ppm"
    rtk.write_ppm(img, str(p))
    assert p.read_bytes() == rtk.format_ppm(img)
    with pytest.raises(rtk.RtkError) as e:
        rtk.write_ppm(img, str(tmp_path / "no_such_dir" / "x.ppm"))
    assert e.value.code == rtk.RTK_ERR_IO


def _write_scene(tmp_path, mutate):
    doc = json.load(open(SCENE5))
    doc["objects"] = doc["objects"][:1]           # keep the file small: the floor quad only
    mutate(doc)
    p = tmp_path / "s.crtscene"
    p.write_text(json.dumps(doc))
    return str(p)


def test_reader_error_reporting(rtk, tmp_path):
    """Malformed scenes: the reference throws (loader.hpp:104,127,145,170,190,224); the C-ABI returns codes."""
    with pytest.raises(rtk.RtkError) as e:
        rtk.parse_scene_file(str(tmp_path / "missing.crtscene"))
    assert e.value.code == rtk.RTK_ERR_IO
    bad = tmp_path / "bad.crtscene"
    bad.write_text('{"settings": ')
    with pytest.raises(rtk.RtkError) as e:
        rtk.parse_scene_file(str(bad))
    assert e.value.code == rtk.RTK_ERR_PARSE

    def expect(mutate, code):
        with pytest.raises(rtk.RtkError) as ei:
            rtk.parse_scene_file(_write_scene(tmp_path, mutate))
        assert ei.value.code == code, ei.value

    expect(lambda d: d["materials"][0].update(type="glossy"), rtk.RTK_ERR_INVALID)            # material type unknown
    expect(lambda d: d["objects"][0]["vertices"].append(1.0), rtk.RTK_ERR_INVALID)            # not multiple of 3
    expect(lambda d: d["objects"][0]["triangles"].append(1), rtk.RTK_ERR_INVALID)
    expect(lambda d: d["objects"][0].update(material_index=9), rtk.RTK_ERR_INVALID)
    expect(lambda d: d["objects"][0]["triangles"].__setitem__(0, 77), rtk.RTK_ERR_INVALID)    # vertex index out of range
    expect(lambda d: d.pop("lights"), rtk.RTK_ERR_PARSE)
    expect(lambda d: d.pop("camera"), rtk.RTK_ERR_PARSE)
    expect(lambda d: d["materials"][1].update(albedo="brick"), rtk.RTK_ERR_INVALID)           # texture that does not exist
    expect(lambda d: (d.update(textures=[{"name": "brick", "type": "bitmap", "file_path": "x.jpg"}]),
                      d["materials"][1].update(albedo="brick")), rtk.RTK_ERR_IO)              # bitmap texture in use, file missing
    notjpeg = tmp_path / "tex.png"
    notjpeg.write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(64))
    expect(lambda d: (d.update(textures=[{"name": "brick", "type": "bitmap", "file_path": str(notjpeg)}]),
                      d["materials"][1].update(albedo="brick")), rtk.RTK_ERR_UNSUPPORTED)     # only baseline JPEG is decoded
    unused = rtk.parse_scene_file(_write_scene(tmp_path, lambda d: d.update(
        textures=[{"name": "brick", "type": "bitmap", "file_path": "x.jpg"}])))               # undecodable but unused: loads
    assert unused.info.n_textures == 1 and unused.info.n_bitmap_bytes == 0
    ok = rtk.parse_scene_file(_write_scene(tmp_path, lambda d: d["settings"]["image_settings"].update(bucket_size=24)))
    assert ok.info.bucket_size == 24 and ok.info.n_triangles == 2
    assert rtk.parse_scene_file(_write_scene(tmp_path, lambda d: None)).info.bucket_size == 64  # loader.hpp:48


def test_scene_create_validates_its_description(rtk, ora):
    flat = ora.load_crtscene(SCENE5)
    bad_idx = flat.indices.copy()
    bad_idx[0, 0] = 10_000
    with pytest.raises(rtk.RtkError):
        rtk.Scene.from_arrays(flat.mesh_material, flat.mesh_nverts, flat.mesh_ntris, flat.vertices, bad_idx, flat.mat_kind,
                              flat.mat_albedo, flat.mat_ior, flat.mat_smooth, flat.light_pos, flat.light_intensity,
                              flat.cam_pos, flat.cam_mat, flat.background, 64, 64)
    with pytest.raises(rtk.RtkError):
        rtk.Scene.from_arrays(np.array([5], np.int32), flat.mesh_nverts[:1], flat.mesh_ntris[:1], flat.vertices[:4],
                              flat.indices[:2], flat.mat_kind, flat.mat_albedo, flat.mat_ior, flat.mat_smooth, flat.light_pos,
                              flat.light_intensity, flat.cam_pos, flat.cam_mat, flat.background, 64, 64)
    sc = rtk.parse_scene_file(SCENE5)
    with pytest.raises(rtk.RtkError):
        rtk.KdTreeSimdAccel(sc, max_depth=99)
    with pytest.raises(rtk.RtkError):
        rtk.KdTreeSimdAccel(sc, max_leaf_size=0)


def test_empty_and_tiny_scenes_build(rtk, ora):
    """Edge cases of the builder: no triangles at all, one triangle, a degenerate (zero-area) triangle."""
    z3 = np.zeros((0, 3), np.float32)
    common = dict(mat_kind=np.array([0], np.int32), mat_albedo=np.ones((1, 3), np.float32), mat_ior=np.ones(1, np.float32),
                  mat_smooth=np.zeros(1, np.int32), light_pos=np.zeros((1, 3), np.float32),
                  light_intensity=np.ones(1, np.float32), cam_pos=np.zeros(3, np.float32),
                  cam_mat=np.eye(3, dtype=np.float32).ravel(), background=np.zeros(3, np.float32), width=8, height=8)
    empty = rtk.Scene.from_arrays(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), z3,
                                  np.zeros((0, 3), np.uint32), **common)
    ti = rtk.KdTreeSimdAccel(empty).tree_info()
    assert (ti.n_nodes, ti.n_leaves, ti.n_leaf_refs) == (1, 1, 0)
    verts = np.array([[-1, -1, -3], [1, -1, -3], [0, 1, -3], [0, 0, -3]], np.float32)
    one = rtk.Scene.from_arrays(np.array([0], np.int32), np.array([4], np.int32), np.array([2], np.int32), verts,
                                np.array([[0, 1, 2], [3, 3, 3]], np.uint32), **common)
    ti = rtk.KdTreeSimdAccel(one, max_leaf_size=1).tree_info()
    assert ti.n_triangles == 2 and ti.n_leaf_refs >= 2
