"""bitmap_texture (scene/texture/bitmap.hpp): the JPEG decoder behind `stbi_load` and the texel lookup.

stb_image is not part of the reference tree (CMakeLists.txt:17-21 fetches it), so csrc/jpeg.cpp (product, C++) and
oracle/stb_jpeg.py (checker, Python) are two independent restatements of its published algorithm.  The reference-held pin is
outputs/textures.png (tests/test_reference_outputs.py); here the two restatements must agree byte for byte, also on the JPEG
layouts no reference file uses (tests/golden/jpeg/, written by tools/make_jpeg_fixtures.py: parity unpinned against stb itself)."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES

JPEGS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg")))
DRAGON = os.path.join(SCENES, "hw12", "textures", "dragon.jpg")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_fixture_set():
    assert len(JPEGS) == 9 and os.path.exists(DRAGON)


def test_product_decoder_equals_oracle_decoder_on_the_reference_texture(rtk, ora):
    from oracle import stb_jpeg
    data = open(DRAGON, "rb").read()
    a, b = rtk.decode_jpeg(data), stb_jpeg.decode(data)
    assert a.shape == (360, 540, 3) and a.dtype == np.uint8
    assert np.array_equal(a, b)


@pytest.mark.parametrize("path", JPEGS, ids=lambda p: os.path.basename(p)[:-4])
def test_decoders_agree_on_other_jpeg_layouts(rtk, path):
    from oracle import stb_jpeg
    data = open(path, "rb").read()
    if "refused" in path:                                   # progressive files: stb decodes them, this restatement does not
        with pytest.raises(rtk.RtkError) as e:
            rtk.decode_jpeg(data)
        assert e.value.code == rtk.RTK_ERR_UNSUPPORTED
        with pytest.raises(ValueError):
            stb_jpeg.decode(data)
        return
    a, b = rtk.decode_jpeg(data), stb_jpeg.decode(data)
    assert np.array_equal(a, b)
    lib = np.load(path[:-4] + ".libjpeg.npy").astype(np.int32)          # libjpeg's decode: another IDCT and upsampler, so only close
    d = np.abs(a.astype(np.int32) - lib)
    assert a.shape == lib.shape and np.mean(d <= 6) > 0.97, (d.max(), np.mean(d <= 6))


def test_decoder_rejects_garbage(rtk):
    for data in (b"", b"\xff\xd8", b"\xff\xd8\xff\xd9", b"not a jpeg at all", open(DRAGON, "rb").read()[:700]):
        with pytest.raises(rtk.RtkError):
            rtk.decode_jpeg(data)
    data = bytearray(open(DRAGON, "rb").read())
    rtk.decode_jpeg(bytes(data[:30000]))                    # a truncated scan decodes (zero bits after the end, as in stb_image)


def test_reader_loads_the_bitmap_like_the_oracle_reader(rtk, ora):
    path = os.path.join(SCENES, "hw12", "scene4.crtscene")
    arr = rtk.parse_scene_file(path).arrays()
    flat = ora.load_crtscene(path)
    assert list(arr["tex_kind"]) == [rtk.TEX_ALBEDO, rtk.TEX_EDGES, rtk.TEX_CHECKER, rtk.TEX_BITMAP]
    assert np.array_equal(arr["tex_bitmap"], flat.tex_bitmap) and list(arr["tex_bitmap"][3]) == [0, 540, 360]
    assert np.array_equal(arr["tex_pixels"], flat.tex_pixels)
    # the flattened description round-trips through rtk_scene_create
    sc2 = rtk.Scene.from_arrays(**{k: arr[k] for k in (
        "mesh_material", "mesh_nverts", "mesh_ntris", "vertices", "indices", "mat_kind", "mat_albedo", "mat_ior", "mat_smooth",
        "light_pos", "light_intensity", "cam_pos", "cam_mat", "background", "width", "height", "bucket_size", "mat_texture", "uvs",
        "mesh_has_uvs", "tex_kind", "tex_color_a", "tex_color_b", "tex_param", "tex_pixels", "tex_bitmap")})
    arr2 = sc2.arrays()
    assert np.array_equal(arr2["tex_pixels"], arr["tex_pixels"]) and np.array_equal(arr2["tex_bitmap"], arr["tex_bitmap"])
    with pytest.raises(rtk.RtkError):
        rtk.Scene.from_arrays(**{**{k: arr[k] for k in (
            "mesh_material", "mesh_nverts", "mesh_ntris", "vertices", "indices", "mat_kind", "mat_albedo", "mat_ior", "mat_smooth",
            "light_pos", "light_intensity", "cam_pos", "cam_mat", "background", "width", "height", "bucket_size", "mat_texture",
            "uvs", "mesh_has_uvs", "tex_kind", "tex_color_a", "tex_color_b", "tex_param")}})       # bitmap kind without texels


def _uv_stress_scene(tmp_path):
    """hw12/scene4's bitmap quad with uvs outside [0, 1] (negative, > 1, huge), a second tiny bitmap, a mirror: the clamps of
    bitmap.hpp:53-57 and the texel addressing, seen directly, through GI rays and through a reflection."""
    doc = json.load(open(os.path.join(SCENES, "hw12", "scene4.crtscene")))
    doc["textures"][3]["file_path"] = DRAGON
    doc["textures"].append({"name": "tiny", "type": "bitmap", "file_path": os.path.join(ROOT, "tests", "golden", "jpeg", "s420_two_wide.jpg")})
    doc["materials"].append({"type": "diffuse", "albedo": "tiny", "smooth_shading": False})
    doc["materials"].append({"type": "reflective", "albedo": [1, 1, 1], "smooth_shading": False})
    o = doc["objects"]
    o[3]["uvs"] = [-0.5, -0.25, 0, 1.75, -0.25, 0, 1.75, 1.5, 0, -0.5, 1.5, 0]
    o[0]["material_index"] = 4
    o[0]["uvs"] = [-3e9, 0, 0, 3e9, 0, 0, 3e9, 1, 0, -3e9, 1, 0]
    o[1]["material_index"] = 3
    o.append({"material_index": 5, "vertices": [-8, -3, -6, 8, -3, -6, 8, -3, 6, -8, -3, 6], "triangles": [0, 2, 1, 0, 3, 2]})
    path = tmp_path / "uvstress.crtscene"
    path.write_text(json.dumps(doc))
    return str(path)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["hw12/scene3", "hw12/scene4", "uvstress"])
def test_bitmap_scenes_bit_exact_through_every_engine(rtk, ora, tmp_path, scene):
    path = _uv_stress_scene(tmp_path) if scene == "uvstress" else os.path.join(SCENES, scene + ".crtscene")
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path))
    oacc = ora.Accel(ora.Scene(ora.load_crtscene(path)), ora.ACCEL_KD_SIMD)
    for (w, h, spp, depth, gi) in [(640, 360, 1, 5, 0), (200, 112, 3, 3, 2)]:
        ref, ocn = oacc.render(w, h, spp, depth, gi)
        assert len(np.unique(ref.reshape(-1, 3), axis=0)) > 200                  # the picture is in the frame
        for mode in (rtk.TRACE_AUTO, rtk.TRACE_GROUP4, rtk.TRACE_STREAM, rtk.TRACE_LANE):
            rgb, cn = acc.render_frame(rtk.RenderConfig(width=w, height=h, spp=spp, max_ray_depth=depth, diffuse_rays=gi, trace_mode=mode))
            assert cn["rays"] == ocn["rays"], mode
            assert np.array_equal(_bits(rgb), _bits(ref)), mode
