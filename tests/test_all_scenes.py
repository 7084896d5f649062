"""Every scene file of the reference that its own loader accepts (lights + materials present): host
logic on CPU (reader + tree == oracle) and frames on the GPU (bit-exact vs the oracle through three engines).
The scene files are the reference's input data, copied under tests/golden/scenes."""
import glob
import os

import numpy as np
import pytest

from conftest import SCENES

ALL = sorted(glob.glob(os.path.join(SCENES, "*", "*.crtscene")))
# the reference's loader throws on these (io/json/loader.hpp:246 `for (auto light : doc["lights"])`, :257 `doc["materials"]`:
# iterating a missing key raises simdjson_error)
NO_LIGHTS = tuple(f"hw07/scene{i}.crtscene" for i in range(5)) + ("hw09/scene0.crtscene",)
NO_MATERIALS = tuple(f"hw08/scene{i}.crtscene" for i in range(4)) + ("hw15/scene0.crtscene",)
BITMAP = ("hw12/scene3.crtscene", "hw12/scene4.crtscene")                  # use the JPEG texture (scenes/hw12/textures/dragon.jpg)
LOADABLE = [p for p in ALL if not p.endswith(NO_LIGHTS + NO_MATERIALS)]
IDS = [os.path.relpath(p, SCENES)[:-len(".crtscene")] for p in LOADABLE]


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_scene_set_is_complete():
    assert len(LOADABLE) == 24 and len(ALL) == 35            # all 35 scene files of the reference
    assert sum(p.endswith(BITMAP) for p in LOADABLE) == 2


@pytest.mark.parametrize("path", [p for p in ALL if p.endswith(NO_LIGHTS + NO_MATERIALS)], ids=lambda p: os.path.basename(os.path.dirname(p)) + "/" + os.path.basename(p))
def test_scenes_the_reference_loader_rejects_are_rejected(rtk, path):
    """The early homework scenes have no "lights" (hw07/*, hw09/scene0) or no "materials" (hw08/*, hw15/scene0): the
    reference's parse_scene_file throws on the missing key (loader.hpp:246, :257); the C-ABI reports RTK_ERR_PARSE instead
    of inventing a light or a material."""
    with pytest.raises(rtk.RtkError) as e:
        rtk.parse_scene_file(path)
    assert e.value.code == rtk.RTK_ERR_PARSE


@pytest.mark.parametrize("path", LOADABLE, ids=IDS)
def test_host_side_matches_oracle(rtk, ora, path):
    sc = rtk.parse_scene_file(path)
    flat = ora.load_crtscene(path)
    arr = sc.arrays()
    for k in ("mesh_material", "mesh_nverts", "mesh_ntris", "vertices", "indices", "mat_kind", "mat_albedo", "mat_ior",
              "mat_smooth", "light_pos", "light_intensity", "cam_pos", "cam_mat", "background", "mat_texture", "mesh_has_uvs",
              "uvs", "tex_kind", "tex_color_a", "tex_color_b", "tex_param", "tex_bitmap", "tex_pixels"):
        assert np.array_equal(_bits(arr[k]), _bits(getattr(flat, k))), k
    box, link, refs = rtk.KdTreeSimdAccel(sc).tree_dump()
    obox, olink, orefs = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD, W=16).dump()
    assert np.array_equal(_bits(box), _bits(obox)) and np.array_equal(link, olink) and np.array_equal(refs, orefs)


@pytest.mark.gpu
@pytest.mark.parametrize("path", LOADABLE, ids=IDS)
def test_frames_match_oracle(rtk, ora, path):
    """Gate A on every loadable scene: 480x270, 1 spp, depth 5, through the default engine, the megakernel and the
    streaming pipeline; plus one GI + multi-sample frame."""
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(path))
    oacc = ora.Accel(ora.Scene(ora.load_crtscene(path)), ora.ACCEL_KD_SIMD)
    ref, ocn = oacc.render(480, 270, 1, 5, 0)
    for mode in (rtk.TRACE_AUTO, rtk.TRACE_GROUP4, rtk.TRACE_STREAM):
        rgb, cn = acc.render_frame(rtk.RenderConfig(width=480, height=270, max_ray_depth=5, trace_mode=mode))
        assert cn["rays"] == ocn["rays"], mode
        assert np.array_equal(_bits(rgb), _bits(ref)), mode
    ref, ocn = oacc.render(160, 90, 3, 4, 2)
    for mode in (rtk.TRACE_AUTO, rtk.TRACE_GROUP4):
        rgb, cn = acc.render_frame(rtk.RenderConfig(width=160, height=90, spp=3, max_ray_depth=4, diffuse_rays=2, trace_mode=mode))
        assert cn["rays"] == ocn["rays"], mode
        assert np.array_equal(_bits(rgb), _bits(ref)), mode


@pytest.mark.gpu
def test_all_procedural_textures_in_one_scene(rtk, ora, tmp_path):
    """hw12/scene4 (four textured quads) with its JPEG material re-pointed at the checker texture and a mirror added
    behind the camera's view, so that albedo / edges / checker are all sampled, directly and through a reflection."""
    import json

    doc = json.load(open(os.path.join(SCENES, "hw12", "scene4.crtscene")))
    doc["materials"][3]["albedo"] = "Black White Checker"
    doc["materials"].append({"type": "reflective", "albedo": [1, 1, 1], "smooth_shading": False})
    doc["objects"].append({"material_index": 4, "vertices": [-8, -3, -6, 8, -3, -6, 8, -3, 6, -8, -3, 6],
                           "triangles": [0, 2, 1, 0, 3, 2]})
    path = tmp_path / "textures.crtscene"
    path.write_text(json.dumps(doc))
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(str(path)))
    oacc = ora.Accel(ora.Scene(ora.load_crtscene(str(path))), ora.ACCEL_KD_SIMD)
    for (w, h, spp, depth, gi) in [(640, 360, 1, 5, 0), (200, 112, 2, 3, 2)]:
        ref, ocn = oacc.render(w, h, spp, depth, gi)
        assert len(np.unique(ref.reshape(-1, 3), axis=0)) > 8          # several texture colours are visible
        for mode in (rtk.TRACE_AUTO, rtk.TRACE_GROUP4, rtk.TRACE_STREAM, rtk.TRACE_LANE):
            rgb, cn = acc.render_frame(rtk.RenderConfig(width=w, height=h, spp=spp, max_ray_depth=depth, diffuse_rays=gi, trace_mode=mode))
            assert cn["rays"] == ocn["rays"], mode
            assert np.array_equal(_bits(rgb), _bits(ref)), mode
