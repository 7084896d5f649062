"""Every scene file of the reference that its own loader accepts (lights + materials present, no textures): host
logic on CPU (reader + tree == oracle) and frames on the GPU (bit-exact vs the oracle through three engines).
The scene files are the reference's input data, copied under tests/golden/scenes."""
import glob
import os

import numpy as np
import pytest

from conftest import SCENES

ALL = sorted(glob.glob(os.path.join(SCENES, "*", "*.crtscene")))
LOADABLE = [p for p in ALL if not p.endswith(("hw08/scene0.crtscene", "hw15/scene0.crtscene"))]
IDS = [os.path.relpath(p, SCENES)[:-len(".crtscene")] for p in LOADABLE]


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_scene_set_is_complete():
    assert len(LOADABLE) == 19


@pytest.mark.parametrize("path", [p for p in ALL if p not in LOADABLE], ids=lambda p: os.path.basename(os.path.dirname(p)) + "/" + os.path.basename(p))
def test_scenes_the_reference_loader_rejects_are_rejected(rtk, path):
    """hw08/scene0 and hw15/scene0 have no "materials": the reference throws in load_mesh (loader.hpp:151 / :253);
    the C-ABI reports RTK_ERR_PARSE instead of inventing a material."""
    with pytest.raises(rtk.RtkError) as e:
        rtk.parse_scene_file(path)
    assert e.value.code == rtk.RTK_ERR_PARSE


@pytest.mark.parametrize("path", LOADABLE, ids=IDS)
def test_host_side_matches_oracle(rtk, ora, path):
    sc = rtk.parse_scene_file(path)
    flat = ora.load_crtscene(path)
    arr = sc.arrays()
    for k in ("mesh_material", "mesh_nverts", "mesh_ntris", "vertices", "indices", "mat_kind", "mat_albedo", "mat_ior",
              "mat_smooth", "light_pos", "light_intensity", "cam_pos", "cam_mat", "background"):
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
