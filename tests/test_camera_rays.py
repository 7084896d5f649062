"""The camera rays of render_frame (render.hpp:35-62), checked against a numpy-float32 transcription of the reference
expression (tests/ref_camera.py) -- in particular at fields of view other than the 90 degrees the reference compiles in,
where the float rounding of `fov_radians` and the float tanf matter (round-1 finding: engine and oracle both evaluated
them in double and agreed with each other)."""
import numpy as np
import pytest

import ref_camera
from conftest import SCENE2, SCENE5

FOVS = [90.0, 60.0, 120.0, 45.0, 33.3]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("fov", FOVS)
def test_oracle_camera_rays_equal_the_reference_expression(ora, fov):
    for path, (w, h) in ((SCENE5, (640, 360)), (SCENE2, (200, 200))):
        flat = ora.load_crtscene(path)
        acc = ora.Accel(ora.Scene(flat), ora.ACCEL_KD_SIMD)
        rays = acc.camera_rays(w, h, fov_degrees=fov)
        want = ref_camera.camera_directions(w, h, fov, flat.cam_mat)
        assert np.array_equal(_bits(rays[..., 3:]), _bits(want)), (path, fov)
        assert np.array_equal(_bits(rays[..., :3]), _bits(np.broadcast_to(flat.cam_pos, (h, w, 3))))


@pytest.mark.gpu
@pytest.mark.parametrize("fov", FOVS)
def test_device_camera_rays_equal_the_reference_expression(rtk, ora, fov):
    for path, (w, h) in ((SCENE5, (1920, 1080)), (SCENE2, (200, 200))):
        scene = rtk.parse_scene_file(path)
        acc = rtk.KdTreeSimdAccel(scene)
        rays = acc.camera_rays(rtk.RenderConfig(width=w, height=h, fov_degrees=fov))
        want = ref_camera.camera_directions(w, h, fov, scene.arrays()["cam_mat"])
        assert np.array_equal(_bits(rays[..., 3:]), _bits(want)), (path, fov)
        assert np.array_equal(_bits(rays[..., :3]), _bits(np.broadcast_to(scene.arrays()["cam_pos"], (h, w, 3))))


@pytest.mark.gpu
@pytest.mark.parametrize("fov", [60.0, 120.0])
@pytest.mark.parametrize("mode", [3, 6, 1])
def test_frames_at_other_fields_of_view(rtk, ora, fov, mode):
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE5))
    oacc = ora.Accel(ora.Scene(ora.load_crtscene(SCENE5)), ora.ACCEL_KD_SIMD)
    rgb, cn = acc.render_frame(rtk.RenderConfig(width=480, height=270, fov_degrees=fov, trace_mode=mode))
    ref, ocn = oacc.render(480, 270, 1, 5, 0, fov_degrees=fov)
    assert cn["rays"] == ocn["rays"]
    assert np.array_equal(_bits(rgb), _bits(ref))


@pytest.mark.gpu
def test_jittered_camera_rays_are_the_oracles(rtk, ora):
    """spp > 1: the jitter comes from the counter-based RNG shared by engine and oracle (not the reference's racy minstd)."""
    acc = rtk.KdTreeSimdAccel(rtk.parse_scene_file(SCENE5))
    oacc = ora.Accel(ora.Scene(ora.load_crtscene(SCENE5)), ora.ACCEL_KD_SIMD)
    for sample in (0, 3):
        got = acc.camera_rays(rtk.RenderConfig(width=320, height=180, spp=4, fov_degrees=75.0), sample=sample)
        want = oacc.camera_rays(320, 180, spp=4, fov_degrees=75.0, sample=sample)
        assert np.array_equal(_bits(got), _bits(want))
