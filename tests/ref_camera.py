"""numpy-float32 transcription of the reference's camera-ray expression (render/render.hpp:26, :35-62), written from
the reference source alone -- NOT from the oracle or the engine -- so that both can be checked against it:

    const F aspect_ratio = static_cast<F>(image_width) / image_height;                        :26
    F raster_x = x;  raster_x += static_cast<F>(0.5);                                          :37-41  (spp == 1)
    const F ndc_x = raster_x / image_width;                                                    :47
    F screen_x = (static_cast<F>(2.) * ndc_x) - static_cast<F>(1.);                            :50
    F screen_y = static_cast<F>(1.) - (static_cast<F>(2.) * ndc_y);                            :51
    screen_x *= aspect_ratio;                                                                  :53
    const F fov_radians = degrees_to_radians(fov_degrees);    // double product, rounded to F  :55, utils/convert.hpp:4-6
    screen_x *= std::tan(fov_radians / static_cast<F>(2.));   // tanf                          :56
    screen_y *= std::tan(fov_radians / static_cast<F>(2.));                                    :57
    direction = normalized(transpose(camera.matrix) * vec3{screen_x, screen_y, -1});           :59-60, mat3.hpp:34-41, :53-60, vec3.hpp:104-108

Every operation is a float32 numpy operation (IEEE, round to nearest, no fused multiply-add); tanf is the C library's.
"""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]


def camera_directions(width: int, height: int, fov_degrees: float, cam_matrix) -> np.ndarray:
    """[h, w, 3] float32 ray directions of the pixel centres (samples_per_pixel == 1)."""
    m = np.asarray(cam_matrix, f32).reshape(3, 3)
    aspect = f32(width) / f32(height)
    x = np.arange(width, dtype=f32)[None, :] + f32(0.5)
    y = np.arange(height, dtype=f32)[:, None] + f32(0.5)
    ndc_x = x / f32(width)
    ndc_y = y / f32(height)
    sx = (f32(2.0) * ndc_x) - f32(1.0)
    sy = f32(1.0) - (f32(2.0) * ndc_y)
    sx = sx * aspect
    fov_radians = f32(np.float64(fov_degrees) * (np.float64(np.pi) / np.float64(180.0)))
    t = f32(_libm.tanf(ctypes.c_float(float(fov_radians / f32(2.0)))))
    sx = sx * t
    sy = sy * t
    sx = np.broadcast_to(sx, (height, width)).astype(f32)
    sy = np.broadcast_to(sy, (height, width)).astype(f32)
    sz = np.full((height, width), f32(-1.0), f32)
    mt = m.T                                                   # transpose(camera.matrix)
    d = [(mt[i, 0] * sx + mt[i, 1] * sy) + mt[i, 2] * sz for i in range(3)]
    length = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    inv = f32(1.0) / length
    return np.stack([d[0] * inv, d[1] * inv, d[2] * inv], axis=-1).astype(f32)
