// Device helpers shared by the megakernel (kernels.hip) and the streaming pipeline (stream.hip).
#pragma once

#include "kernels.hpp"

namespace rtk {
namespace dev {

// Counter-based RNG.  The reference's thread_local minstd_rand (utils/rand.hpp:5-19) is seeded identically on every
// thread and handed out by a racy tile queue, so its stream cannot be reproduced.  Here a draw is
// keyed by the POSITION OF A RAY IN ITS SAMPLE'S RAY TREE: root key = f(seed, absolute pixel, sample); a secondary
// ray's key derives from its parent ray's key and its child index; draw j at a ray is a pure function of (key, j).
// Frames therefore depend neither on the wave / bucket / rank layout nor on the order in which a traversal visits
// the tree (recursive CPU code, the per-lane state machine and the level-by-level wavefront draw the same numbers).
// det_sincos is a fixed double-precision sin/cos (Cody-Waite + Taylor) standing in for
// std::sin/std::cos(float) at render.hpp:160-167: integer and IEEE-double operations only.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    const uint32_t s = x * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ uint32_t root_key(uint32_t seed_hash, uint32_t pixel, uint32_t sample) {
    return pcg_hash(sample + pcg_hash(pixel + seed_hash));                  // seed_hash = pcg_hash(seed)
}
__device__ __forceinline__ uint32_t child_key(uint32_t key, uint32_t child) { return pcg_hash(key ^ (0x632BE5ABu * (child + 1u))); }
__device__ __forceinline__ float urand_key(uint32_t key, uint32_t j) {
    const uint32_t h = pcg_hash(key + j * 0x9E3779B9u + 0x85EBCA6Bu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

static __device__ __noinline__ void det_sincos(float angle, float &s, float &c) {
    const double x = (double)angle;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const double kf = __builtin_floor(x * two_over_pi + 0.5);
    const double r = (x - kf * pio2_hi) - kf * pio2_lo;
    const double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;
    ps = ps * r2 + 1.0 / 6227020800.0;
    ps = ps * r2 - 1.0 / 39916800.0;
    ps = ps * r2 + 1.0 / 362880.0;
    ps = ps * r2 - 1.0 / 5040.0;
    ps = ps * r2 + 1.0 / 120.0;
    ps = ps * r2 - 1.0 / 6.0;
    const double sr = r + r * (r2 * ps);
    double pc = 1.0 / 20922789888000.0;
    pc = pc * r2 - 1.0 / 87178291200.0;
    pc = pc * r2 + 1.0 / 479001600.0;
    pc = pc * r2 - 1.0 / 3628800.0;
    pc = pc * r2 + 1.0 / 40320.0;
    pc = pc * r2 - 1.0 / 720.0;
    pc = pc * r2 + 1.0 / 24.0;
    pc = pc * r2 - 0.5;
    const double cr = 1.0 + r2 * pc;
    const long long k = (long long)kf;
    double sv, cv;
    switch ((int)(k & 3)) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    s = (float)sv;
    c = (float)cv;
}

// sample(texture, hit, uvs) (scene/texture/albedo.hpp:9-11, edge.hpp:12-21, checker.hpp:13-28, bitmap.hpp:46-59).
// hit_w is evaluated in double there (`1. - hit_u - hit_v` with a double literal) and rounded to float.
__device__ __forceinline__ V3 sample_texture(const DevTexture *T, const DevTriUv *uv, const float u, const float v,
                                             const uint8_t *tex_pixels) {
    const V3 a = mk(T->a[0], T->a[1], T->a[2]), b = mk(T->b[0], T->b[1], T->b[2]);
    if (T->kind == RTK_TEX_ALBEDO) return a;
    const float w = (float)((1.0 - (double)u) - (double)v);
    if (T->kind == RTK_TEX_EDGES) return (u < T->param || v < T->param || w < T->param) ? a : b;
    const float fx = (w * uv->uv[0] + u * uv->uv[2]) + v * uv->uv[4];          // hit_w * uvs.x + hit_u * uvs.y + hit_v * uvs.z
    const float fy = (w * uv->uv[1] + u * uv->uv[3]) + v * uv->uv[5];
    if (T->kind == RTK_TEX_BITMAP) {
        // bitmap.hpp:53-59: `size_t row = (1. - final_uv.y) * height` is a double product, `size_t column = final_uv.x * width` a
        // float one (the size_t converts to F); both truncate.  A negative value is undefined there: x86-64 yields 0 above -1 and
        // a huge number (clamped to the last row / column) below, which is what is done here.
        const int bw = T->bmp[0], bh = T->bmp[1];
        const double rd = (1.0 - (double)fy) * (double)bh;
        const float cf = fx * (float)bw;
        const long long ri = (long long)rd, ci = (long long)cf;      // NaN converts to 0 on the device; on x86 to INT64_MIN (last row)
        const int row = (ri < 0 || ri > bh - 1 || rd != rd) ? bh - 1 : (int)ri;
        const int col = (ci < 0 || ci > bw - 1 || cf != cf) ? bw - 1 : (int)ci;
        const uint8_t *px = tex_pixels + (size_t)T->bmp[2] + ((size_t)row * (size_t)bw + (size_t)col) * 3u;
        const float color_scale = (float)(1.0 / 255.0);              // bitmap.hpp:19
        return mk((float)px[0] * color_scale, (float)px[1] * color_scale, (float)px[2] * color_scale);
    }
    const int u2 = (int)(fx / T->param), v2 = (int)(fy / T->param);
    return ((u2 + v2) % 2 == 0) ? a : b;
}

// Camera ray of pixel (px, py) (render.hpp:35-62); `key` is the sample's root key (draws 0 and 1 jitter the sample).
__device__ __forceinline__ Ray camera_ray(const RenderArgs &A, const uint32_t px, const uint32_t py, const uint32_t key) {
    float rx = (float)px, ry = (float)py;
    if (A.spp == 1) { rx += 0.5f; ry += 0.5f; }
    else {
        rx += urand_key(key, 0u);
        ry += urand_key(key, 1u);
    }
    const float ndc_x = rx / A.width_f, ndc_y = ry / A.height_f;  // render.hpp:43-44 ((float)width, (float)height: RenderArgs)
    float sx = (2.0f * ndc_x) - 1.0f;
    float sy = 1.0f - (2.0f * ndc_y);
    sx *= A.aspect;
    sx *= A.tan_half_fov;                                       // render.hpp:55-57: float *= tanf(float) (host, api.hip)
    sy *= A.tan_half_fov;
    const float *M = A.cam_mat;                                 // transpose(camera.matrix) * dir
    V3 d = mk(M[0] * sx + M[3] * sy + M[6] * -1.0f, M[1] * sx + M[4] * sy + M[7] * -1.0f,
              M[2] * sx + M[5] * sy + M[8] * -1.0f);
    d = normalized(d);
    return make_ray(mk(A.cam_pos[0], A.cam_pos[1], A.cam_pos[2]), d);
}

}  // namespace dev
}  // namespace rtk
