// Ray repacking for the batched intersect<cull> (kd_tree_simd.hpp:187-264 called per ray by the reference; here a batch).
//
// The wave-cooperative walk is as fast as its 64 rays are alike: the camera rays of a frame in pixel order run at 31 Grays/s,
// the same rays shuffled at 1.  A batch that arrives in no useful order is therefore sorted first: every ray gets a key -- the
// cell of its origin and direction in a grid over the batch's own bounds, Morton-interleaved over the dimensions that vary at
// all (camera rays: the three direction components, 10 bits each; rays from everywhere to everywhere: six dimensions, 5 bits
// each) -- the (key, index) pairs are radix-sorted (rocPRIM), and k_intersect walks the rays in that order, lane i taking ray
// perm[i] and writing hit perm[i].  A ray's result does not depend on its neighbours in the wave (trace.hip.hpp: every
// strategy shows each lane exactly its own leaf / triangle sequence), so the hits are the same bits in the same places.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "repack.hpp"

namespace rtk {
namespace dev {

namespace {

// monotone float <-> uint map (NaNs sort last and are never the minimum or maximum that matters: they are skipped)
__device__ __forceinline__ uint32_t f2key(const float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(const uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off));
    return v;
}

}  // namespace

// bounds[0..5] = min of (o.xyz, d.xyz), bounds[6..11] = max, as f2key values (initialised to 0xFFFFFFFF / 0 by the host);
// bounds[12] counts the sampled waves whose own directions are far apart (the coherence probe of RTK_TRACE_AUTO), bounds[13] the
// sampled waves, bounds[14] (a float) sums the extents of the waves' origins.  Grid-stride over the sampled waves, minima / maxima kept per lane, ONE set of atomics per workgroup (one per
// wave of 64 rays, the first version, spent 10 ms on 65,536 x 14 atomics to the same fourteen words).
__global__ void k_bounds_init(uint32_t *bounds) {
    if (threadIdx.x < (unsigned)kRepackBoundsWords) bounds[threadIdx.x] = threadIdx.x < 6u ? 0xFFFFFFFFu : 0u;
}

template <bool PROBE>
__global__ __launch_bounds__(256) void k_ray_bounds(const rtk_ray *rays, size_t n, uint32_t *bounds, uint32_t wave_stride) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_waves = (n + 63) / 64;
    const size_t first = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, step = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float inf = __builtin_inff();
    float lo[6], hi[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { lo[k] = inf; hi[k] = -inf; }
    uint32_t wide = 0u, seen = 0u;
    float osum = 0.0f;
    for (size_t w = first; w * wave_stride < n_waves; w += step) {
        const size_t i = w * wave_stride * 64 + lane;
        const bool have = i < n;
        const float *p = reinterpret_cast<const float *>(rays + (have ? i : 0));
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            v[k] = p[k];
            const bool ok = have && (v[k] == v[k]) && __builtin_fabsf(v[k]) < 1.0e30f;
            lo[k] = __builtin_fminf(lo[k], ok ? v[k] : inf);
            hi[k] = __builtin_fmaxf(hi[k], ok ? v[k] : -inf);
        }
        if (PROBE) {                                                        // this wave's own directions: more than ~15 degrees wide?
            float dw = 0.0f;
#pragma unroll
            for (int k = 3; k < 6; ++k) {
                const bool ok = have && (v[k] == v[k]);
                dw = __builtin_fmaxf(dw, wave_max_f(ok ? v[k] : -inf) - wave_min_f(ok ? v[k] : inf));
            }
            wide += dw > 0.25f ? 1u : 0u;
            seen += 1u;
            float ow = 0.0f;                                                // ... and how far apart its origins lie
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool ok = have && (v[k] == v[k]) && __builtin_fabsf(v[k]) < 1.0e30f;
                ow = __builtin_fmaxf(ow, wave_max_f(ok ? v[k] : -inf) - wave_min_f(ok ? v[k] : inf));
            }
            osum += ow > 0.0f ? ow : 0.0f;
        }
    }
    __shared__ uint32_t sh[14];
    __shared__ float sh_osum;
    if (threadIdx.x < 14u) sh[threadIdx.x] = threadIdx.x < 6u ? 0xFFFFFFFFu : 0u;
    if (threadIdx.x == 14u) sh_osum = 0.0f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float l = wave_min_f(lo[k]), h = wave_max_f(hi[k]);
        if (lane == 0u && l <= h) { atomicMin(sh + k, f2key(l)); atomicMax(sh + 6 + k, f2key(h)); }
    }
    if (PROBE && lane == 0u) { atomicAdd(sh + 12, wide); atomicAdd(sh + 13, seen); atomicAdd(&sh_osum, osum); }
    __syncthreads();
    if (threadIdx.x < 6u) atomicMin(bounds + threadIdx.x, sh[threadIdx.x]);
    else if (threadIdx.x < 12u) atomicMax(bounds + threadIdx.x, sh[threadIdx.x]);
    else if (PROBE && threadIdx.x < 14u) atomicAdd(bounds + threadIdx.x, sh[threadIdx.x]);
    else if (PROBE && threadIdx.x == 14u) atomicAdd(reinterpret_cast<float *>(bounds + 14), sh_osum);   // sum of the waves' origin extents
}

// Is the batch a row-major RASTER of rays (the camera rays of a frame, row after row)?  Then a wave of 64 consecutive rays is a
// 64x1 strip of pixels, and the triangles such a strip can touch are twice those of an 8x8 block: dealt to the waves as 8x8
// blocks (k_intersect, IntersectArgs::raster_w) the same batch runs in half the time (2^24 camera rays on hw09/scene5: 0.545 ->
// 0.303 ms).  One workgroup looks at the first rays: the step between neighbours is small and regular, and every W-th step is a
// jump (the end of a row).  out[0] = W (a multiple of 8, >= 64, at least 16 rows seen), else 0.  Only lane placement depends on
// it -- a wrong guess costs speed, never a result.
// ... and the probe's verdict, on the device (the host reads the same words): bounds[16] = 1 when the batch comes in no useful
// order and is to be sorted (the criteria of RTK_TRACE_AUTO: a quarter of the probed waves with directions more than 0.25 apart, or
// origins spread over a quarter of the batch's extent), bounds[17] = how many of the six ray coordinates vary at all.
__device__ void probe_verdict(uint32_t *bounds) {
    auto k2f = [](uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); };
    float scale = 0.0f, ext[6], oext = 0.0f;
    for (int k = 0; k < 6; ++k) {
        const float lo = k2f(bounds[k]), hi = k2f(bounds[6 + k]);
        ext[k] = hi - lo;
        scale = __builtin_fmaxf(scale, __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi)));
    }
    uint32_t dims = 0u;
    for (int k = 0; k < 6; ++k)
        if (ext[k] > 1.0e-6f * scale && ext[k] < 1.0e30f) { dims += 1u; if (k < 3 && ext[k] > oext) oext = ext[k]; }
    const uint32_t waves = bounds[13];
    const float wide = waves ? (float)bounds[12] / (float)waves : 0.0f;
    const float spread = (waves && oext > 0.0f) ? (__uint_as_float(bounds[14]) / (float)waves) / oext : 0.0f;
    bounds[16] = (wide >= 0.25f || spread >= 0.25f) ? 1u : 0u;
    bounds[17] = dims;
}

__global__ __launch_bounds__(256) void k_raster_probe(const rtk_ray *rays, size_t n, uint32_t *bounds, uint32_t want_raster) {
    uint32_t *out = bounds + 15;
    if (threadIdx.x == 0u) probe_verdict(bounds);
    if (want_raster == 0u) { if (threadIdx.x == 0u) out[0] = 0u; return; }
    __shared__ float s_step0;
    __shared__ uint32_t s_first, s_bad;
    constexpr uint32_t kLook = 16384;
    const uint32_t m = n < kLook ? (uint32_t)n : kLook;
    auto dir_step = [&](uint32_t i, uint32_t j) {
        const float *a = reinterpret_cast<const float *>(rays + i), *b = reinterpret_cast<const float *>(rays + j);
        const float s = (__builtin_fabsf(a[3] - b[3]) + __builtin_fabsf(a[4] - b[4])) + __builtin_fabsf(a[5] - b[5]);
        return (s == s) ? s : 1.0e30f;
    };
    if (threadIdx.x == 0u) { s_step0 = m >= 2u ? dir_step(0u, 1u) : 0.0f; s_first = 0xFFFFFFFFu; s_bad = 0u; }
    __syncthreads();
    const float s0 = s_step0;
    if (!(s0 > 0.0f && s0 < 0.05f)) { if (threadIdx.x == 0u) out[0] = 0u; return; }      // neighbours are not neighbours
    for (uint32_t c = 64u + 8u * threadIdx.x; c < m; c += 8u * blockDim.x)               // the first jump: the end of row 0 (a multiple of 8)
        if (dir_step(c - 1u, c) > 8.0f * s0) atomicMin(&s_first, c);
    __syncthreads();
    const uint32_t W = s_first;
    if (W == 0xFFFFFFFFu || W < 64u || (W & 7u) != 0u || (size_t)W * 16u > n) { if (threadIdx.x == 0u) out[0] = 0u; return; }
    // 16 rows: steps inside a row stay small, rows end where they should, and a ray lies next to the one W before it
    for (uint32_t k = threadIdx.x; k < 16u * 32u; k += blockDim.x) {
        const uint32_t row = k >> 5, x = (uint32_t)(((size_t)(k & 31u) * (W - 2u)) / 31u);
        const uint32_t i = row * W + x;
        bool ok = dir_step(i, i + 1u) < 4.0f * s0;
        if (row > 0u) ok = ok && dir_step(i, i - W) < 8.0f * s0;
        ok = ok && ((k & 31u) != 31u || (size_t)(row + 1u) * W >= n || dir_step(row * W + W - 1u, row * W + W) > 8.0f * s0);
        if (!ok) atomicAdd(&s_bad, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0u) out[0] = s_bad == 0u ? W : 0u;
}

// key[i] = Morton code of ray i's cell, idx[i] = i
__global__ __launch_bounds__(256) void k_ray_keys(const rtk_ray *rays, size_t n, const uint32_t *bounds, uint32_t *keys, uint32_t *idx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo[6], ext[6];
    uint32_t active = 0u, n_active = 0u;
    float scale = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        lo[k] = key2f(bounds[k]);
        const float hi = key2f(bounds[6 + k]);
        ext[k] = hi - lo[k];
        scale = __builtin_fmaxf(scale, __builtin_fmaxf(__builtin_fabsf(lo[k]), __builtin_fabsf(hi)));
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (ext[k] > 1.0e-6f * scale && ext[k] < 1.0e30f) { active |= 1u << k; n_active += 1u; }
    uint32_t key = 0u;
    if (n_active != 0u) {
        const uint32_t bits = 30u / n_active;                               // per dimension: 3 active -> 10, 6 -> 5
        const float cells = (float)(1u << bits);
        const float *p = reinterpret_cast<const float *>(rays + i);
        uint32_t q[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float t = (p[k] - lo[k]) / ext[k] * cells;
            t = (t == t) ? t : 0.0f;                                        // a NaN ray goes to cell 0 (it misses everything anyway)
            t = __builtin_fminf(__builtin_fmaxf(t, 0.0f), cells - 1.0f);
            q[k] = (uint32_t)t;
        }
        // interleave, most significant bit first, origin dimensions ahead of direction dimensions within a bit plane
        for (uint32_t b = bits; b-- > 0u;) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if ((active >> k) & 1u) key = (key << 1) | ((q[k] >> b) & 1u);
        }
    }
    keys[i] = key;
    idx[i] = (uint32_t)i;
}

}  // namespace dev

hipError_t launch_ray_bounds(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, uint32_t wave_stride, hipStream_t s) {
    // (set on the device: a copy from pageable host memory is staged synchronously, ~20 us in front of every probe)
    hipLaunchKernelGGL(dev::k_bounds_init, dim3(1), dim3(64), 0, s, d_bounds);
    if (wave_stride == 0u) wave_stride = 1u;
    const size_t waves = (n + 63) / 64;
    const size_t sampled = (waves + wave_stride - 1) / wave_stride;
    size_t blocks = (sampled + 3) / 4;
    if (blocks == 0) return hipSuccess;
    if (blocks > 2048) blocks = 2048;                                       // grid-stride: 8 workgroups per CU
    if (wave_stride > 1u) hipLaunchKernelGGL(dev::k_ray_bounds<true>, dim3((unsigned)blocks), dim3(256), 0, s, d_rays, n, d_bounds, wave_stride);
    else hipLaunchKernelGGL(dev::k_ray_bounds<false>, dim3((unsigned)blocks), dim3(256), 0, s, d_rays, n, d_bounds, wave_stride);
    return hipGetLastError();
}

hipError_t launch_raster_probe(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, bool want_raster, hipStream_t s) {
    hipLaunchKernelGGL(dev::k_raster_probe, dim3(1), dim3(256), 0, s, d_rays, n, d_bounds, want_raster ? 1u : 0u);
    return hipGetLastError();
}

RepackProbe decode_probe(const uint32_t *h) {
    auto k2f = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
    RepackProbe r;
    float scale = 0.0f, ext[6], oext = 0.0f;
    for (int k = 0; k < 6; ++k) {
        const float lo = k2f(h[k]), hi = k2f(h[6 + k]);
        ext[k] = hi - lo;
        const float m = std::fabs(lo) > std::fabs(hi) ? std::fabs(lo) : std::fabs(hi);
        if (m > scale) scale = m;
    }
    r.active_dims = 0;
    for (int k = 0; k < 6; ++k)
        if (ext[k] > 1.0e-6f * scale && ext[k] < 1.0e30f) { r.active_dims += 1; if (k < 3 && ext[k] > oext) oext = ext[k]; }
    r.waves = h[13];
    r.wide_dir_fraction = h[13] ? float(h[12]) / float(h[13]) : 0.0f;
    float osum; std::memcpy(&osum, &h[14], 4);
    r.origin_spread = (h[13] && oext > 0.0f) ? (osum / float(h[13])) / oext : 0.0f;
    return r;
}

hipError_t repack_temp_bytes(size_t n, size_t *bytes) {
    uint32_t *nk = nullptr;
    return rocprim::radix_sort_pairs(nullptr, *bytes, nk, nk, nk, nk, n, 0u, 30u, nullptr);
}

hipError_t launch_ray_sort(const rtk_ray *d_rays, size_t n, const uint32_t *d_bounds, uint32_t *d_keys /* [2n] */, uint32_t *d_idx /* [2n] */,
                           void *d_temp, size_t temp_bytes, hipStream_t s, unsigned begin_bit) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dev::k_ray_keys, dim3(blocks), dim3(256), 0, s, d_rays, n, d_bounds, d_keys, d_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // (begin_bit > 0: the key's lowest bits are left unsorted -- a radix pass less; rays that differ only there are neighbours anyway)
    return rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys + n, d_idx, d_idx + n, n, begin_bit < 30u ? begin_bit : 0u, 30u, s);   // sorted indices: d_idx + n
}

}  // namespace rtk
