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
    static const uint32_t init[kRepackBoundsWords] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    hipError_t e = hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
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
                           void *d_temp, size_t temp_bytes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dev::k_ray_keys, dim3(blocks), dim3(256), 0, s, d_rays, n, d_bounds, d_keys, d_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys + n, d_idx, d_idx + n, n, 0u, 30u, s);   // sorted indices: d_idx + n
}

}  // namespace rtk
