// Ray repacking for the batched intersect<cull> (kd_tree_simd.hpp:187-264 called per ray by the reference; here a batch).
//
// The wave-cooperative walk is as fast as its 64 rays are alike: the camera rays of a frame in pixel order run at 31 Grays/s,
// the same rays shuffled at 1.  A batch that arrives in no useful order is therefore sorted first: every ray gets a key -- the
// cell of its origin and direction in a grid over the batch's own bounds, Morton-interleaved over the dimensions that vary at
// all (directions as the two coordinates of their octahedral map -- camera rays: 15 bits each; rays from everywhere to everywhere: five dimensions, 6 bits
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


// Directions enter the keys as TWO numbers: the octahedral map of the direction (L1-normalised; the hemisphere away from the pole folded
// outwards), with the pole at the dominant axis of the batch's first ray so that a frustum of camera rays lies in the unfolded middle.
// Three direction components spent a third of the key bits on a coordinate the other two determine.  pole = axis | negative << 2.
__device__ __forceinline__ uint32_t octa_pole(const rtk_ray *rays, const size_t n) {
    if (n == 0) return 2u;
    const float *p = reinterpret_cast<const float *>(rays);
    const float ax = __builtin_fabsf(p[3]), ay = __builtin_fabsf(p[4]), az = __builtin_fabsf(p[5]);
    const uint32_t axis = (ax > ay && ax > az) ? 0u : (ay > az ? 1u : 2u);
    return axis | ((p[3 + axis] < 0.0f) ? 4u : 0u);
}
__device__ __forceinline__ float2 octa_map(const float dx, const float dy, const float dz, const uint32_t pole) {
    const uint32_t axis = pole & 3u;
    float a = axis == 0u ? dy : (axis == 1u ? dz : dx), b = axis == 0u ? dz : (axis == 1u ? dx : dy), c = axis == 0u ? dx : (axis == 1u ? dy : dz);
    if (pole & 4u) c = -c;
    const float inv = 1.0f / ((__builtin_fabsf(a) + __builtin_fabsf(b)) + __builtin_fabsf(c));
    a *= inv; b *= inv;
    if (c < 0.0f) {
        const float ta = (1.0f - __builtin_fabsf(b)) * (a >= 0.0f ? 1.0f : -1.0f), tb = (1.0f - __builtin_fabsf(a)) * (b >= 0.0f ? 1.0f : -1.0f);
        a = ta; b = tb;
    }
    return make_float2(a, b);
}

}  // namespace

// bounds[0..5] = min of (o.xyz, octahedral u, v, 0), bounds[6..11] = max, as f2key values (initialised to 0xFFFFFFFF / 0 by the host);
// bounds[12] counts the sampled waves whose own directions are far apart (the coherence probe of RTK_TRACE_AUTO), bounds[13] the
// sampled waves, bounds[14] (a float) sums the extents of the waves' origins.  Grid-stride over the sampled waves, minima / maxima kept per lane, ONE set of atomics per workgroup (one per
// wave of 64 rays, the first version, spent 10 ms on 65,536 x 14 atomics to the same fourteen words).
// Every workgroup of k_ray_bounds leaves its fifteen words in its own row of the partials (bounds + kRepackBoundsWords + 16 * block); one
// workgroup folds the rows afterwards.  (Atomics on the fifteen result words from 1,024 workgroups took 50 of the probe's 55 us.)
__device__ void fold_partials(uint32_t *bounds, const uint32_t n_blocks) {
    __shared__ uint32_t f[19];
    __shared__ float f_osum;
    auto is_min = [](uint32_t w) { return w < 6u || w == 15u || w == 16u; };
    auto is_max = [](uint32_t w) { return (w >= 6u && w < 12u) || w == 17u || w == 18u; };
    if (threadIdx.x < 19u) f[threadIdx.x] = is_min(threadIdx.x) ? 0xFFFFFFFFu : 0u;
    if (threadIdx.x == 19u) f_osum = 0.0f;
    __syncthreads();
    uint32_t acc[19];
#pragma unroll
    for (uint32_t w = 0; w < 19u; ++w) acc[w] = is_min(w) ? 0xFFFFFFFFu : 0u;
    float osum = 0.0f;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        const uint32_t *p = bounds + kRepackBoundsWords + (uint32_t)kRepackRowWords * b;
#pragma unroll
        for (uint32_t w = 0; w < 19u; ++w) {
            if (w == 14u) continue;
            const uint32_t x = p[w];
            acc[w] = is_min(w) ? (x < acc[w] ? x : acc[w]) : (is_max(w) ? (x > acc[w] ? x : acc[w]) : acc[w] + x);
        }
        osum += __uint_as_float(p[14]);
    }
    // (one set of LDS atomics per wave: 256 lanes on one address take their turns, 30 us of them)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (uint32_t w = 0; w < 19u; ++w) {
            if (w == 14u) continue;
            const uint32_t x = (uint32_t)__shfl_xor((int)acc[w], off);
            acc[w] = is_min(w) ? (x < acc[w] ? x : acc[w]) : (is_max(w) ? (x > acc[w] ? x : acc[w]) : acc[w] + x);
        }
        osum += __shfl_xor(osum, off);
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (uint32_t w = 0; w < 19u; ++w) {
            if (w == 14u) continue;
            if (is_min(w)) atomicMin(f + w, acc[w]); else if (is_max(w)) atomicMax(f + w, acc[w]); else atomicAdd(f + w, acc[w]);
        }
        atomicAdd(&f_osum, osum);
    }
    __syncthreads();
    if (threadIdx.x < 14u) bounds[threadIdx.x] = f[threadIdx.x];
    if (threadIdx.x == 14u) bounds[14] = __float_as_uint(f_osum);
    if (threadIdx.x >= 15u && threadIdx.x < 19u) bounds[3u + threadIdx.x] = f[threadIdx.x];      // [18..19] u, v minima, [20..21] maxima
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_bounds_fold(uint32_t *bounds, uint32_t n_blocks) { fold_partials(bounds, n_blocks); }

template <bool PROBE>
__global__ __launch_bounds__(256) void k_ray_bounds(const rtk_ray *rays, size_t n, uint32_t *bounds, uint32_t wave_stride) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t n_waves = (n + 63) / 64;
    const size_t first = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, step = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float inf = __builtin_inff();
    float lo[8], hi[8];                                                     // o.xyz, d.xyz, octahedral u, v
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo[k] = inf; hi[k] = -inf; }
    uint32_t wide = 0u, seen = 0u;
    float osum = 0.0f;
    const uint32_t pole = octa_pole(rays, n);
    for (size_t w = first; w * wave_stride < n_waves; w += step) {
        const size_t i = w * wave_stride * 64 + lane;
        const bool have = i < n;
        const float *p = reinterpret_cast<const float *>(rays + (have ? i : 0));
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = p[k];
        const float2 uv = octa_map(v[3], v[4], v[5], pole);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = k < 6 ? v[k] : (k == 6 ? uv.x : uv.y);
            const bool ok = have && (x == x) && __builtin_fabsf(x) < 1.0e30f;
            lo[k] = __builtin_fminf(lo[k], ok ? x : inf);
            hi[k] = __builtin_fmaxf(hi[k], ok ? x : -inf);
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
    // a row of the partials: [0..5] minima, [6..11] maxima, [12] wide, [13] seen, [14] origin extents, [15..16] u, v minima, [17..18] maxima
    __shared__ uint32_t sh[19];
    __shared__ float sh_osum;
    if (threadIdx.x < 19u) sh[threadIdx.x] = (threadIdx.x < 6u || threadIdx.x == 15u || threadIdx.x == 16u) ? 0xFFFFFFFFu : 0u;
    if (threadIdx.x == 19u) sh_osum = 0.0f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float l = wave_min_f(lo[k]), h = wave_max_f(hi[k]);
        if (lane == 0u && l <= h) { atomicMin(sh + (k < 6 ? k : 9 + k), f2key(l)); atomicMax(sh + (k < 6 ? 6 + k : 11 + k), f2key(h)); }
    }
    if (PROBE && lane == 0u) { atomicAdd(sh + 12, wide); atomicAdd(sh + 13, seen); atomicAdd(&sh_osum, osum); }
    __syncthreads();
    uint32_t *row = bounds + kRepackBoundsWords + (uint32_t)kRepackRowWords * blockIdx.x;      // folded by fold_partials
    if (threadIdx.x < 19u && threadIdx.x != 14u) row[threadIdx.x] = sh[threadIdx.x];
    else if (threadIdx.x == 14u) row[14] = __float_as_uint(sh_osum);        // sum of the waves' origin extents
}

// Is the batch a row-major RASTER of rays (the camera rays of a frame, row after row)?  Then a wave of 64 consecutive rays is a
// 64x1 strip of pixels, and the triangles such a strip can touch are twice those of an 8x8 block: dealt to the waves as 8x8
// blocks (k_intersect, IntersectArgs::raster_w) the same batch runs in half the time (2^24 camera rays on hw09/scene5: 0.545 ->
// 0.303 ms).  One workgroup looks at the first rays: the step between neighbours is small and regular, and every W-th step is a
// jump (the end of a row).  out[0] = W (a multiple of 8, >= 64, at least 16 rows seen), else 0.  Only lane placement depends on
// it -- a wrong guess costs speed, never a result.
// ... and the probe's verdict, on the device (the host reads the same words): bounds[16] = 1 when the batch comes in no useful
// order and is to be sorted (the criteria of RTK_TRACE_AUTO: a quarter of the probed waves with directions more than 0.25 apart, or
// origins spread over a quarter of the batch's extent), bounds[17] = how many of the key's coordinates vary at all (k_ray_keys).
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
    if (oext == 0.0f) {                                                     // one origin: k_ray_keys takes the directions' octahedral coordinates
        dims = 0u;
        for (int k = 0; k < 2; ++k) {
            const float e = k2f(bounds[20 + k]) - k2f(bounds[18 + k]);
            if (e > 1.0e-6f && e < 1.0e30f) dims += 1u;
        }
    }
    const uint32_t waves = bounds[13];
    const float wide = waves ? (float)bounds[12] / (float)waves : 0.0f;
    const float spread = (waves && oext > 0.0f) ? (__uint_as_float(bounds[14]) / (float)waves) / oext : 0.0f;
    bounds[16] = (wide >= 0.25f || spread >= 0.25f) ? 1u : 0u;
    bounds[17] = dims;
}

__global__ __launch_bounds__(256) void k_raster_probe(const rtk_ray *rays, size_t n, uint32_t *bounds, uint32_t want_raster, uint32_t fold_blocks) {
    uint32_t *out = bounds + 15;
    if (fold_blocks != 0u) fold_partials(bounds, fold_blocks);
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

// x's bits b -> b * n (n - 1 zero bits between neighbours): the Morton interleave of n coordinates is the OR of their spreads, shifted
__device__ __forceinline__ uint32_t spread_bits(uint32_t x, const uint32_t n, const uint32_t bits) {
    if (n == 1u) return x;
    if (n == 2u) {
        x &= 0x7FFFu;
        x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
        return x;
    }
    if (n == 3u) {
        x &= 0x3FFu;
        x = (x ^ (x << 16)) & 0xFF0000FFu; x = (x ^ (x << 8)) & 0x0300F00Fu; x = (x ^ (x << 4)) & 0x030C30C3u; x = (x ^ (x << 2)) & 0x09249249u;
        return x;
    }
    uint32_t out = 0u;
    for (uint32_t b = 0; b < bits; ++b) out |= ((x >> b) & 1u) << (b * n);
    return out;
}

// key[i] = Morton code of ray i's cell, idx[i] = i.
// Workgroups stay and take tile after tile of 256 rays (384 float4 in a row, read as such, handed to their lanes through LDS, the next
// tile's loads in flight while this one's keys are made): 65,536 workgroups of one tile each spent their 3 us lives mostly waiting for
// the bounds, their rays and a free slot -- 180 us for 0.54 GB.
__global__ __launch_bounds__(256) void k_ray_keys(const rtk_ray *rays, size_t n, const uint32_t *bounds, uint32_t *keys, uint32_t *idx, uint32_t dirs3, uint32_t only_if_sort) {
    if (only_if_sort != 0u && bounds[16] == 0u) return;      // launched ahead of the probe's verdict (api.hip): the batch is walked as it comes
    __shared__ float4 tile[384];
    const bool aligned = (reinterpret_cast<uintptr_t>(rays) & 15u) == 0u;
    const size_t n_tiles = (n + 255u) / 256u;
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
    // One origin for all rays (camera rays, shadow rays of a point): the directions as the TWO coordinates of their octahedral map,
    // 15 bits each -- three components spend a third of the key on a number the other two determine.  (Where origins vary as well the
    // three components sort better: 2^24 uniform rays 10.0 ms against 11.0 -- their top bits are the octant, which is what a walk shares.)
    const bool octa = (active & 7u) == 0u && dirs3 == 0u;
    if (octa) {
        active = 0u; n_active = 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[3 + k] = key2f(bounds[18 + k]);
            ext[3 + k] = key2f(bounds[20 + k]) - lo[3 + k];
            if (ext[3 + k] > 1.0e-6f && ext[3 + k] < 1.0e30f) { active |= 8u << k; n_active += 1u; }
        }
    }
    const uint32_t bits = n_active != 0u ? 30u / n_active : 0u;             // per dimension: 2 active -> 15, 3 -> 10, 6 -> 5
    const float cells = (float)(1u << bits);
    float to_cell[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) to_cell[k] = cells / ext[k];               // (only the order of the rays hangs on these, no result)
    const uint32_t pole = octa_pole(rays, n);
    float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f), pre1 = pre0;
    auto prefetch = [&](const size_t t) {
        if (t < n_tiles && aligned && (t + 1u) * 256u <= n) {
            const float4 *src = reinterpret_cast<const float4 *>(rays + t * 256u);
            pre0 = src[threadIdx.x];
            if (threadIdx.x < 128u) pre1 = src[256u + threadIdx.x];
        }
    };
    prefetch(blockIdx.x);
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const bool staged = aligned && (t + 1u) * 256u <= n;                 // (workgroup-uniform)
        if (staged) {
            tile[threadIdx.x] = pre0;
            if (threadIdx.x < 128u) tile[256u + threadIdx.x] = pre1;
        }
        __syncthreads();
        prefetch(t + gridDim.x);
        const size_t i = t * 256u + threadIdx.x;
        uint32_t key = 0u;
        if (i < n && n_active != 0u) {
            const float *p = staged ? reinterpret_cast<const float *>(tile) + 6u * threadIdx.x : reinterpret_cast<const float *>(rays + i);
            float v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = p[k];
            if (octa) { const float2 uv = octa_map(v[3], v[4], v[5], pole); v[3] = uv.x; v[4] = uv.y; }
            // interleave, most significant bit first, origin dimensions ahead of direction dimensions within a bit plane
            uint32_t place = n_active;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (!((active >> k) & 1u)) continue;                        // (uniform)
                float x = (v[k] - lo[k]) * to_cell[k];
                x = (x == x) ? x : 0.0f;                                    // a NaN ray goes to cell 0 (it misses everything anyway)
                x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), cells - 1.0f);   // (so does one outside the sampled bounds: a border cell)
                place -= 1u;
                key |= spread_bits((uint32_t)x, n_active, bits) << place;
            }
        }
        __syncthreads();                                                     // the tile is read: the next one may land
        if (i < n) { keys[i] = key; idx[i] = (uint32_t)i; }
    }
}

}  // namespace dev

hipError_t launch_ray_bounds(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, uint32_t wave_stride, bool probe, hipStream_t s,
                             unsigned *unfolded_blocks) {
    if (wave_stride == 0u) wave_stride = 1u;
    const size_t waves = (n + 63) / 64;
    const size_t sampled = (waves + wave_stride - 1) / wave_stride;
    size_t blocks = (sampled + 3) / 4;
    if (blocks > (size_t)kRepackMaxBlocks) blocks = kRepackMaxBlocks;       // grid-stride: 8 workgroups per CU
    if (blocks != 0) {
        if (probe) hipLaunchKernelGGL(dev::k_ray_bounds<true>, dim3((unsigned)blocks), dim3(256), 0, s, d_rays, n, d_bounds, wave_stride);
        else hipLaunchKernelGGL(dev::k_ray_bounds<false>, dim3((unsigned)blocks), dim3(256), 0, s, d_rays, n, d_bounds, wave_stride);
    }
    if (unfolded_blocks != nullptr) *unfolded_blocks = (unsigned)blocks;    // launch_raster_probe folds
    else hipLaunchKernelGGL(dev::k_bounds_fold, dim3(1), dim3(256), 0, s, d_bounds, (uint32_t)blocks);
    return hipGetLastError();
}

hipError_t launch_raster_probe(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, bool want_raster, hipStream_t s, unsigned fold_blocks) {
    hipLaunchKernelGGL(dev::k_raster_probe, dim3(1), dim3(256), 0, s, d_rays, n, d_bounds, want_raster ? 1u : 0u, (uint32_t)fold_blocks);
    return hipGetLastError();
}

hipError_t repack_temp_bytes(size_t n, size_t *bytes) {
    uint32_t *nk = nullptr;
    return rocprim::radix_sort_pairs(nullptr, *bytes, nk, nk, nk, nk, n, 0u, 30u, nullptr);
}

hipError_t launch_ray_keys(const rtk_ray *d_rays, size_t n, const uint32_t *d_bounds, uint32_t *d_keys, uint32_t *d_idx, hipStream_t s,
                           bool dirs3, bool only_if_sort) {
    if (n == 0) return hipSuccess;
    const size_t tiles = (n + 255) / 256;
    const unsigned blocks = (unsigned)(tiles < 2048 ? tiles : 2048);           // 8 workgroups per CU, grid-stride over the tiles
    hipLaunchKernelGGL(dev::k_ray_keys, dim3(blocks), dim3(256), 0, s, d_rays, n, d_bounds, d_keys, d_idx, dirs3 ? 1u : 0u, only_if_sort ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_key_sort(size_t n, uint32_t *d_keys /* [2n] */, uint32_t *d_idx /* [2n] */, void *d_temp, size_t temp_bytes, hipStream_t s,
                           unsigned begin_bit) {
    if (n == 0) return hipSuccess;
    // (begin_bit > 0: the key's lowest bits are left unsorted -- a radix pass less; rays that differ only there are neighbours anyway)
    return rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys + n, d_idx, d_idx + n, n, begin_bit < 30u ? begin_bit : 0u, 30u, s);   // sorted indices: d_idx + n
}

}  // namespace rtk
