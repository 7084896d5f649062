// Streaming ("wavefront") frame pipeline for fork-free scenes (no refractive material, no diffuse GI rays).
//
// In such scenes color_hit (render/render.hpp:133-308) never forks: a pixel's colour is the colour of the last
// surface of its reflection chain (reflective = tail call, :239-250), i.e. a background / constant colour or ONE
// diffuse light sum (:184-208).  The recursion therefore unrolls into per-depth batches:
//
//   k_path(d)    closest hit for every path ray of depth d (camera rays at d = 0) + material switch; reflective hits
//                append a path ray for depth d+1, diffuse hits append a shading point, everything else writes its
//                pixel.  Appends are compacted with wave ballot / mbcnt prefix and one atomic per wave and queue.
//   k_shadow(d)  one closest-hit shadow query per (shading point, light), is_occluded semantics (:110-131)
//   k_resolve(d) sums the unoccluded lights IN LIGHT ORDER (float addition is not associative) and writes the pixel
//
// Every kernel carries one ray and one candidate per lane and nothing else, so eight waves fit on a SIMD and
// scalar-load latency is hidden by occupancy instead of by helper waves; the work of a heavy pixel block is spread
// over several short waves instead of one long one.  Results are bit-identical to the megakernel (same device
// functions, same operation order).
#include <hip/hip_runtime.h>

#include "common.hip.hpp"
#include "stream.hpp"

namespace rtk {
namespace dev {

namespace {

__device__ __forceinline__ uint32_t lane_prefix(const unsigned long long mask) {      // set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// wave-compacted append: returns this lane's slot (only meaningful where `want`)
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, const bool want) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return 0u;
    uint32_t base = 0u;
    if (__lane_id() == 0u) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return base + lane_prefix(mask);
}

// Dynamic work distribution for the queue-driven (persistent) stages: a unit's first item is its own index, every
// further item is drawn from an atomic ticket, so units that got cheap items simply come back for more and the
// stage ends when the queue is empty instead of when the unluckiest static share is done.
__device__ __forceinline__ uint32_t next_item(uint32_t *ticket, const uint32_t n_units) {
    uint32_t t = 0u;
    if (__lane_id() == 0u) t = atomicAdd(ticket, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + n_units;
}

// final_color += colour ; after the last sample: pixels[y][x] = final_color / spp   (render.hpp:66-74)
__device__ __forceinline__ void emit_pixel(const StreamArgs &S, const uint32_t pix, const V3 ret) {
    V3 sum;
    if (S.sample == 0) sum = mk(0.0f + ret.x, 0.0f + ret.y, 0.0f + ret.z);
    else {
        const float *sb = S.ws.sumbuf + (size_t)pix * 3;
        sum = mk(sb[0] + ret.x, sb[1] + ret.y, sb[2] + ret.z);
    }
    if (S.sample == S.r.spp - 1) {
        const float n = (float)S.r.spp;
        float *o = S.r.out + (size_t)pix * 3;
        o[0] = sum.x / n; o[1] = sum.y / n; o[2] = sum.z / n;
    } else {
        float *sb = S.ws.sumbuf + (size_t)pix * 3;
        sb[0] = sum.x; sb[1] = sum.y; sb[2] = sum.z;
    }
}

__device__ __forceinline__ void add_rays(const StreamArgs &S, const Stats &st, const uint32_t rays, const bool stats,
                                         const uint32_t shard) {
    const uint32_t total = wave_sum(rays);
    unsigned long long *c = S.r.counters;
    if (stats) {
        const uint32_t h = wave_sum(st.hits), nd = wave_sum(st.nodes), bp = wave_sum(st.boxpass), lv = wave_sum(st.leaves),
                       tr = wave_sum(st.tris), pk = wave_sum(st.packets16);
        if (__lane_id() == 0u) {
            atomicAdd(c + 2, (unsigned long long)h); atomicAdd(c + 3, (unsigned long long)nd);
            atomicAdd(c + 4, (unsigned long long)bp); atomicAdd(c + 5, (unsigned long long)lv);
            atomicAdd(c + 6, (unsigned long long)tr); atomicAdd(c + 7, (unsigned long long)pk);
        }
    }
    if (__lane_id() == 0u && total != 0u) atomicAdd(c + 8 + (shard % (uint32_t)kRayCounterShards), (unsigned long long)total);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// k_path: LEVEL0 = camera rays, one wave per 8x8 pixel block (same bucket / rank mapping as k_render);
// otherwise depth-`level` reflection rays from the queue, waves striding over groups of 64 rays.
template <bool LEVEL0, bool STATS, int SLICES>
__global__ __launch_bounds__(256, 8) void k_path(StreamArgs S) {
    const RenderArgs &A = S.r;
    // SLICES > 1: the workgroup's wave 0 owns the rays, waves 1.. help with large leaves (trace.hip.hpp)
    __shared__ GroupShared group_sh[1];
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (SLICES > 1 && wave_in_block != 0u) {
        group_helper_loop<SLICES>(A.tree, &group_sh[0], wave_in_block);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = SLICES > 1 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;    // work-unit id
    const uint32_t n_waves = SLICES > 1 ? gridDim.x : gridDim.x * (blockDim.x >> 6);
    const V3 background = mk(A.background[0], A.background[1], A.background[2]);
    const uint32_t level = S.level;
    const uint32_t n_rays = LEVEL0 ? 0u : S.ws.ctrl[kCtrlPathCount + level];
    const uint32_t n_items = LEVEL0 ? 1u : (n_rays + 63u) >> 6;
    const PathRay *qin = S.ws.path[level & 1u];
    PathRay *qout = S.ws.path[(level + 1u) & 1u];
    Stats st = {0, 0, 0, 0, 0, 0};
    SliceCtx sx = {SLICES > 1 ? &group_sh[0] : nullptr, A.slice_min_tris, 0u, true, 0u};
    uint32_t nrays = 0;

#ifdef RTK_DEBUG_WAVE_TIME
    const unsigned long long dbg_b0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t item = LEVEL0 ? 0u : gwave; item < n_items;
         item = LEVEL0 ? n_items : next_item(S.ws.ctrl + kCtrlTicket + level, n_waves)) {
#ifdef RTK_DEBUG_WAVE_TIME
        const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        bool valid;
        uint32_t pix = 0;
        Ray ray;
        if (LEVEL0) {
            const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
            const uint32_t local_bucket = gwave / bpb, sub = gwave % bpb;
            const uint32_t bucket = (uint32_t)A.rank + local_bucket * (uint32_t)A.world;
            const uint32_t bx = (bucket % A.tiles_x) * A.bucket, by = (bucket / A.tiles_x) * A.bucket;
            const uint32_t lx = (sub % A.blocks_per_bucket_side) * 8u + (lane & 7u);
            const uint32_t ly = (sub / A.blocks_per_bucket_side) * 8u + (lane >> 3);
            const uint32_t px = bx + lx, py = by + ly;
            valid = (bucket < A.n_buckets) & (lx < A.bucket) & (ly < A.bucket) & (px < A.width) & (py < A.height);
            pix = (uint32_t)A.out_index(local_bucket, lx, ly, px, py);
            ray = camera_ray(A, px, py, root_key(pcg_hash(A.seed), py * A.width + px, (uint32_t)S.sample));
        } else {
            const uint32_t i = item * 64u + lane;
            valid = i < n_rays;
            const float4 *q = reinterpret_cast<const float4 *>(qin + (valid ? i : 0u));
            const float4 a = q[0], b = q[1];
            pix = __float_as_uint(a.w);
            ray = make_ray(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z));
        }
        Cand c;
        c.t = kFltMax; c.u = c.v = 0.f; c.k = kMiss;
        c = trace<RTK_TRACE_WAVE, STATS, false, SLICES>(A.tree, nullptr, ray, LEVEL0, valid, st, sx);
        nrays += valid ? 1u : 0u;

        // ---- material switch (color_hit, render.hpp:133-308, fork-free subset)
        bool write = false, push_path = false, push_hit = false;
        V3 ret = background, P = background, n_or_d = background, o_next = background;
        uint32_t mat = 0;
        if (valid) {
            if (c.k == kMiss) { write = true; ret = background; }                       // camera miss / reflective miss: background
            else if ((int)level == A.max_depth) { write = true; ret = background; }     // render.hpp:138-139
            else {
                const Surface s = reconstruct(A.tree, c);
                P = ray.o + (c.t * ray.d);
                mat = s.material;
                const DevMaterial *m = A.materials + mat;
                const int kind = m->kind;
                if (kind == RTK_MAT_CONSTANT) { write = true; ret = mk(m->albedo[0], m->albedo[1], m->albedo[2]); }
                else if (kind == RTK_MAT_REFLECTIVE) {                                   // render.hpp:239-250
                    const V3 rd = ray.d - ((2.0f * dot(ray.d, s.hit_normal)) * s.hit_normal);
                    o_next = P + (A.reflection_bias * rd);
                    n_or_d = rd;
                    push_path = true;
                } else {                                                                 // diffuse: light loop comes later
                    n_or_d = m->smooth ? s.hit_normal : s.face_normal;
                    push_hit = true;
                }
            }
        }
        if (write) emit_pixel(S, pix, ret);
        {
            const uint32_t slot = wave_append(S.ws.ctrl + kCtrlPathCount + level + 1u, push_path);
            if (push_path) {
                float4 *q = reinterpret_cast<float4 *>(qout + slot);
                q[0] = make_float4(o_next.x, o_next.y, o_next.z, __uint_as_float(pix));
                q[1] = make_float4(n_or_d.x, n_or_d.y, n_or_d.z, 0.f);
            }
        }
        {
            const uint32_t slot = wave_append(S.ws.ctrl + kCtrlHitCount + level, push_hit);
            if (push_hit) {
                float4 *q = reinterpret_cast<float4 *>(S.ws.hits + slot);
                q[0] = make_float4(P.x, P.y, P.z, __uint_as_float(pix));
                q[1] = make_float4(n_or_d.x, n_or_d.y, n_or_d.z, __uint_as_float(mat));
            }
        }
#ifdef RTK_DEBUG_WAVE_TIME
        if (lane == 0u && level < 4u) {
            const uint32_t dt = (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_t0);
            uint32_t *d = S.ws.ctrl + kCtrlDebug + ((LEVEL0 ? 0u : 1u) * 4u + level) * 4u;
            atomicAdd(d + 0, dt); atomicMax(d + 1, dt); atomicAdd(d + 2, 1u);
        }
#endif
    }
#ifdef RTK_DEBUG_WAVE_TIME
    if (lane == 0u && level < 4u)
        atomicMax(S.ws.ctrl + kCtrlDebug + ((LEVEL0 ? 0u : 1u) * 4u + level) * 4u + 3, (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_b0));
#endif
    if (SLICES > 1) group_post_exit(&group_sh[0]);
    add_rays(S, st, nrays, STATS, gwave);
}

// ------------------------------------------------------------------------------------------------
// k_shadow: item = (group of 64 shading points, light).  Light loop body of render.hpp:184-206 up to the
// occlusion decision; the contribution is stored and summed in light order by k_resolve.
template <bool STATS, int SLICES>
__global__ __launch_bounds__(256, 8) void k_shadow(StreamArgs S) {
    const RenderArgs &A = S.r;
    // SLICES > 1: the workgroup's wave 0 owns the rays, waves 1.. help with large leaves (trace.hip.hpp)
    __shared__ GroupShared group_sh[1];
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (SLICES > 1 && wave_in_block != 0u) {
        group_helper_loop<SLICES>(A.tree, &group_sh[0], wave_in_block);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = SLICES > 1 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;    // work-unit id
    const uint32_t n_waves = SLICES > 1 ? gridDim.x : gridDim.x * (blockDim.x >> 6);
    const uint32_t n_hits = S.ws.ctrl[kCtrlHitCount + S.level];
    const uint32_t n_lights = (uint32_t)A.n_lights;
    const uint32_t n_items = ((n_hits + 63u) >> 6) * n_lights;
    const float PI_F = 3.14159265358979323846f;
    Stats st = {0, 0, 0, 0, 0, 0};
    SliceCtx sx = {SLICES > 1 ? &group_sh[0] : nullptr, A.slice_min_tris, 0u, true, 0u};
    uint32_t nrays = 0;

#ifdef RTK_DEBUG_WAVE_TIME
    const unsigned long long dbg_b0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t item = gwave; item < n_items; item = next_item(S.ws.ctrl + kCtrlTicket + (kMaxRayDepth + 2) + S.level, n_waves)) {
#ifdef RTK_DEBUG_WAVE_TIME
        const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t group = item / n_lights, k = item % n_lights;
        const uint32_t h = group * 64u + lane;
        const bool valid = h < n_hits;
        const float4 *q = reinterpret_cast<const float4 *>(S.ws.hits + (valid ? h : 0u));
        const float4 a = q[0], b = q[1];
        const V3 P = mk(a.x, a.y, a.z), ncos = mk(b.x, b.y, b.z);
        const DevLight *L = A.lights + k;                                    // wave-uniform
        V3 ld = mk(L->pos[0], L->pos[1], L->pos[2]) - P;
        const float radius = length(ld);
        const float area = 4.0f * PI_F * radius * radius;
        ld = normalized(ld);
        const float d0 = dot(ld, ncos);
        const float cosine = (0.0f < d0) ? d0 : 0.0f;                        // std::max(0, dot)
        const float contrib = (L->intensity / area) * cosine;
        const bool shoot = valid & (0.0f < radius);                          // is_occluded's loop guard, render.hpp:114
        const Ray ray = make_ray(P + (A.shadow_bias * ld), ld);
        Cand c;
        c.t = kFltMax; c.u = c.v = 0.f; c.k = kMiss;
        c = trace<RTK_TRACE_WAVE, STATS, false, SLICES>(A.tree, nullptr, ray, false, shoot, st, sx);
        nrays += shoot ? 1u : 0u;
        const bool clear = !shoot | (c.k == kMiss) | (radius < c.t);         // render.hpp:117 (no transmissive surface here)
        if (valid) S.ws.contrib[(size_t)h * n_lights + k] = make_float2(contrib, clear ? 1.0f : 0.0f);
#ifdef RTK_DEBUG_WAVE_TIME
        if (lane == 0u && S.level < 4u) {
            const uint32_t dt = (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_t0);
            uint32_t *d = S.ws.ctrl + kCtrlDebug + (2u * 4u + S.level) * 4u;
            atomicAdd(d + 0, dt); atomicMax(d + 1, dt); atomicAdd(d + 2, 1u);
        }
#endif
    }
#ifdef RTK_DEBUG_WAVE_TIME
    if (lane == 0u && S.level < 4u)
        atomicMax(S.ws.ctrl + kCtrlDebug + (2u * 4u + S.level) * 4u + 3, (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_b0));
#endif
    if (SLICES > 1) group_post_exit(&group_sh[0]);
    add_rays(S, st, nrays, STATS, gwave);
}

// ------------------------------------------------------------------------------------------------
// k_resolve: final_color += ((intensity / area) * cosine) * albedo for the unoccluded lights, in light order;
// final_color /= (diffuse_reflection_ray_count + 1) with a count of 0 (render.hpp:205-208).
__global__ __launch_bounds__(256) void k_resolve(StreamArgs S) {
    const RenderArgs &A = S.r;
    const uint32_t n_hits = S.ws.ctrl[kCtrlHitCount + S.level];
    const uint32_t n_lights = (uint32_t)A.n_lights;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n_hits; h += stride) {
        const float4 *q = reinterpret_cast<const float4 *>(S.ws.hits + h);
        const uint32_t pix = __float_as_uint(q[0].w), mat = __float_as_uint(q[1].w);
        const DevMaterial *m = A.materials + mat;
        const V3 albedo = mk(m->albedo[0], m->albedo[1], m->albedo[2]);
        V3 acc = mk(0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < n_lights; ++k) {
            const float2 cv = S.ws.contrib[(size_t)h * n_lights + k];
            if (cv.y != 0.0f) acc = acc + (cv.x * albedo);
        }
        const float div = (float)(A.diffuse_rays + 1);
        emit_pixel(S, pix, mk(acc.x / div, acc.y / div, acc.z / div));
    }
}

}  // namespace dev

// ------------------------------------------------------------------------------------------------ launchers

namespace {

template <bool LEVEL0, int SLICES>
void launch_path(const dev::StreamArgs &S, bool stats, unsigned units, hipStream_t s) {
    const unsigned blocks = SLICES > 1 ? units : (units + 3) / 4, threads = SLICES > 1 ? 64u * SLICES : 256u;
    if (stats) hipLaunchKernelGGL((dev::k_path<LEVEL0, true, SLICES>), dim3(blocks), dim3(threads), 0, s, S);
    else hipLaunchKernelGGL((dev::k_path<LEVEL0, false, SLICES>), dim3(blocks), dim3(threads), 0, s, S);
}
template <int SLICES>
void launch_shadow(const dev::StreamArgs &S, bool stats, unsigned units, hipStream_t s) {
    const unsigned blocks = SLICES > 1 ? units : (units + 3) / 4, threads = SLICES > 1 ? 64u * SLICES : 256u;
    if (stats) hipLaunchKernelGGL((dev::k_shadow<true, SLICES>), dim3(blocks), dim3(threads), 0, s, S);
    else hipLaunchKernelGGL((dev::k_shadow<false, SLICES>), dim3(blocks), dim3(threads), 0, s, S);
}

}  // namespace

// One sample of one frame.  `slices` = waves per 64-ray work unit (1 or 4).
hipError_t launch_stream_sample(const dev::StreamArgs &base, bool stats, int slices, hipStream_t s) {
    dev::StreamArgs S = base;
    const dev::RenderArgs &A = S.r;
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint64_t tiles = (uint64_t)A.buckets_per_rank * bpb;
    if (tiles == 0) return hipSuccess;
    if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(S.ws.ctrl, 0, dev::kCtrlWords * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    // queue-driven stages are persistent: a fixed number of work units' worth of waves stride over the queue
    // (8 waves per SIMD on 256 CUs = 8192 waves)
    const unsigned persist_units = slices > 1 ? 8192u / (unsigned)slices : 8192u;
    for (int level = 0; level <= A.max_depth; ++level) {
        S.level = (uint32_t)level;
        if (level == 0) {
            if (slices > 1) launch_path<true, 4>(S, stats, (unsigned)tiles, s);
            else launch_path<true, 1>(S, stats, (unsigned)tiles, s);
        } else {
            if (slices > 1) launch_path<false, 4>(S, stats, persist_units, s);
            else launch_path<false, 1>(S, stats, persist_units, s);
        }
        if (level < A.max_depth) {
            if (A.n_lights > 0) {
                if (slices > 1) launch_shadow<4>(S, stats, persist_units, s);
                else launch_shadow<1>(S, stats, persist_units, s);
            }
            hipLaunchKernelGGL(dev::k_resolve, dim3(1024), dim3(256), 0, s, S);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace rtk
