// Streaming ("wavefront") frame pipeline: color_hit's recursion (render/render.hpp:133-308) evaluated level by level.
//
// The megakernel (kernels.hip) walks each pixel's ray tree depth-first inside one lane: a pixel behind a refractive
// object needs hundreds of dependent traces while the other lanes of its wave idle.  Here every ray of a sample's
// ray tree is a NODE (stream.hpp) and all nodes of one depth are traced together:
//
//   k_path(d)     closest hit for every depth-d ray (camera rays at d = 0) + material switch.  Leaves get their
//                 value at once; reflective / refractive / diffuse-GI hits append their child rays as depth d+1
//                 nodes (compacted with a wave prefix sum and one atomic per wave); diffuse hits also append a
//                 shading point.
//   k_shadow(d)   one is_occluded query (render.hpp:110-131, stepping through transmissive surfaces) per
//                 (shading point, light)
//   k_combine(d)  bottom-up, d = max_depth-1 .. 0: inner nodes compute their value from their children in exactly the
//                 order the recursion uses (GI children first, then the lights in order, then the division; fresnel
//                 blend of the reflection and refraction child), so the floats equal the recursive evaluation bit
//                 for bit; depth 0 writes the pixels.
//
// Random numbers are keyed by tree position (common.hip.hpp), so the breadth-first order changes nothing.
// The trace kernels carry one ray + one candidate per lane (8 waves per SIMD), work units are drawn from atomic
// tickets, and a unit of 64 rays is served by an owner wave plus helper waves that split large leaves
// (trace.hip.hpp).  If a queue overflows its preallocated capacity (deep forks in an adversarial scene) the frame is
// redone by the megakernel — slower, never wrong.
#include <hip/hip_runtime.h>

#include "common.hip.hpp"
#include "stream.hpp"

namespace rtk {
namespace dev {

namespace {

__device__ __forceinline__ uint32_t lane_prefix(const unsigned long long mask) {      // set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave-compacted append of `n` records per lane (n may differ per lane): returns the lane's first slot relative to
// the queue's start, or 0xFFFFFFFF for every lane when the wave's records do not fit below `cap`.
__device__ __forceinline__ uint32_t wave_append_n(uint32_t *counter, const uint32_t n, const uint32_t base, const uint32_t cap,
                                                  uint32_t *overflow) {
    const uint32_t lane = __lane_id();
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += t;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0u) return 0u;
    uint32_t start = 0u;
    if (lane == 0u) start = atomicAdd(counter, total);
    start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
    if (base + start + total > cap) {
        if (lane == 0u) atomicExch(overflow, 1u);
        return 0xFFFFFFFFu;
    }
    return start + (incl - n);
}

__device__ __forceinline__ uint32_t wave_append_1(uint32_t *counter, const bool want, const uint32_t base, const uint32_t cap,
                                                  uint32_t *overflow) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return 0u;
    uint32_t start = 0u;
    if (__lane_id() == 0u) start = atomicAdd(counter, (uint32_t)__popcll(mask));
    start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
    if (base + start + (uint32_t)__popcll(mask) > cap) {
        if (__lane_id() == 0u) atomicExch(overflow, 1u);
        return 0xFFFFFFFFu;
    }
    return start + lane_prefix(mask);
}

// Histogram count, one atomic per wave and bin for the first few bins the lanes want (see bin_slot): wave-wide, call it from
// converged code with `want` as the predicate.
__device__ __forceinline__ void bin_count(uint32_t *bins, const uint32_t key, const bool want) {
    const uint32_t lane = __lane_id();
    unsigned long long rem = __builtin_amdgcn_ballot_w64(want);
    for (int round = 0; round < 4 && rem != 0ull; ++round) {
        const int leader = __builtin_ctzll(rem);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(want && key == k0) & rem;
        if ((int)lane == leader) atomicAdd(bins + k0, (uint32_t)__popcll(same));
        rem &= ~same;
    }
    if ((rem >> lane) & 1ull) atomicAdd(bins + key, 1u);
}

// Dynamic work distribution for the queue-driven stages: a unit's first item is its own index, every further item
// is drawn from an atomic ticket, so the stage ends when the queue is empty rather than when the unluckiest static
// share is done.
// waves per SIMD the trace kernels are built for (= their register budget: 8 -> 64 VGPRs, 6 -> 80, 5 -> 96, 4 -> 128)
#ifndef RTK_STREAM_WAVES
#define RTK_STREAM_WAVES 8
#endif
#ifndef RTK_TICKET_ITEMS
#define RTK_TICKET_ITEMS 4
#endif
constexpr uint32_t kTicketItems = RTK_TICKET_ITEMS;
__device__ __forceinline__ uint32_t next_item(uint32_t *ticket, const uint32_t n_units) {
    uint32_t t = 0u;
    if (__lane_id() == 0u) t = atomicAdd(ticket, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + n_units;
}

// first node id and node count of depth level `level`
__device__ __forceinline__ void level_range(const uint32_t *count_words, const uint32_t level0_count, const uint32_t level,
                                            const uint32_t cap, uint32_t &base, uint32_t &count) {
    unsigned long long b = 0ull;
    for (uint32_t j = 0; j < level; ++j) b += (j == 0u) ? level0_count : count_words[j];
    const unsigned long long c = (level == 0u) ? level0_count : count_words[level];
    // a counter that ran past the capacity (overflow) must never turn into an out-of-bounds node id
    base = b < cap ? (uint32_t)b : cap;
    count = (b + c <= cap) ? (uint32_t)c : cap - base;
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

__device__ __forceinline__ uint32_t grid_cell(const StreamArgs &S, const V3 p) {
    const float fx = (p.x - S.grid_lo[0]) * S.grid_scale[0], fy = (p.y - S.grid_lo[1]) * S.grid_scale[1],
                fz = (p.z - S.grid_lo[2]) * S.grid_scale[2];
    const uint32_t cx = fx > 0.f ? (fx < 15.f ? (uint32_t)fx : 15u) : 0u;        // NaN compares false -> cell 0
    const uint32_t cy = fy > 0.f ? (fy < 15.f ? (uint32_t)fy : 15u) : 0u;
    const uint32_t cz = fz > 0.f ? (fz < 15.f ? (uint32_t)fz : 15u) : 0u;
    return (cx << 8) | (cy << 4) | cz;
}
// 15 bits.  Frames without diffuse rays (key_dirs == 0): the origin's cell in a 16^3 grid over the scene and the
// direction's octant -- reflections and refractions leave a surface patch in a few directions, position is what tells
// them apart.  Frames with diffuse rays: an 8^3 origin grid and the direction's cell in a 4^3 grid over the L1-normalised
// direction, interleaved from the most significant bit down (p2 d1 p1 d0 p0 per axis triple) -- diffuse rays leave one cell in
// every direction of an octant, and made wide bundles (config 4's shape -6 %, config 5's -9 %; on the refractive dragon of
// config 3 the coarser origin grid costs 4 %, hence the switch).
__device__ __forceinline__ uint32_t ray_sort_key(const StreamArgs &S, const V3 o, const V3 d) {
    if (S.key_dirs == 0u) {
        const uint32_t octant = (d.x > 0.f ? 1u : 0u) | (d.y > 0.f ? 2u : 0u) | (d.z > 0.f ? 4u : 0u);
        return (octant << 12) | grid_cell(S, o);
    }
    const uint32_t c = grid_cell(S, o);                                     // 4 bits per axis; the top 3 are used
    const uint32_t px = (c >> 9) & 7u, py = (c >> 5) & 7u, pz = (c >> 1) & 7u;
    const float l1 = (__builtin_fabsf(d.x) + __builtin_fabsf(d.y)) + __builtin_fabsf(d.z);
    const float k = l1 > 0.f ? 2.0f / l1 : 0.f;
    const float fx = d.x * k + 2.0f, fy = d.y * k + 2.0f, fz = d.z * k + 2.0f;     // [0, 4]
    const uint32_t dx = fx > 0.f ? (fx < 3.f ? (uint32_t)fx : 3u) : 0u;          // NaN compares false -> cell 0
    const uint32_t dy = fy > 0.f ? (fy < 3.f ? (uint32_t)fy : 3u) : 0u;
    const uint32_t dz = fz > 0.f ? (fz < 3.f ? (uint32_t)fz : 3u) : 0u;
    auto tri = [](uint32_t a, uint32_t b, uint32_t cc, uint32_t bit) { return (((a >> bit) & 1u) << 2) | (((b >> bit) & 1u) << 1) | ((cc >> bit) & 1u); };
    return (tri(px, py, pz, 2u) << 12) | (tri(dx, dy, dz, 1u) << 9) | (tri(px, py, pz, 1u) << 6) | (tri(dx, dy, dz, 0u) << 3) | tri(px, py, pz, 0u);
}

__device__ __forceinline__ void store_ray(RayRec *dst, const V3 o, const V3 d, const uint32_t parent, const uint32_t pixel,
                                          const uint32_t key, const uint32_t info) {
    float4 *q = reinterpret_cast<float4 *>(dst);
    q[0] = make_float4(o.x, o.y, o.z, __uint_as_float(parent));
    q[1] = make_float4(d.x, d.y, d.z, __uint_as_float(pixel));
    q[2] = make_float4(__uint_as_float(key), __uint_as_float(info), 0.f, 0.f);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// k_path: LEVEL0 = camera rays, one work unit per 8x8 pixel block (same bucket / rank mapping as k_render);
// otherwise the depth-`level` nodes, units of 64 consecutive nodes drawn from a ticket.
template <bool LEVEL0, bool STATS, int SLICES, int MODE>
__global__ __launch_bounds__(256, RTK_STREAM_WAVES) void k_path(StreamArgs S) {
    const RenderArgs &A = S.r;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevNode *lds_nodes = reinterpret_cast<DevNode *>(smem);
    constexpr bool kStage = (MODE != RTK_TRACE_WAVE);          // the per-lane walk reads the node array from LDS
    if (kStage) {
        const float4 *src = reinterpret_cast<const float4 *>(A.tree.nodes);
        float4 *dst = reinterpret_cast<float4 *>(lds_nodes);
        for (uint32_t i = threadIdx.x; i < A.tree.n_nodes * 2u; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    // SLICES > 1: the workgroup's wave 0 owns the rays, waves 1.. help with large leaves (trace.hip.hpp)
    __shared__ GroupStorage<(SLICES > 1 ? SLICES : 1)> group_st;
    GroupShared *const group_sh = group_st.get();
    // SLICES > 1: role 0 = owner; the owner role rotates with the workgroup index so that owners spread over the SIMDs
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(SLICES > 1 ? ((threadIdx.x >> 6) + blockIdx.x) % (uint32_t)SLICES : (threadIdx.x >> 6)));
    if (SLICES > 1 && wave_in_block != 0u) {
        group_helper_loop<SLICES>(A.tree, group_sh, wave_in_block);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gunit = SLICES > 1 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint32_t n_units_grid = SLICES > 1 ? gridDim.x : gridDim.x * (blockDim.x >> 6);
    const V3 background = mk(A.background[0], A.background[1], A.background[2]);
    const V3 black = mk(0.f, 0.f, 0.f);
    const float PI_F = 3.14159265358979323846f;
    const uint32_t level = S.level;
    uint32_t *ctrl = S.ws.ctrl;
    if (ctrl[kCtrlOverflow] != 0u) {                                           // a queue overflowed: the megakernel redoes the frame
        if (SLICES > 1) group_post_exit(group_sh);
        return;
    }
    uint32_t base, count;
    level_range(ctrl + kCtrlNodeCount, S.n_level0, level, S.ws.node_cap, base, count);
    const uint32_t next_base = base + count;                                  // where depth level+1 starts
    uint32_t hit_base = 0u;
    for (uint32_t j = 0; j < level; ++j) hit_base += ctrl[kCtrlHitCount + j];
    const uint32_t n_items = (count + 63u) >> 6;
    Stats st = {0, 0, 0, 0, 0, 0};
    // bundles of the current trace (trace.hip.hpp "Bundle culling"): the group's shared block, or one area per wave when every wave owns rays
    __shared__ __attribute__((aligned(16))) float wave_bundles[SLICES > 1 ? 1 : 4][kMaxBundles * kBundleFloats];
    SliceCtx sx = {SLICES > 1 ? group_sh : nullptr, A.slice_min_tris, 0u, true, 0u,
                   SLICES > 1 ? group_sh->bundles : (MODE == RTK_TRACE_WAVE ? wave_bundles[(threadIdx.x >> 6) & 3u] : nullptr)};
    sx.rebundle = false;
    uint32_t nrays = 0;

    // (a ticket is good for kTicketItems consecutive work units: the ticket word is one address every wave of the kernel pulls on)
    // (only where there are many units per wave: a short queue is better spread over all waves, one unit each)
    const uint32_t kPer = (!LEVEL0 && n_items >= 16u * n_units_grid) ? kTicketItems : 1u;
    for (uint32_t tb = gunit; tb * kPer < n_items; tb = LEVEL0 ? n_items : next_item(ctrl + kCtrlTicket + level, n_units_grid))
    for (uint32_t item = tb * kPer; item < n_items && item < (tb + 1u) * kPer; ++item) {
        const bool in_range = item * 64u + lane < count;
        uint32_t node = base + item * 64u + lane;
        if (!LEVEL0 && S.nodes_sorted) {
            node = in_range ? S.ws.node_order[item * 64u + lane] : base;
            node = node < S.ws.node_cap ? node : S.ws.node_cap - 1u;          // never trust an index read from memory
        }
        bool valid = in_range;
        uint32_t pix = 0xFFFFFFFFu, key = 0u;
        bool miss_bg = true;
        Ray ray;
        if (LEVEL0) {
            const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
            const uint32_t n_blocks = S.n_root >> 6;                           // items [b * n_blocks, (b + 1) * n_blocks): sample b of the batch
            const uint32_t blk = item % n_blocks, in_batch = item / n_blocks;
            const uint32_t local_bucket = blk / bpb, sub = blk % bpb;
            const uint32_t bucket = rank_bucket((uint32_t)A.rank, local_bucket, (uint32_t)A.world, A.skew_q);
            const uint32_t bx = (bucket % A.tiles_x) * A.bucket, by = (bucket / A.tiles_x) * A.bucket;
            const uint32_t lx = (sub % A.blocks_per_bucket_side) * 8u + (lane & 7u);
            const uint32_t ly = (sub / A.blocks_per_bucket_side) * 8u + (lane >> 3);
            const uint32_t px = bx + lx, py = by + ly;
            valid = valid & (bucket < A.n_buckets) & (lx < A.bucket) & (ly < A.bucket) & (px < A.width) & (py < A.height);
            if (valid) pix = (uint32_t)A.out_index(local_bucket, lx, ly, px, py);
            key = root_key(pcg_hash(A.seed), py * A.width + px, (uint32_t)S.sample + in_batch);
            ray = camera_ray(A, px, py, key);
        } else {
            const float4 *q = reinterpret_cast<const float4 *>(S.ws.rays + node);
            const float4 a = q[0], b = q[1], c = q[2];
            pix = __float_as_uint(b.w);
            key = __float_as_uint(c.x);
            const uint32_t info = __float_as_uint(c.y);
            valid = valid & ((info & kRayValid) != 0u);
            miss_bg = (info & kRayMissBackground) != 0u;
            ray = make_ray(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z));
        }
        const Cand c = trace<MODE, STATS, kStage, SLICES>(A.tree, lds_nodes, ray, LEVEL0, valid, st, sx, S.auto_min_lanes);
        nrays += valid ? 1u : 0u;

        // ---- color_hit's material switch (render.hpp:133-308): node kind + the rays it spawns
        uint32_t kind = NODE_LEAF, nchild = 0u, aux = 0u, mat = 0u;
        V3 value = black, P = black, hn = black, ncos = black;
        V3 c0o = black, c0d = black, c1o = black, c1d = black;                  // explicit children (reflect / refract)
        bool c0_bg = false, push_hit = false;
        if (valid) {
            if (c.k == kMiss) value = miss_bg ? background : black;
            else if ((int)level == A.max_depth) value = background;                            // render.hpp:138-139
            else {
                const Surface s = reconstruct(A.tree, c);
                P = ray.o + (c.t * ray.d);
                hn = s.hit_normal;
                mat = s.material;
                const DevMaterial *m = A.materials + mat;
                const int mkind = m->kind;
                if (mkind == RTK_MAT_CONSTANT) value = mk(m->albedo[0], m->albedo[1], m->albedo[2]);
                else if (mkind == RTK_MAT_REFLECTIVE) {                                         // :239-250
                    c0d = ray.d - ((2.0f * dot(ray.d, hn)) * hn);
                    c0o = P + (A.reflection_bias * c0d);
                    c0_bg = true; kind = NODE_PASS; nchild = 1u;
                } else if (mkind == RTK_MAT_REFRACTIVE) {                                       // :252-301
                    V3 n = normalized(m->smooth ? hn : s.face_normal);
                    const V3 i = normalized(ray.d);
                    float eta_i = 1.0f, eta_r = m->ior;
                    if (0.0f < dot(i, n)) { const float tmp = eta_i; eta_i = eta_r; eta_r = tmp; n = neg(n); }
                    const float cos_i_n = -dot(i, n);
                    const float sin_i_n = __builtin_sqrtf(1.0f - cos_i_n * cos_i_n);
                    const V3 rd = i - ((2.0f * dot(i, n)) * n);
                    const V3 ro = P + (A.reflection_bias * rd);
                    if (eta_r / eta_i < sin_i_n) {                                              // total internal reflection
                        c0o = ro; c0d = rd; kind = NODE_PASS; nchild = 1u;
                    } else {
                        const float sin_r = ((sin_i_n * eta_i) / eta_r);
                        const float cos_r = __builtin_sqrtf(1.0f - sin_r * sin_r);
                        const V3 r = (cos_r * neg(n)) + (sin_r * normalized(i + (cos_i_n * n)));
                        const double x = (double)(1.0f + dot(i, n));                           // :300, x^5 in double
                        aux = __float_as_uint((float)(0.5 * (x * x * x * x * x)));
                        c0o = P + (A.refraction_bias * r); c0d = r;                             // child 0: refraction
                        c1o = ro; c1d = rd;                                                     // child 1: reflection
                        kind = NODE_REFR; nchild = 2u;
                    }
                } else if (mkind == RTK_MAT_TEXTURE) {                                          // :211-238
                    ncos = m->smooth ? hn : s.face_normal;
                    value = sample_texture(A.textures + m->texture, A.tri_uv + s.tri, c.u, c.v, A.tex_pixels);
                    kind = NODE_TEX; push_hit = true;
                } else {                                                                        // diffuse, :148-209
                    ncos = m->smooth ? hn : s.face_normal;
                    kind = NODE_DIFF; nchild = (uint32_t)A.diffuse_rays; push_hit = true;
                }
            }
        }
        // ---- append the children (depth level+1 nodes) and the shading point
        uint32_t child_slot = wave_append_n(ctrl + kCtrlNodeCount + level + 1u, nchild, next_base, S.ws.node_cap, ctrl + kCtrlOverflow);
        const uint32_t hit_slot = wave_append_1(ctrl + kCtrlHitCount + level, push_hit, hit_base, S.ws.hit_cap, ctrl + kCtrlOverflow);
        if (child_slot == 0xFFFFFFFFu || hit_slot == 0xFFFFFFFFu) { kind = NODE_LEAF; nchild = 0u; push_hit = false; child_slot = 0u; }
        const uint32_t first_child = next_base + child_slot;
        const bool spawn01 = kind == NODE_PASS || kind == NODE_REFR;
        if (spawn01) {
            store_ray(S.ws.rays + first_child, c0o, c0d, node, pix, child_key(key, 0u), kRayValid | (c0_bg ? kRayMissBackground : 0u));
            if (kind == NODE_REFR) store_ray(S.ws.rays + first_child + 1u, c1o, c1d, node, pix, child_key(key, 1u), kRayValid);
        }
        if (S.bin_children) {                                                                   // (wave-wide: bin_count)
            bin_count(S.ws.node_bins, ray_sort_key(S, c0o, c0d), spawn01);
            bin_count(S.ws.node_bins, ray_sort_key(S, c1o, c1d), kind == NODE_REFR);
        }
        const bool shade_point = kind == NODE_DIFF || kind == NODE_TEX;
        if (S.bin_hits) bin_count(S.ws.hit_bins, grid_cell(S, P), shade_point);
        if (shade_point) {
            aux = hit_base + hit_slot;
            float4 *q = reinterpret_cast<float4 *>(S.ws.hits + aux);
            q[0] = make_float4(P.x, P.y, P.z, __uint_as_float(node));
            q[1] = make_float4(ncos.x, ncos.y, ncos.z, __uint_as_float(mat));
        }
        // GI rays, :151-176.  The loop runs wave-wide (diffuse_rays is uniform; `mine` says whose ray it is) so that the
        // histogram of the spawned rays can be counted per wave (bin_count); a lane's arithmetic is what it was.
        if (wave_any(shade_point && nchild != 0u)) {
            for (uint32_t gi = 0; gi < (uint32_t)A.diffuse_rays; ++gi) {
                const bool mine = shade_point && gi < nchild;
                const V3 right = normalized(cross(ray.d, hn));
                const V3 up = hn;
                const V3 fwd = cross(right, up);
                float s1, c1, s2, c2;
                det_sincos(PI_F * urand_key(key, 2u + 2u * gi), s1, c1);
                V3 rv = mk(c1, s1, 0.0f);
                det_sincos(PI_F * urand_key(key, 3u + 2u * gi) * 2.0f, s2, c2);
                rv = mk(c2 * rv.x + 0.0f * rv.y + (-s2) * rv.z, 0.0f * rv.x + 1.0f * rv.y + 0.0f * rv.z,
                        s2 * rv.x + 0.0f * rv.y + c2 * rv.z);
                const V3 org = P + (A.reflection_bias * hn);
                const V3 dir = mk(right.x * rv.x + right.y * rv.y + right.z * rv.z, up.x * rv.x + up.y * rv.y + up.z * rv.z,
                                  fwd.x * rv.x + fwd.y * rv.y + fwd.z * rv.z);
                if (mine) store_ray(S.ws.rays + first_child + gi, org, dir, node, pix, child_key(key, gi), kRayValid);
                if (S.bin_children) bin_count(S.ws.node_bins, ray_sort_key(S, org, dir), mine);
            }
        }
        if (in_range) {
            float4 *q = reinterpret_cast<float4 *>(S.ws.nodes + node);
            q[0] = make_float4(value.x, value.y, value.z, __uint_as_float(kind));
            q[1] = make_float4(__uint_as_float(first_child), __uint_as_float(aux), __uint_as_float(nchild), __uint_as_float(pix));
        }
    }
    if (SLICES > 1) group_post_exit(group_sh);
    add_rays(S, st, nrays, STATS, gunit);
}

// ------------------------------------------------------------------------------------------------
// k_shadow: item = (group of 64 shading points of depth `level`, light).  Light loop body of render.hpp:184-206 up
// to the occlusion decision (is_occluded, :110-131); the contribution is stored and summed in light order later.
template <bool STATS, int SLICES, int MODE>
__global__ __launch_bounds__(256, RTK_STREAM_WAVES) void k_shadow(StreamArgs S) {
    const RenderArgs &A = S.r;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevNode *lds_nodes = reinterpret_cast<DevNode *>(smem);
    constexpr bool kStage = (MODE != RTK_TRACE_WAVE);
    if (kStage) {
        const float4 *src = reinterpret_cast<const float4 *>(A.occl_on ? A.occl.nodes : A.tree.nodes);
        float4 *dst = reinterpret_cast<float4 *>(lds_nodes);
        for (uint32_t i = threadIdx.x; i < A.tree.n_nodes * 2u; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    __shared__ GroupStorage<(SLICES > 1 ? SLICES : 1)> group_st;
    GroupShared *const group_sh = group_st.get();
    // SLICES > 1: role 0 = owner; the owner role rotates with the workgroup index so that owners spread over the SIMDs
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)(SLICES > 1 ? ((threadIdx.x >> 6) + blockIdx.x) % (uint32_t)SLICES : (threadIdx.x >> 6)));
    if (SLICES > 1 && wave_in_block != 0u) {
        group_helper_loop<SLICES>(A.occl_on ? A.occl : A.tree, group_sh, wave_in_block);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gunit = SLICES > 1 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint32_t n_units_grid = SLICES > 1 ? gridDim.x : gridDim.x * (blockDim.x >> 6);
    uint32_t *ctrl = S.ws.ctrl;
    if (ctrl[kCtrlOverflow] != 0u) {
        if (SLICES > 1) group_post_exit(group_sh);
        return;
    }
    uint32_t hit_base = 0u, n_hits = 0u;
    {
        unsigned long long hb = 0ull;
        for (uint32_t j = 0; j < S.level; ++j) hb += ctrl[kCtrlHitCount + j];
        const unsigned long long hc = ctrl[kCtrlHitCount + S.level];
        hit_base = hb < S.ws.hit_cap ? (uint32_t)hb : S.ws.hit_cap;
        n_hits = (hb + hc <= S.ws.hit_cap) ? (uint32_t)hc : S.ws.hit_cap - hit_base;
    }
    const uint32_t n_lights = (uint32_t)A.n_lights;
    const uint32_t n_items = ((n_hits + 63u) >> 6) * n_lights;
    const float PI_F = 3.14159265358979323846f;
    Stats st = {0, 0, 0, 0, 0, 0};
    // bundles of the current trace (trace.hip.hpp "Bundle culling"): the group's shared block, or one area per wave when every wave owns rays
    __shared__ __attribute__((aligned(16))) float wave_bundles[SLICES > 1 ? 1 : 4][kMaxBundles * kBundleFloats];
    SliceCtx sx = {SLICES > 1 ? group_sh : nullptr, A.slice_min_tris, 0u, true, 0u,
                   SLICES > 1 ? group_sh->bundles : (MODE == RTK_TRACE_WAVE ? wave_bundles[(threadIdx.x >> 6) & 3u] : nullptr)};
    sx.rebundle = false;
    uint32_t nrays = 0;

    const uint32_t kPer = n_items >= 16u * n_units_grid ? kTicketItems : 1u;           // (see k_path)
    for (uint32_t tb = gunit; tb * kPer < n_items; tb = next_item(ctrl + kCtrlTicket + kLevels + S.level, n_units_grid))
    for (uint32_t item = tb * kPer; item < n_items && item < (tb + 1u) * kPer; ++item) {
        const uint32_t group = item / n_lights, k = item % n_lights;
        const uint32_t hl = group * 64u + lane;
        const bool valid = hl < n_hits;
        uint32_t h = hit_base + (valid ? hl : 0u);
        if (S.hits_sorted) {
            h = valid ? S.ws.hit_order[hit_base + hl] : hit_base;          // (each level has its own stretch of hit_order: levels overlap in time)
            h = h < S.ws.hit_cap ? h : S.ws.hit_cap - 1u;
        }
        const float4 *q = reinterpret_cast<const float4 *>(S.ws.hits + h);
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
        Ray ray = make_ray(P + (A.shadow_bias * ld), ld);
        float max_t = radius;
        bool pending = valid & (0.0f < radius);                              // is_occluded's loop guard, render.hpp:114
        bool clear = true;
        while (wave_any(pending)) {
            // no transmissive material in the scene: the query may stop at the first hit nearer than the light (trace(), `exit_t`)
            // (occl_on: RTK_TRAVERSAL_FAST's tree of the opaque triangles -- nothing transmissive in it either)
            const float exit_t = (A.shadow_exit | A.occl_on) ? max_t : -1.0f;
            // every ray of the item ends in light k: a pencil bundle with that apex (trace.hip.hpp); rays that stepped through a
            // transmissive surface still lie on their line
            const Cand c = trace<MODE, STATS, kStage, SLICES>(A.occl_on ? A.occl : A.tree, lds_nodes, ray, false, pending, st, sx, S.auto_min_lanes, exit_t,
                                                              kClsHasApex | k, mk(L->pos[0], L->pos[1], L->pos[2]));
            if (pending) {
                nrays += 1u;
                bool clr = (c.k == kMiss) | (max_t < c.t);                   // :117
                bool again = false;
                if (!clr && A.has_refractive && !A.occl_on) {
                    const uint32_t m = A.tree.shade[A.tree.tri_ids[c.k]].material;
                    if (A.materials[m].kind == RTK_MAT_REFRACTIVE) {         // transmissive: step through, :126-127
                        const V3 hp = ray.o + (c.t * ray.d);
                        ray.o = hp + (A.shadow_bias * ray.d);
                        max_t -= c.t;
                        if (0.0f < max_t) again = true; else clr = true;
                    }
                }
                if (!again) { clear = clr; pending = false; }
            }
        }
        if (valid) S.ws.contrib[(size_t)h * n_lights + k] = make_float2(contrib, clear ? 1.0f : 0.0f);
    }
    if (SLICES > 1) group_post_exit(group_sh);
    add_rays(S, st, nrays, STATS, gunit);
}

// ------------------------------------------------------------------------------------------------
// Counting sort of a level's rays / shading points (see kSortBins): k_path filled the histogram while appending;
// k_sort_scan turns it into start offsets, k_sort_scatter_* hands every record a slot.
// (256 threads, not 1,024: the kernel is one workgroup in the middle of a level's dependency chain, and while the other
// samples' kernels fill the chip a 16-wave workgroup waits for a CU to drain -- 563 us on average on config 4's shape, against
// 15 us of work; a 4-wave one takes the next free slot.)
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *bins) {
    __shared__ uint32_t part[256];
    constexpr uint32_t per = kSortBins / 256u;
    static_assert(kSortBins % (256u * 4u) == 0u, "k_sort_scan reads four bins at a time");
    uint4 *b4 = reinterpret_cast<uint4 *>(bins + threadIdx.x * per);
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per / 4u; ++i) { const uint4 v = b4[i]; sum += (v.x + v.y) + (v.z + v.w); }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = 0; i < per / 4u; ++i) {
        const uint4 v = b4[i];
        uint4 o;
        o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run; run += v.w;
        b4[i] = o;
    }
}

// A record's slot in its bin: one returning atomic per record is what these kernels spend their time on (the records of a
// level arrive in the order their parents were processed, so the lanes of a wave mostly want the same few bins, and a
// returning atomic on one address is a ~1 us round trip that the others queue behind).  The lanes that share the first
// pending lane's bin are served by ONE atomic, for up to four bins; whoever is left after that takes its own.
__device__ __forceinline__ uint32_t bin_slot(uint32_t *bins, const uint32_t key, const bool have) {
    const uint32_t lane = __lane_id();
    unsigned long long rem = __builtin_amdgcn_ballot_w64(have);
    uint32_t slot = 0xFFFFFFFFu;
    for (int round = 0; round < 4 && rem != 0ull; ++round) {
        const int leader = __builtin_ctzll(rem);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(have && key == k0) & rem;
        uint32_t first = 0u;
        if ((int)lane == leader) first = atomicAdd(bins + k0, (uint32_t)__popcll(same));
        first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
        if ((same >> lane) & 1ull) slot = first + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        rem &= ~same;
    }
    if ((rem >> lane) & 1ull) slot = atomicAdd(bins + key, 1u);
    return slot;
}

__global__ __launch_bounds__(256) void k_sort_scatter_nodes(StreamArgs S) {
    if (S.ws.ctrl[kCtrlOverflow] != 0u) return;
    uint32_t base, count;
    level_range(S.ws.ctrl + kCtrlNodeCount, S.n_level0, S.level, S.ws.node_cap, base, count);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < count; i0 += stride) {           // (whole waves stay in the loop: bin_slot is wave-wide)
        const uint32_t i = i0 + threadIdx.x;
        const bool have = i < count;
        const float4 *q = reinterpret_cast<const float4 *>(S.ws.rays + base + (have ? i : 0u));
        const float4 a = q[0], b = q[1];
        const uint32_t slot = bin_slot(S.ws.node_bins, ray_sort_key(S, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z)), have);
        if (have && slot < count) S.ws.node_order[slot] = base + i;
    }
}

__global__ __launch_bounds__(256) void k_sort_scatter_hits(StreamArgs S) {
    const uint32_t *ctrl = S.ws.ctrl;
    if (ctrl[kCtrlOverflow] != 0u) return;
    unsigned long long hb = 0ull;
    for (uint32_t j = 0; j < S.level; ++j) hb += ctrl[kCtrlHitCount + j];
    const unsigned long long hc = ctrl[kCtrlHitCount + S.level];
    const uint32_t hit_base = hb < S.ws.hit_cap ? (uint32_t)hb : S.ws.hit_cap;
    const uint32_t n_hits = (hb + hc <= S.ws.hit_cap) ? (uint32_t)hc : S.ws.hit_cap - hit_base;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n_hits; i0 += stride) {
        const uint32_t i = i0 + threadIdx.x;
        const bool have = i < n_hits;
        const float4 a = *reinterpret_cast<const float4 *>(S.ws.hits + hit_base + (have ? i : 0u));
        const uint32_t slot = bin_slot(S.ws.hit_bins, grid_cell(S, mk(a.x, a.y, a.z)), have);
        if (have && slot < n_hits) S.ws.hit_order[hit_base + slot] = hit_base + i;
    }
}

// ------------------------------------------------------------------------------------------------
// k_combine: one thread per depth-`level` node.  Children (depth level+1) are final by now.
// The value of node `nd` from its children, in the recursion's operation order (stored back for inner nodes).
__device__ __forceinline__ V3 combine_node(const StreamArgs &S, NodeRes *nd, uint32_t &pix) {
    const RenderArgs &A = S.r;
    const uint32_t n_lights = (uint32_t)A.n_lights;
    const float4 *q = reinterpret_cast<const float4 *>(nd);
    const float4 a = q[0], b = q[1];
    const uint32_t kind = __float_as_uint(a.w), first = __float_as_uint(b.x), aux = __float_as_uint(b.y),
                   nchild = __float_as_uint(b.z);
    pix = __float_as_uint(b.w);
    V3 v = mk(a.x, a.y, a.z);
    if (kind == NODE_PASS) {                                               // render.hpp:249 / :275
        const NodeRes *c = S.ws.nodes + first;
        v = mk(c->value[0], c->value[1], c->value[2]);
    } else if (kind == NODE_REFR) {                                        // :301
        const NodeRes *c0 = S.ws.nodes + first, *c1 = c0 + 1;
        const float fresnel = __uint_as_float(aux);
        const V3 refr = mk(c0->value[0], c0->value[1], c0->value[2]), refl = mk(c1->value[0], c1->value[1], c1->value[2]);
        v = (fresnel * refl) + ((1.0f - fresnel) * refr);
    } else if (kind == NODE_TEX) {                                         // :211-238
        V3 acc = mk(0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < n_lights; ++k) {
            const float2 cv = S.ws.contrib[(size_t)aux * n_lights + k];
            if (cv.y != 0.0f) acc = acc + (cv.x * v);                      // v still holds the sampled texture colour
        }
        v = acc;
    } else if (kind == NODE_DIFF) {                                        // :151-208
        const HitRec *h = S.ws.hits + aux;
        const DevMaterial *m = A.materials + h->mat;
        const V3 albedo = mk(m->albedo[0], m->albedo[1], m->albedo[2]);
        V3 acc = mk(0.f, 0.f, 0.f);
        for (uint32_t g = 0; g < nchild; ++g) {                            // a GI ray that missed is worth 0: adding it changes nothing
            const NodeRes *c = S.ws.nodes + first + g;
            acc = acc + mk(c->value[0], c->value[1], c->value[2]);
        }
        for (uint32_t k = 0; k < n_lights; ++k) {
            const float2 cv = S.ws.contrib[(size_t)aux * n_lights + k];
            if (cv.y != 0.0f) acc = acc + (cv.x * albedo);
        }
        const float div = (float)(A.diffuse_rays + 1);
        v = mk(acc.x / div, acc.y / div, acc.z / div);
    }
    if (kind != NODE_LEAF) { nd->value[0] = v.x; nd->value[1] = v.y; nd->value[2] = v.z; }
    return v;
}

__global__ __launch_bounds__(256) void k_combine(StreamArgs S) {
    if (S.ws.ctrl[kCtrlOverflow] != 0u) return;                                // the megakernel redoes the frame
    if (S.level == 0u)                                                         // pixels are only touched while no sample of the frame has overflowed
        for (uint32_t j = 0; j < S.n_lanes; ++j) if (*S.lane_overflow[j] != 0u) return;
    uint32_t base, count;
    level_range(S.ws.ctrl + kCtrlNodeCount, S.n_level0, S.level, S.ws.node_cap, base, count);
    const uint32_t stride = gridDim.x * blockDim.x;
    if (S.level != 0u) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
            uint32_t pix;
            (void)combine_node(S, S.ws.nodes + base + i, pix);
        }
        return;
    }
    // depth 0: one thread per pixel slot; the camera rays of the batch's samples are nodes i, n_root + i, 2 n_root + i, ...
    // final_color += colour, sample after sample; after the last sample pixels[y][x] = final_color / spp (render.hpp:66-74).
    // The running sum of a pixel lives in ws.sumbuf between the batches of one pass and in the output buffer between the passes
    // of a progressive frame (rtk_render_params.sample_begin); it is added to in sample order either way.
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < S.n_root; i += stride) {
        V3 sum = mk(0.f, 0.f, 0.f);
        uint32_t pix = 0xFFFFFFFFu;
        for (uint32_t b = 0; b < S.n_batch; ++b) {
            uint32_t p;
            const V3 ret = combine_node(S, S.ws.nodes + (size_t)b * S.n_root + i, p);
            if (b == 0u) {
                pix = p;
                if (pix == 0xFFFFFFFFu) break;                             // not a pixel of the frame (block overhanging its edge)
                if (S.sample == 0) sum = mk(0.0f + ret.x, 0.0f + ret.y, 0.0f + ret.z);
                else {
                    const float *sb = (S.sample == S.r.sample_begin ? S.r.out : S.ws.sumbuf) + (size_t)pix * 3;
                    sum = mk(sb[0] + ret.x, sb[1] + ret.y, sb[2] + ret.z);
                }
            } else {
                sum = sum + ret;
            }
        }
        if (pix == 0xFFFFFFFFu) continue;
        if (S.sample + (int)S.n_batch == S.r.sample_end) {
            float *o = S.r.out + (size_t)pix * 3;
            if (S.r.sample_end == S.r.spp) {
                const float n = (float)S.r.spp;
                o[0] = sum.x / n; o[1] = sum.y / n; o[2] = sum.z / n;
            } else {
                o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
            }
        } else {
            float *sb = S.ws.sumbuf + (size_t)pix * 3;
            sb[0] = sum.x; sb[1] = sum.y; sb[2] = sum.z;
        }
    }
}

// zeroes the counters when the streamed frame overflowed (the megakernel that redoes it counts from scratch)
__global__ void k_reset_counters_if(unsigned long long *counters, StreamArgs S) {
    __shared__ uint32_t any;
    if (threadIdx.x == 0u) {
        uint32_t f = 0u;
        for (uint32_t j = 0; j < S.n_lanes; ++j) f |= *S.lane_overflow[j];
        S.ws.ctrl[kCtrlOverflow] = f;                                           // lane 0's word is what the fallback launch looks at
        any = f;
    }
    __syncthreads();
    if (any != 0u && threadIdx.x < (unsigned)kCounterWords) counters[threadIdx.x] = 0ull;
}

}  // namespace dev

// ------------------------------------------------------------------------------------------------ launchers

namespace {

template <bool LEVEL0, int SLICES, int MODE>
void launch_path(const dev::StreamArgs &S, bool stats, unsigned units, hipStream_t s) {
    const unsigned blocks = SLICES > 1 ? units : (units + 3) / 4, threads = SLICES > 1 ? 64u * SLICES : 256u;
    const size_t lds = MODE != RTK_TRACE_WAVE ? (size_t)S.r.tree.n_nodes * sizeof(DevNode) : 0;
    if (stats) hipLaunchKernelGGL((dev::k_path<LEVEL0, true, SLICES, MODE>), dim3(blocks), dim3(threads), lds, s, S);
    else hipLaunchKernelGGL((dev::k_path<LEVEL0, false, SLICES, MODE>), dim3(blocks), dim3(threads), lds, s, S);
}
template <int SLICES, int MODE>
void launch_shadow(const dev::StreamArgs &S, bool stats, unsigned units, hipStream_t s) {
    const unsigned blocks = SLICES > 1 ? units : (units + 3) / 4, threads = SLICES > 1 ? 64u * SLICES : 256u;
    const size_t lds = MODE != RTK_TRACE_WAVE ? (size_t)S.r.tree.n_nodes * sizeof(DevNode) : 0;
    if (stats) hipLaunchKernelGGL((dev::k_shadow<true, SLICES, MODE>), dim3(blocks), dim3(threads), lds, s, S);
    else hipLaunchKernelGGL((dev::k_shadow<false, SLICES, MODE>), dim3(blocks), dim3(threads), lds, s, S);
}

}  // namespace

// One batch of samples (S.sample .. S.sample + S.n_batch - 1) of one frame.  Depth levels below `deep_level` use the workgroup-cooperative wave walk (coherent rays);
// from `deep_level` on `deep_mode` (RTK_TRACE_AUTO / _LANE / _WAVE) applies.  From depth `sort_from_level` on, the
// level's rays and shading points are counting-sorted for coherence before they are cut into 64-ray work units.
hipError_t launch_stream_sample(const dev::StreamArgs &base, bool stats, int deep_level, int deep_mode, int sort_from_level,
                                hipStream_t s, hipEvent_t wait_before_emit, hipEvent_t done, const StreamSide *side, int slices) {
    dev::StreamArgs S = base;
    const dev::RenderArgs &A = S.r;
    if (S.n_root == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(S.ws.ctrl, 0, dev::kCtrlOverflow * sizeof(uint32_t), s);      // keeps the overflow word
    if (e != hipSuccess) return e;
    if ((size_t)A.tree.n_nodes * sizeof(DevNode) > kMaxNodeLdsBytes) deep_mode = RTK_TRACE_WAVE;
    if (sort_from_level < 1) sort_from_level = 1;
    if (sort_from_level <= A.max_depth) {
        e = hipMemsetAsync(S.ws.node_bins, 0, dev::kSortBins * sizeof(uint32_t), s);
        if (e == hipSuccess) e = hipMemsetAsync(S.ws.hit_bins, 0, dev::kSortBins * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    // queue-driven stages: 8192 waves in flight (8 per SIMD on 256 CUs)
    const unsigned group_units = 2048u, wave_units = 8192u;
    for (int level = 0; level <= A.max_depth; ++level) {
        S.level = (uint32_t)level;
        const bool deep = level >= deep_level && deep_mode != RTK_TRACE_WAVE;
        S.nodes_sorted = (level >= sort_from_level) ? 1u : 0u;
        S.hits_sorted = (level >= sort_from_level && level < A.max_depth) ? 1u : 0u;
        S.bin_children = (level + 1 >= sort_from_level && level < A.max_depth) ? 1u : 0u;
        S.bin_hits = S.hits_sorted;
        if (level == 0 && slices == 1) launch_path<true, 1, RTK_TRACE_WAVE>(S, stats, S.n_level0 / 64u, s);
        else if (level == 0) launch_path<true, 4, RTK_TRACE_WAVE>(S, stats, S.n_level0 / 64u, s);
        else if (!deep && slices == 2) launch_path<false, 2, RTK_TRACE_WAVE>(S, stats, group_units * 2u, s);
        else if (!deep && slices == 1) launch_path<false, 1, RTK_TRACE_WAVE>(S, stats, wave_units, s);
        else if (!deep) launch_path<false, 4, RTK_TRACE_WAVE>(S, stats, group_units, s);
        else if (deep_mode == RTK_TRACE_LANE) launch_path<false, 1, RTK_TRACE_LANE>(S, stats, wave_units, s);
        else launch_path<false, 1, RTK_TRACE_AUTO>(S, stats, wave_units, s);
        if (level < A.max_depth && A.n_lights > 0) {
            if (S.hits_sorted) {
                hipLaunchKernelGGL(dev::k_sort_scan, dim3(1), dim3(256), 0, s, S.ws.hit_bins);
                hipLaunchKernelGGL(dev::k_sort_scatter_hits, dim3(1024), dim3(256), 0, s, S);
                e = hipMemsetAsync(S.ws.hit_bins, 0, dev::kSortBins * sizeof(uint32_t), s);
                if (e != hipSuccess) return e;
            }
            // k_shadow(level) only feeds k_combine: on a side stream it overlaps the path kernels of the deeper levels (and the
            // other side stream's k_shadow), instead of making them wait for its slowest work unit
            hipStream_t ss = s;
            if (side != nullptr) {
                const int par = level & 1;
                ss = side->stream[par];
                e = hipEventRecord(side->ready[par], s);
                if (e == hipSuccess) e = hipStreamWaitEvent(ss, side->ready[par], 0);
                if (e != hipSuccess) return e;
            }
            if (!deep && slices == 2) launch_shadow<2, RTK_TRACE_WAVE>(S, stats, group_units * 2u, ss);
            else if (!deep && slices == 1) launch_shadow<1, RTK_TRACE_WAVE>(S, stats, wave_units, ss);
            else if (!deep) launch_shadow<4, RTK_TRACE_WAVE>(S, stats, group_units, ss);
            else if (deep_mode == RTK_TRACE_LANE) launch_shadow<1, RTK_TRACE_LANE>(S, stats, wave_units, ss);
            else launch_shadow<1, RTK_TRACE_AUTO>(S, stats, wave_units, ss);
        }
        if (S.bin_children) {                                  // order the next level's rays
            dev::StreamArgs N = S;
            N.level = (uint32_t)level + 1u;
            hipLaunchKernelGGL(dev::k_sort_scan, dim3(1), dim3(256), 0, s, S.ws.node_bins);
            hipLaunchKernelGGL(dev::k_sort_scatter_nodes, dim3(1024), dim3(256), 0, s, N);
            e = hipMemsetAsync(S.ws.node_bins, 0, dev::kSortBins * sizeof(uint32_t), s);
            if (e != hipSuccess) return e;
        }
    }
    if (side != nullptr) {                                       // k_combine needs every k_shadow
        for (int par = 0; par < 2; ++par) {
            e = hipEventRecord(side->done[par], side->stream[par]);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, side->done[par], 0);
            if (e != hipSuccess) return e;
        }
    }
    for (int level = A.max_depth > 0 ? A.max_depth - 1 : 0; level >= 0; --level) {
        S.level = (uint32_t)level;
        if (level == 0 && wait_before_emit != nullptr) {                    // the running pixel sums take the samples in order
            e = hipStreamWaitEvent(s, wait_before_emit, 0);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(dev::k_combine, dim3(level == 0 ? 2048 : 1024), dim3(256), 0, s, S);
    }
    if (done != nullptr) {
        e = hipEventRecord(done, s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t launch_stream_overflow_reset(const dev::StreamArgs &S, hipStream_t s) {
    hipLaunchKernelGGL(dev::k_reset_counters_if, dim3(1), dim3(256), 0, s, S.r.counters, S);
    return hipGetLastError();
}

}  // namespace rtk
