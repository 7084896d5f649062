// HIP kernels for gfx950: batched closest-hit intersect, the device-side render loop
// (render_frame / color_hit / is_occluded of render/render.hpp), and the multi-GPU bucket assembly.
// One ray per lane, 256-thread workgroups (4 wave64), kd-tree nodes staged in LDS.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "trace.hip.hpp"
#include "common.hip.hpp"

namespace rtk {
namespace dev {

// ------------------------------------------------------------------------------------------------
// node staging: the whole traversal-ordered node array goes to LDS with coalesced 16-byte loads.
__device__ __forceinline__ void stage_nodes(const DevNode *g_nodes, uint32_t n_nodes, DevNode *lds_nodes) {
    const float4 *src = reinterpret_cast<const float4 *>(g_nodes);
    float4 *dst = reinterpret_cast<float4 *>(lds_nodes);
    for (uint32_t i = threadIdx.x; i < n_nodes * 2u; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ void flush_stats(const Stats &st, uint32_t rays, unsigned long long *counters) {
    // counters: rays, primary, hits, nodes, boxpass, leaves, tris, packets16 (rtk_counters order)
    const uint32_t r = wave_sum(rays), h = wave_sum(st.hits), nd = wave_sum(st.nodes), bp = wave_sum(st.boxpass),
                   lv = wave_sum(st.leaves), tr = wave_sum(st.tris), pk = wave_sum(st.packets16);
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(counters + 0, (unsigned long long)r);
        atomicAdd(counters + 2, (unsigned long long)h);
        atomicAdd(counters + 3, (unsigned long long)nd);
        atomicAdd(counters + 4, (unsigned long long)bp);
        atomicAdd(counters + 5, (unsigned long long)lv);
        atomicAdd(counters + 6, (unsigned long long)tr);
        atomicAdd(counters + 7, (unsigned long long)pk);
    }
}

// ------------------------------------------------------------------------------------------------
// accel.intersect<cull>(ray) for a batch (kd_tree_simd.hpp:187-264): lane i takes ray i.
template <int MODE, bool STATS, bool LDS_NODES>
__global__ __launch_bounds__(256) void k_intersect(IntersectArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevNode *lds_nodes = reinterpret_cast<DevNode *>(smem);
    if (LDS_NODES) stage_nodes(A.tree.nodes, A.tree.n_nodes, lds_nodes);

    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = slot < A.n;
    size_t i = (active && A.perm != nullptr) ? (size_t)A.perm[slot] : slot;            // repacked batches: the ray this slot was dealt
    uint32_t raster_w = A.raster_w;
    if (A.verdict != nullptr) {                                                        // (wave-uniform scalar loads)
        if (A.verdict[16] != 0u) return;                                               // to be sorted: the host launches again
        raster_w = A.verdict[15];
        if (raster_w < 64u || (raster_w & 7u) != 0u || (size_t)raster_w * 16u > A.n) raster_w = 0u;
    }
    if (raster_w != 0u) {
        // a raster of rows: bands of 8 rows are dealt as 8x8 blocks, one per wave; what is left over after the last whole band stays as it is
        const size_t band = (size_t)raster_w * 8u;
        const size_t tiled = (A.n / band) * band;
        if (slot < tiled) {
            const size_t b = slot / band;
            const uint32_t within = (uint32_t)(slot % band), tile = within >> 6, l = within & 63u;
            i = (b * 8u + (l >> 3)) * (size_t)raster_w + (size_t)tile * 8u + (l & 7u);
        }
    }
    Ray r;
    if (active) {
        const float *p = reinterpret_cast<const float *>(A.rays + i);
        r = make_ray(mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]));
    } else {
        r = make_ray(mk(0.f, 0.f, 0.f), mk(1.f, 1.f, 1.f));
    }
    Stats st = {0, 0, 0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) float wave_bundles[4][kMaxBundles * kBundleFloats];
    SliceCtx sx = {nullptr, 0u, 0u, true, 0u, wave_bundles[(threadIdx.x >> 6) & 3u]};
    const Cand c = trace<MODE, STATS, LDS_NODES>(A.tree, lds_nodes, r, A.cull != 0, active, st, sx);
    if (active) {
        float4 o0, o1;
        if (c.k != kMiss) {
            const Surface s = reconstruct(A.tree, c);
            o0 = make_float4(c.t, c.u, c.v, __uint_as_float(s.tri));
            o1 = make_float4(__uint_as_float(s.mesh), s.hit_normal.x, s.hit_normal.y, s.hit_normal.z);
        } else {
            o0 = make_float4(-1.0f, 0.0f, 0.0f, __uint_as_float(kMiss));
            o1 = make_float4(__uint_as_float(kMiss), 0.0f, 0.0f, 0.0f);
        }
        float4 *out = reinterpret_cast<float4 *>(A.out + i);
        out[0] = o0;
        out[1] = o1;
    }
    if (STATS) flush_stats(st, active ? 1u : 0u, A.counters);
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Device-side render loop.  One lane = one pixel; a wave = an 8x8 pixel block of a bucket, buckets are
// dealt to ranks round-robin (bucket i -> rank i % world).  color_hit's recursion (render.hpp:133-308) is
// run as a per-lane state machine around ONE wave-wide trace call per iteration; refractive forks and
// diffuse-GI parents keep their partial results in a small per-lane frame stack, evaluated in the same
// order as the recursion so the floating-point results are identical.

// "Light burst": the occlusion queries of a diffuse hit towards its lights (render.hpp:184-206) are independent of each other.
// When every lane that is about to query a light is at the SAME light k, the owner posts GROUP_EXTRA and each helper wave
// traces the whole query of one further light (k + 1, k + 2, ...) on its own -- a single-class pencil bundle per wave, no
// slicing -- while the owner traces light k (and whatever other rays its lanes have pending).  The owner then adds the
// contributions in light order, so the float sums are those of the sequential loop.  Four dependent traces become one step.
// Scenes with transmissive materials keep the sequential loop (a query there may need several segments).
// The lanes of a burst are dealt to 1, 2 or 4 jobs per light by WHERE their rays go (the widest axis of their directions cut
// at its middle, the halves once more): a job's time is set by the triangles inside the cone of its rays, so two jobs that
// each take every other row of the pixel block would each cost as much as the whole (measured), two halves of the cone half.
// (Halves of equal extent, not of equal counts: a half with a few far-out rays has the whole cone again -- measured too.)
// Every wave computes the same partition from the same rays.
__device__ __forceinline__ bool upper_half(const V3 d, const bool sel) {
    Ray q;
    q.o = d; q.d = d; q.inv = d;                                            // (only q.d is read)
    const DirSpread S = dir_spread(q, sel);
    const float dv = S.axis == 0u ? d.x : (S.axis == 1u ? d.y : d.z);
    return sel & !(dv <= S.mid);
}
__device__ __forceinline__ uint32_t burst_part(const V3 d, const bool in, const uint32_t plog) {
    uint32_t part = 0u;
    if (plog == 0u) return part;
    if (upper_half(d, in)) part = 1u;
    if (plog >= 2u) {
#pragma unroll
        for (uint32_t h = 0u; h < 2u; ++h)
            if (upper_half(d, in & (part == h))) part |= 2u;
    }
    return part;
}

template <int SLICES>
struct ShadowBurstService {
    const RenderArgs &A;
    float4 *res;             // [jobs][64] answers: contribution, clear, flags (bit 0: a ray was traced; bit 1: this job answers for the lane)
    // Jobs (light, part) are handed out through a counter in LDS: a wave that is done takes the next one, so unequal jobs
    // (one half of a cone over the mesh, the other past it) even out.  Job 0 is the owner's own.
    __device__ __forceinline__ void operator()(GroupShared *sh, const uint32_t slice) const {
        const uint32_t lane = __lane_id();
        const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->count);
        const uint32_t plog = hdr & 0xFFu, n_jobs = hdr >> 8;                                   // log2(parts); lights x parts
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->first);
        const unsigned long long mask = sh->pass_mask;
        const float4 a = sh->ray_o[lane], b = sh->ray_d[lane];
        const V3 P = mk(a.x, a.y, a.z), ncos = mk(a.w, b.x, b.y);
        const bool in_burst = ((mask >> lane) & 1ull) != 0ull;
        for (;;) {
            uint32_t job = 0u;
            if (lane == 0u) job = atomicAdd(&sh->pad[0], 1u);
            job = (uint32_t)__builtin_amdgcn_readfirstlane((int)job);
            if (job >= n_jobs) break;
            const uint32_t k = first + (job >> plog);                        // this job's light (< n_lights: the owner counted the jobs)
            const uint32_t part = job & ((1u << plog) - 1u);
            const float PI_F = 3.14159265358979323846f;
            const DevLight *L = A.lights + k;                                // wave-uniform
            const V3 lp = mk(L->pos[0], L->pos[1], L->pos[2]);
            V3 ld = lp - P;                                                  // the light loop body of k_render's ST_LIGHT, verbatim
            const float radius = length(ld);
            const float area = 4.0f * PI_F * radius * radius;
            ld = normalized(ld);
            const float d0 = dot(ld, ncos);
            const float cosine = (0.0f < d0) ? d0 : 0.0f;
            const float contrib = (L->intensity / area) * cosine;
            const bool mine = in_burst & (burst_part(ld, in_burst, plog) == part);
            const bool q = mine & (0.0f < radius);                           // is_occluded's loop guard, render.hpp:114
            const Ray ray = make_ray(P + (A.shadow_bias * ld), ld);
            Stats st = {0, 0, 0, 0, 0, 0};
            SliceCtx sx = {nullptr, 0xFFFFFFFFu, 0u, true, 0u, group_private_bundles<SLICES>(sh, slice)};
            const float exit_t = A.shadow_exit ? radius : -1.0f;
            const Cand c = trace<RTK_TRACE_WAVE, false, false, 1>(A.tree, nullptr, ray, false, q, st, sx, kAutoMinLanes, exit_t,
                                                                  kClsHasApex | (0x100u + k), lp);
            bool clear = true;
            uint32_t fl = mine ? 2u : 0u;
            if (q) { clear = (c.k == kMiss) | (radius < c.t); fl |= 1u; }   // render.hpp:117
            res[job * 64u + lane] = make_float4(contrib, clear ? 1.0f : 0.0f, __uint_as_float(fl), 0.0f);
        }
    }
};

// The lane index, computed where it is asked for.  From threadIdx / __lane_id() the compiler derives a dozen per-lane
// constants in the kernel's prologue (pixel coordinates as floats, the pixel's RNG hash, its output address, one LDS
// address per array and stride, 1 << lane as a 64-bit pair ...), keeps them across the traversal loops and, out of registers
// there, spills every one of them to scratch right away: 22 dwords per lane written by each of a frame's 32,400 owner waves,
// 180 MB of HBM traffic per launch.  An `asm volatile` is opaque and is not hoisted: two instructions at each use instead.
__device__ __forceinline__ uint32_t fresh_lane() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// a ray whose inv_direction is filled in by k_render right before the query (see there)
__device__ __forceinline__ Ray spawn_ray(const V3 o, const V3 d) {
    Ray r;
    r.o = o; r.d = d; r.inv = d;
    return r;
}

enum : int { ST_NEW_SAMPLE = 0, ST_TRACE, ST_SHADE, ST_LIGHT, ST_RETURN, ST_DONE };
enum : int { PEND_CHILD_BG = 0, PEND_CHILD_BLACK = 1, PEND_SHADOW = 2 };
enum : uint32_t { FR_REFR_A = 0, FR_REFR_B = 1, FR_DIFFUSE = 2 };

struct Frame {            // 15 dwords, lives in scratch; touched only at refractive / GI events
    float a[12];
    uint32_t meta;        // kind | depth << 8 | iteration << 16
    uint32_t tri;
    uint32_t key;         // RNG key of the ray whose hit this frame shades
};

template <int MODE, bool STATS, bool FORKS, bool LDS_NODES, int SLICES, bool PRIMED = false>
// GROUP4 without per-ray statistics is built for RTK_G4_WAVES (5) waves per SIMD = 96 VGPRs: a fifth resident workgroup per CU.
// The lean build (FORKS = false) fits with its cold state parked in LDS (`park_lds`); 4 waves / 128 VGPRs and 6 / 80 are slower.
__global__ __launch_bounds__(SLICES > 1 ? 64 * SLICES : 256, SLICES == 16 ? 1 : SLICES == 8 ? 2 : (SLICES == 4 && !STATS ? RTK_G4_WAVES : 4)) void k_render(RenderArgs A) {
    // PRIMED (second pass of a two-pass frame): pixel blocks come from tile_order (most expensive first) and the
    // camera ray's hit is read from A.prim instead of being traced again
#ifdef RTK_DEBUG_PHASES
    const unsigned long long ph_entry = __builtin_readcyclecounter();
    const unsigned long long ph_rt0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz, the same clock on every CU
#endif
    if (PRIMED && blockIdx.x >= *A.n_listed) return;
    if (A.only_if != nullptr && *A.only_if == 0u) return;        // fallback launch behind the streaming pipeline: nothing overflowed
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DevNode *lds_nodes = reinterpret_cast<DevNode *>(smem);
    // the per-lane path reads nodes from LDS; the wave-cooperative paths fetch them with scalar loads instead
    constexpr bool kStage = LDS_NODES && (MODE == RTK_TRACE_LANE || MODE == RTK_TRACE_AUTO);
    if (kStage) stage_nodes(A.tree.nodes, A.tree.n_nodes, lds_nodes);
    __shared__ GroupStorage<(SLICES > 1 ? SLICES : 1)> group_st;
    GroupShared *const group_sh = group_st.get();
    // the wave index is wave-uniform: say so (readfirstlane) or everything derived from it is compiled per-lane
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // The owner role rotates with the workgroup index: the dispatcher places wave i of every workgroup on SIMD i, so a
    // fixed owner wave would put every owner of a CU on the same SIMD and leave the other three to the (mostly idle) helpers.
    uint32_t slice = SLICES > 1 ? (wave_in_wg + blockIdx.x) % (uint32_t)SLICES : 0u;
    // ---- which 8x8 pixel block (tile/bucket.hpp:7-21 buckets, 8x8 blocks inside, round-robin over ranks).
    // SLICES > 1: the whole workgroup serves ONE block (one wave owns the rays, the others help with big leaves) -- except
    // for the blocks the cost feedback found cheap (background, a few nodes): those are packed SLICES to a workgroup,
    // every wave renders its own and nobody helps ("light" workgroups; they never touch the barrier protocol).
    // (Cutting the most expensive blocks into 4x4-pixel quadrants was tried and dropped: a quadrant's rays visit the same
    // leaves as the whole block's, so it cost as much as the block.)
    uint32_t gwave;
    bool light = false;
    if (PRIMED) {
        gwave = A.tile_order[blockIdx.x];
    } else if (SLICES > 1 && A.order_in != nullptr) {
        const uint32_t n_wgs = A.order_hdr[0], n_total = A.order_hdr[1];
        if (blockIdx.x >= n_wgs) return;
        const uint32_t desc = A.wg_list[blockIdx.x];                        // k_order_by_cost: workgroups by expected duration
        if ((desc >> 31) == 0u) {
            gwave = desc;
        } else {
            const uint32_t idx = (desc & 0x7FFFFFFFu) + wave_in_wg;
            if (idx >= n_total) return;
            gwave = A.order_in[idx];
            light = true;
            slice = 0u;
        }
    } else {
        const uint32_t unit = SLICES > 1 ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
        gwave = (A.order_in != nullptr && unit < A.n_units) ? A.order_in[unit] : unit;
    }
    // one parking area per wave that can own rays: every wave of a light or SLICES == 1 workgroup, one otherwise -- the
    // others then hold the answers of a light burst's jobs (ShadowBurstService)
    constexpr int kParkSlots = 4;                        // light workgroups exist for SLICES == 4 only (api.hip)
    __shared__ __attribute__((aligned(16))) float park_lds[kParkSlots][18][64];
    float4 *const burst_res = reinterpret_cast<float4 *>(&park_lds[1][0][0]);
    // one job per wave: twice as many, handed out as waves fall idle, came out slower (each job pays its own ray setup,
    // bundles and leaf-list pass, and the owner only ever traces job 0): config 2 0.21 -> 0.255 ms, one rank of eight 0.155 -> 0.18
    constexpr uint32_t kBurstMaxJobs = SLICES > 8 ? 8u : SLICES > 1 ? (uint32_t)SLICES : 1u;
    static_assert(SLICES <= 1 || (size_t)kBurstMaxJobs * 64 * sizeof(float4) <= sizeof(float) * (kParkSlots - 1) * 18 * 64, "burst answers fit the spare parking slots");
    if (SLICES > 1 && slice != 0u) {                 // helper waves (trace.hip.hpp, "Workgroup-cooperative leaves")
        group_helper_loop<SLICES>(A.tree, group_sh, slice, ShadowBurstService<SLICES>{A, burst_res});
        return;
    }
    // bundles of the current trace (trace.hip.hpp "Bundle culling"): in the group's shared block when helpers must see them,
    // wave-private otherwise (light workgroups and SLICES == 1 have one owner per wave)
    __shared__ __attribute__((aligned(16))) float wave_bundles[4][kMaxBundles * kBundleFloats];
    SliceCtx sx = {SLICES > 1 ? group_sh : nullptr, light ? 0xFFFFFFFFu : A.slice_min_tris, 0u, true, 0u,
                   (SLICES > 1 && !light) ? group_sh->bundles : wave_bundles[wave_in_wg & 3u]};
    const uint32_t park_slot = (SLICES > 1 && !light) ? 0u : (wave_in_wg & 3u);
    const unsigned long long cost_t0 = __builtin_readcyclecounter();
    const unsigned long long real_t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, for the frame's critical path (bench.py)
    constexpr bool writer = true;
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint32_t local_bucket = gwave / bpb, sub = gwave % bpb;
    const uint32_t bucket = rank_bucket((uint32_t)A.rank, local_bucket, (uint32_t)A.world, A.skew_q);
    bool valid = bucket < A.n_buckets;
    const uint32_t bx = (bucket % A.tiles_x) * A.bucket, by = (bucket / A.tiles_x) * A.bucket;
    const uint32_t sub_x0 = (sub % A.blocks_per_bucket_side) * 8u, sub_y0 = (sub / A.blocks_per_bucket_side) * 8u;   // wave-uniform
    // this lane's pixel: recomputed where it is needed (see fresh_lane) instead of living in registers across the traversal
    struct Pixel { uint32_t lx, ly, px, py; };
    const auto my_pixel = [&]() {
        const uint32_t l = fresh_lane();
        Pixel p;
        p.lx = sub_x0 + (l & 7u); p.ly = sub_y0 + (l >> 3);
        p.px = bx + p.lx; p.py = by + p.ly;
        return p;
    };
    {
        const Pixel p = my_pixel();
        valid = valid & (p.lx < A.bucket) & (p.ly < A.bucket) & (p.px < A.width) & (p.py < A.height);
    }

#ifdef RTK_DEBUG_PHASES
    const unsigned long long ph_begin = __builtin_readcyclecounter();
#endif
    const V3 background = mk(A.background[0], A.background[1], A.background[2]);
    const V3 black = mk(0.f, 0.f, 0.f);
    const float PI_F = 3.14159265358979323846f;
    const uint32_t seed_hash = pcg_hash(A.seed);

    // ---- per-lane path state
    int state = valid ? ST_NEW_SAMPLE : ST_DONE;
    int pend = PEND_CHILD_BG;
    int sample = A.sample_begin, depth = 0, light_k = 0;
    uint32_t rkey = 0;                    // RNG key of the ray in flight (position in the sample's ray tree)
    uint32_t nrays_wave = 0;              // rays of the whole wave so far (wave-uniform: a scalar, not a register per lane)
    bool cull = false;
    Ray ray = spawn_ray(black, mk(1.f, 1.f, 1.f));
    V3 pixel_sum = black, ret = black;
    if (A.sample_begin > 0 && valid) {                                     // a later pass of a progressive frame: the running sum so far
        const Pixel p = my_pixel();
        const float *o = A.out + A.out_index(local_bucket, p.lx, p.ly, p.px, p.py) * 3;
        pixel_sum = mk(o[0], o[1], o[2]);
    }
    // hit being shaded / lit
    V3 P = black, hn = black, fn = black, din = black, ncos = black, acc = black, albedo = black;
    uint32_t hit_tri = 0, hit_mat = 0;
    float shadow_max_t = 0.f, contrib = 0.f;
    bool lit_textured = false;
    Frame frames[FORKS ? kMaxRayDepth : 1];
    int fsp = 0;
    Stats st = {0, 0, 0, 0, 0, 0};
    Cand cand;
    cand.t = kFltMax; cand.u = cand.v = 0.f; cand.k = kMiss;
    bool primed = false;
    // apex hint of the ray in flight (trace.hip.hpp "Pencil bundles"): camera rays leave the camera; their reflections off a
    // plane leave the camera's mirror image.  A hint only: wrong or noisy, it loosens the culling and changes no result.
    V3 mirror_apex = black;
    bool burst_done = false;                 // the trace of this iteration ran as a light burst (wave-uniform)
    unsigned long long burst_lanes = 0ull;
    uint32_t burst_plog_done = 0u, burst_part0 = 0u;   // log2(parts) of that burst; this lane's part for its first light
    uint32_t burst_nl_done = 0u;                       // lights of that burst
#ifdef RTK_DEBUG_PHASES
    unsigned long long ph_first_trace = 0, ph_after_first = 0, ph_wait = 0;
    float ph_tr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ph_kind[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ph_job[3][8] = {{0.f}}, ph_own[1] = {0.f};                 // the helpers' jobs of the last light burst
#endif

    for (;;) {
        // ---------- resolve: run each lane forward until it needs a ray traced (or is done)
        while (state != ST_TRACE && state != ST_DONE) {
            if (state == ST_NEW_SAMPLE) {                                   // render.hpp:35-69
                if (sample == A.sample_end) {
                    const float inv = A.spp_f;
                    if (writer) {
                        const Pixel p = my_pixel();
                        float *o = A.out + A.out_index(local_bucket, p.lx, p.ly, p.px, p.py) * 3;
                        // the last pass divides (render.hpp:72; x / 1.0f == x, bit for bit); earlier passes leave the running sum
                        if (A.spp == 1 || A.sample_end != A.spp) { o[0] = pixel_sum.x; o[1] = pixel_sum.y; o[2] = pixel_sum.z; }
                        else { o[0] = pixel_sum.x / inv; o[1] = pixel_sum.y / inv; o[2] = pixel_sum.z / inv; }
                    }
                    state = ST_DONE;
                    continue;
                }
                const Pixel p = my_pixel();
                rkey = root_key(seed_hash, p.py * A.width + p.px, (uint32_t)sample);
                ray = camera_ray(A, p.px, p.py, rkey);
                mirror_apex = ray.o;
                cull = true; depth = 0; pend = PEND_CHILD_BG; fsp = 0;
                primed = PRIMED;
                state = ST_TRACE;
            } else if (state == ST_SHADE) {                                 // color_hit, render.hpp:133-308
                if (depth == A.max_depth) { ret = background; state = ST_RETURN; continue; }   // :138-139
                const DevMaterial *mat = A.materials + hit_mat;
                const int kind = mat->kind;
                if (kind == RTK_MAT_CONSTANT) {
                    ret = mk(mat->albedo[0], mat->albedo[1], mat->albedo[2]);                  // :302-303
                    state = ST_RETURN;
                } else if (kind == RTK_MAT_REFLECTIVE) {                                        // :239-250
                    const V3 rd = din - ((2.0f * dot(din, hn)) * hn);
                    const V3 ro = P + (A.reflection_bias * rd);
                    // the reflected line leaves the mirror image of the incoming line's apex: as far behind P as that apex was
                    mirror_apex = P - ((length(P - mirror_apex) / length(rd)) * rd);
                    ray = spawn_ray(ro, rd);
                    rkey = child_key(rkey, 0u);
                    cull = false; depth += 1; pend = PEND_CHILD_BG;
                    state = ST_TRACE;
                } else if (FORKS && kind == RTK_MAT_REFRACTIVE) {                               // :252-301 (FORKS = the general kernel)
                    V3 n = normalized(mat->smooth ? hn : fn);
                    const V3 i = normalized(din);
                    float eta_i = 1.0f, eta_r = mat->ior;
                    if (0.0f < dot(i, n)) { const float tmp = eta_i; eta_i = eta_r; eta_r = tmp; n = neg(n); }
                    const float cos_i_n = -dot(i, n);
                    const float sin_i_n = __builtin_sqrtf(1.0f - cos_i_n * cos_i_n);
                    const V3 rd = i - ((2.0f * dot(i, n)) * n);
                    const V3 ro = P + (A.reflection_bias * rd);
                    if (eta_r / eta_i < sin_i_n) {                                              // total internal reflection
                        ray = spawn_ray(ro, rd);
                        rkey = child_key(rkey, 0u);
                        cull = false; depth += 1; pend = PEND_CHILD_BLACK;
                        state = ST_TRACE;
                    } else {
                        const float sin_r = ((sin_i_n * eta_i) / eta_r);
                        const float cos_r = __builtin_sqrtf(1.0f - sin_r * sin_r);
                        const V3 r = (cos_r * neg(n)) + (sin_r * normalized(i + (cos_i_n * n)));
                        const double x = (double)(1.0f + dot(i, n));                           // :300, x^5 in double
                        const float fresnel = (float)(0.5 * (x * x * x * x * x));
                        if (FORKS) {
                            Frame &f = frames[fsp++];
                            f.a[0] = ro.x; f.a[1] = ro.y; f.a[2] = ro.z;
                            f.a[3] = rd.x; f.a[4] = rd.y; f.a[5] = rd.z;
                            f.a[6] = fresnel;
                            f.meta = FR_REFR_A | ((uint32_t)depth << 8);
                            f.key = rkey;
                        }
                        ray = spawn_ray(P + (A.refraction_bias * r), r);
                        rkey = child_key(rkey, 0u);
                        cull = false; depth += 1; pend = PEND_CHILD_BLACK;
                        state = ST_TRACE;
                    }
                } else {                                                                        // diffuse :148-209, texture :211-238
                    lit_textured = FORKS && (kind == RTK_MAT_TEXTURE); // light loop only: no GI rays, no final division
                    if (!lit_textured) albedo = mk(mat->albedo[0], mat->albedo[1], mat->albedo[2]);
                    ncos = mat->smooth ? hn : fn;
                    acc = black;
                    if (FORKS && A.diffuse_rays > 0 && !lit_textured) {
                        Frame &f = frames[fsp++];
                        f.a[0] = P.x; f.a[1] = P.y; f.a[2] = P.z;
                        f.a[3] = din.x; f.a[4] = din.y; f.a[5] = din.z;
                        f.a[6] = hn.x; f.a[7] = hn.y; f.a[8] = hn.z;
                        f.a[9] = 0.f; f.a[10] = 0.f; f.a[11] = 0.f;
                        f.meta = FR_DIFFUSE | ((uint32_t)depth << 8);
                        f.tri = hit_tri;
                        f.key = rkey;
                        state = ST_RETURN;      // the frame handler below issues GI ray 0 (ret = 0 adds nothing)
                        ret = black;
                        f.meta |= 0xFFFF0000u;  // iteration = -1: "no child returned yet"
                    } else {
                        light_k = 0;
                        state = ST_LIGHT;
                    }
                }
            } else if (state == ST_LIGHT) {                                 // light loop, render.hpp:184-208
                if (light_k == A.n_lights) {
                    const float div = A.gi_div_f;
                    ret = lit_textured ? acc : mk(acc.x / div, acc.y / div, acc.z / div);
                    state = ST_RETURN;
                    continue;
                }
                const DevLight *L = A.lights + light_k;
                V3 ld = mk(L->pos[0], L->pos[1], L->pos[2]) - P;
                const float radius = length(ld);
                const float area = 4.0f * PI_F * radius * radius;
                ld = normalized(ld);
                const float d0 = dot(ld, ncos);
                const float cosine = (0.0f < d0) ? d0 : 0.0f;                // std::max(0, dot)
                contrib = (L->intensity / area) * cosine;
                if (0.0f < radius) {                                         // is_occluded's loop guard, :114
                    ray = spawn_ray(P + (A.shadow_bias * ld), ld);
                    shadow_max_t = radius;
                    cull = false; pend = PEND_SHADOW;
                    state = ST_TRACE;
                } else {
                    acc = acc + (contrib * albedo);
                    light_k += 1;
                }
            } else {                                                         // ST_RETURN: hand `ret` to the caller
                if (!FORKS || fsp == 0) {
                    pixel_sum = pixel_sum + ret;                             // render.hpp:66
                    sample += 1;
                    state = ST_NEW_SAMPLE;
                    continue;
                }
                Frame &f = frames[fsp - 1];
                const uint32_t kind = f.meta & 0xFFu;
                const int fdepth = (int)((f.meta >> 8) & 0xFFu);
                if (kind == FR_REFR_A) {                                     // refraction subtree done -> reflection ray
                    f.a[7] = ret.x; f.a[8] = ret.y; f.a[9] = ret.z;
                    f.meta = FR_REFR_B | ((uint32_t)fdepth << 8);
                    ray = spawn_ray(mk(f.a[0], f.a[1], f.a[2]), mk(f.a[3], f.a[4], f.a[5]));
                    rkey = child_key(f.key, 1u);
                    cull = false; depth = fdepth + 1; pend = PEND_CHILD_BLACK;
                    state = ST_TRACE;
                } else if (kind == FR_REFR_B) {                              // :301
                    const float fresnel = f.a[6];
                    const V3 refr = mk(f.a[7], f.a[8], f.a[9]);
                    ret = (fresnel * ret) + ((1.0f - fresnel) * refr);
                    fsp -= 1;
                } else {                                                     // FR_DIFFUSE: GI loop, :151-182
                    int it = (int)(short)(f.meta >> 16);
                    if (it >= 0) { f.a[9] += ret.x; f.a[10] += ret.y; f.a[11] += ret.z; }
                    it += 1;
                    const V3 fP = mk(f.a[0], f.a[1], f.a[2]);
                    const V3 fhn = mk(f.a[6], f.a[7], f.a[8]);
                    if (it < A.diffuse_rays) {
                        f.meta = FR_DIFFUSE | ((uint32_t)fdepth << 8) | ((uint32_t)it << 16);
                        const V3 fd = mk(f.a[3], f.a[4], f.a[5]);
                        const V3 right = normalized(cross(fd, fhn));
                        const V3 up = fhn;
                        const V3 fwd = cross(right, up);
                        const float a_xy = PI_F * urand_key(f.key, 2u + 2u * (uint32_t)it);
                        float s1, c1;
                        det_sincos(a_xy, s1, c1);
                        V3 rv = mk(c1, s1, 0.0f);
                        const float a_xz = PI_F * urand_key(f.key, 3u + 2u * (uint32_t)it) * 2.0f;
                        float s2, c2;
                        det_sincos(a_xz, s2, c2);
                        rv = mk(c2 * rv.x + 0.0f * rv.y + (-s2) * rv.z, 0.0f * rv.x + 1.0f * rv.y + 0.0f * rv.z,
                                s2 * rv.x + 0.0f * rv.y + c2 * rv.z);
                        const V3 org = fP + (A.reflection_bias * fhn);
                        const V3 dir = mk(right.x * rv.x + right.y * rv.y + right.z * rv.z,
                                          up.x * rv.x + up.y * rv.y + up.z * rv.z,
                                          fwd.x * rv.x + fwd.y * rv.y + fwd.z * rv.z);
                        ray = spawn_ray(org, dir);
                        rkey = child_key(f.key, (uint32_t)it);
                        cull = false; depth = fdepth + 1; pend = PEND_CHILD_BLACK;
                        state = ST_TRACE;
                    } else {                                                 // GI done: light this hit
                        P = fP; hn = fhn;
                        acc = mk(f.a[9], f.a[10], f.a[11]);
                        hit_tri = f.tri;
                        const DevShade *sh = A.tree.shade + hit_tri;
                        fn = mk(sh->fn[0], sh->fn[1], sh->fn[2]);
                        hit_mat = sh->material;
                        const DevMaterial *mat = A.materials + hit_mat;
                        albedo = mk(mat->albedo[0], mat->albedo[1], mat->albedo[2]);
                        ncos = mat->smooth ? hn : fn;
                        lit_textured = false;
                        depth = fdepth;
                        fsp -= 1;
                        light_k = 0;
                        state = ST_LIGHT;
                    }
                }
            }
        }

        // ---------- one wave-wide closest-hit query for every lane that has a ray pending
        const bool need = (state == ST_TRACE);
        if (__ballot(need) == 0ull) break;
        // hn/fn/din/ret are written when a hit (or a miss) is consumed below and read by ST_SHADE / ST_RETURN in the very next
        // resolve pass, never across a trace.  Saying so keeps twelve registers out of the traversal loops (the difference
        // between 5 resident waves per SIMD with scratch spills and 5 without).
        hn = black; fn = black; din = black; ret = black;
        hit_tri = 0u; hit_mat = 0u;              // likewise: set when a hit is consumed, read by the ST_SHADE that follows at once
        // Most queries of a frame never enter the tree (background): the root box is tested first, and when no lane passes
        // nothing else is set up -- no parking, no bundles.
        constexpr bool kRootFirst = !STATS && MODE == RTK_TRACE_WAVE;
        // ray3's inv_direction (ray3.hpp:11-14): the same three divisions, made here instead of where the ray is spawned -- the
        // quotients are only read by the box tests of the query, and would otherwise sit in (or be spilled from) three
        // registers from the shading code to this point
        ray.inv = mk((1.0f / ray.d.x), (1.0f / ray.d.y), (1.0f / ray.d.z));
        const bool in_root = kRootFirst ? enters_root(A.tree, ray, need) : need;
        // light burst (ShadowBurstService): every lane about to query a light is at the same light, and more lights follow
        bool burst = false;
        uint32_t burst_k = 0u, burst_plog = 0u, burst_nl = 0u;
        unsigned long long burst_mask = 0ull;
        bool elsewhere = false;                  // this lane's occlusion query is answered by a helper wave (another part)
        uint32_t burst_my_part = 0u;
        if (SLICES > 1 && !STATS && !light && A.has_refractive == 0) {
            const bool sh_lane = need & (pend == PEND_SHADOW);
            burst_mask = __builtin_amdgcn_ballot_w64(sh_lane);
            if (burst_mask != 0ull) {
                burst_k = (uint32_t)__builtin_amdgcn_readlane(light_k, __builtin_ctzll(burst_mask));
                const uint32_t left = (uint32_t)A.n_lights - burst_k;
                burst_nl = left < kBurstMaxJobs ? left : kBurstMaxJobs;                      // lights of this burst
                burst_plog = (burst_nl * 4u <= kBurstMaxJobs) ? 2u : (burst_nl * 2u <= kBurstMaxJobs) ? 1u : 0u;
                burst = (__builtin_amdgcn_ballot_w64(sh_lane & ((uint32_t)light_k != burst_k)) == 0ull) & (1u < (burst_nl << burst_plog));
                if (burst) burst_my_part = burst_part(ray.d, sh_lane, burst_plog);   // (every shadow lane of a burst is at light burst_k)
                elsewhere = burst & sh_lane & (burst_my_part != 0u);
            }
        }
        const bool quiet = kRootFirst && !burst && !(PRIMED && wave_any(primed)) && !wave_any(in_root);
        if (quiet) {
            cand.t = kFltMax; cand.u = cand.v = 0.f; cand.k = kMiss;
            burst_done = false;
        } else {
        // State that is live across the query but not used by it waits in LDS instead of in registers: with it the
        // traversal loops overflow the 96 registers of a 5-waves-per-SIMD build into scratch (measured: a background block
        // spent more time on scratch reloads than on its rays).  [var][lane] layout: conflict-free, 36 LDS operations per query.
        float *const park = &park_lds[park_slot][0][fresh_lane()];
        park[0 * 64] = pixel_sum.x; park[1 * 64] = pixel_sum.y; park[2 * 64] = pixel_sum.z;
        park[3 * 64] = acc.x; park[4 * 64] = acc.y; park[5 * 64] = acc.z;
        park[6 * 64] = albedo.x; park[7 * 64] = albedo.y; park[8 * 64] = albedo.z;
        park[9 * 64] = ncos.x; park[10 * 64] = ncos.y; park[11 * 64] = ncos.z;
        park[12 * 64] = P.x; park[13 * 64] = P.y; park[14 * 64] = P.z;
        park[15 * 64] = mirror_apex.x; park[16 * 64] = mirror_apex.y; park[17 * 64] = mirror_apex.z;
        if (PRIMED && wave_any(primed)) {
            // first iteration of the pass: every pending ray is a camera ray whose hit the first pass already found
            if (need) {
                const Pixel p = my_pixel();
                const float4 pc = A.prim[A.out_index(local_bucket, p.lx, p.ly, p.px, p.py)];
                cand.t = pc.x; cand.u = pc.y; cand.v = pc.z; cand.k = __float_as_uint(pc.w);
            }
            nrays_wave -= (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(need));   // counted by the first pass
            primed = false;
            burst_done = false;
        } else {
#ifdef RTK_DEBUG_PHASES
            const unsigned long long tr0 = __builtin_readcyclecounter();
            if (ph_first_trace == 0) ph_first_trace = tr0;
#endif
            // shadow rays of scenes without transmissive materials only ask "is the closest hit nearer than the light":
            // they may stop at the first hit that says yes (trace(), `exit_t`; A.shadow_exit is set by the host).
            const float exit_t = (A.shadow_exit && pend == PEND_SHADOW) ? shadow_max_t : -1.0f;
            // ray class for the bundle culling: shadow rays by light, everything else by depth (camera rays = 0)
            // (shadow rays end in their light: the lines of a class pass through that point, the "apex" of a pencil bundle)
            uint32_t cls = (uint32_t)depth;
            V3 apex = black;
            if (pend == PEND_CHILD_BG) { cls |= kClsHasApex; apex = mirror_apex; }      // camera rays and chains of reflections
            if (pend == PEND_SHADOW) {
                const DevLight *L = A.lights + light_k;
                cls = kClsHasApex | (0x100u + (uint32_t)light_k + 0x40u * (uint32_t)depth);   // (different depths: different surfaces)
                apex = mk(L->pos[0], L->pos[1], L->pos[2]);
            }
            if (burst) {
                const uint32_t lane = fresh_lane();
                group_sh->ray_o[lane] = make_float4(P.x, P.y, P.z, ncos.x);
                group_sh->ray_d[lane] = make_float4(ncos.y, ncos.z, 0.f, 0.f);
                if (lane == 0u) {
                    group_sh->pass_mask = burst_mask; group_sh->first = burst_k; group_sh->count = burst_plog | ((burst_nl << burst_plog) << 8);
                    group_sh->pad[0] = 1u;                                  // next job to hand out (job 0 is traced right here)
                    // (an opaque constant: folded, it becomes one lane of a three-register tuple {first, count, kind} that is
                    // set up in the kernel's prologue and spilled to scratch until it is needed here)
                    uint32_t extra = GROUP_EXTRA;
                    asm volatile("" : "+v"(extra));
                    group_sh->kind = extra;
                }
                __syncthreads();                                            // B1: the helpers start on their lights
                sx.min_tris = 0xFFFFFFFFu;                                  // (they are busy: the owner's own leaves stay whole)
            }
            cand = trace<MODE, STATS, kStage, SLICES, kRootFirst>(A.tree, lds_nodes, ray, cull, in_root & !elsewhere, st, sx, kAutoMinLanes, exit_t, cls, apex);
#ifdef RTK_DEBUG_PHASES
            const unsigned long long ph_w0 = __builtin_readcyclecounter();
#endif
            if (burst) {
                __syncthreads();                                            // B2: their answers are in LDS
                sx.min_tris = A.slice_min_tris;
            }
#ifdef RTK_DEBUG_PHASES
            ph_wait += __builtin_readcyclecounter() - ph_w0;
            if (burst && SLICES > 1) ph_own[0] = (float)(ph_w0 - tr0);
            for (int i = 0; i < 6; ++i) if (sx.n_trace == (uint32_t)i) {
                ph_tr[i] = (float)(__builtin_readcyclecounter() - tr0);
                ph_kind[i] = burst ? 100.f + (float)burst_plog : (float)__popcll(__builtin_amdgcn_ballot_w64(in_root));
            }
#endif
            burst_done = burst; burst_lanes = burst_mask; burst_plog_done = burst_plog; burst_part0 = burst_my_part;
            burst_nl_done = burst_nl;
#ifdef RTK_DEBUG_PHASES
            sx.c_trace += __builtin_readcyclecounter() - tr0; sx.n_trace += 1u;
            if (ph_after_first == 0) ph_after_first = __builtin_readcyclecounter();
#endif
        }

        float *const unpark = &park_lds[park_slot][0][fresh_lane()];
        pixel_sum = mk(unpark[0 * 64], unpark[1 * 64], unpark[2 * 64]);
        acc = mk(unpark[3 * 64], unpark[4 * 64], unpark[5 * 64]);
        albedo = mk(unpark[6 * 64], unpark[7 * 64], unpark[8 * 64]);
        ncos = mk(unpark[9 * 64], unpark[10 * 64], unpark[11 * 64]);
        P = mk(unpark[12 * 64], unpark[13 * 64], unpark[14 * 64]);
        mirror_apex = mk(unpark[15 * 64], unpark[16 * 64], unpark[17 * 64]);
        }

        // ---------- consume
        uint32_t more_rays = 0u;                 // this lane's rays of this iteration (<= 1 + the lights of a burst)
        if (need) {
            const uint32_t lane = fresh_lane();
            more_rays = 1u;
            const bool hit = cand.k != kMiss;
            if (pend == PEND_SHADOW) {                                       // is_occluded, render.hpp:110-131
                bool clear = !hit | (shadow_max_t < cand.t);
                const bool in_burst = SLICES > 1 && burst_done && ((burst_lanes >> lane) & 1ull) != 0ull;
                const uint32_t my_part = in_burst ? burst_part0 : 0u;
                if (my_part != 0u) clear = burst_res[my_part * 64u + lane].y != 0.0f;     // light k of the other parts: a helper's answer
                bool again = false;
                if (!clear && A.has_refractive) {
                    const uint32_t tri = A.tree.tri_ids[cand.k];
                    const uint32_t m = A.tree.shade[tri].material;
                    if (A.materials[m].kind == RTK_MAT_REFRACTIVE) {         // transmissive: step through, :126-127
                        const V3 hp = ray.o + (cand.t * ray.d);
                        ray.o = hp + (A.shadow_bias * ray.d);
                        shadow_max_t -= cand.t;
                        if (0.0f < shadow_max_t) again = true; else clear = true;
                    }
                }
                if (!again) {
                    if (clear) acc = acc + (contrib * albedo);               // :205
                    light_k += 1;
                    state = ST_LIGHT;
                    if (in_burst) {
                        // the helpers' lights, in light order: the same float sum as the sequential loop
                        const uint32_t plog = burst_plog_done;
                        for (uint32_t s = 1u; s < burst_nl_done; ++s) {
                            for (uint32_t pp = 0u; pp < (1u << plog); ++pp) {          // exactly one part answers for this lane
                                const float4 res = burst_res[((s << plog) + pp) * 64u + lane];
                                const uint32_t fl = __float_as_uint(res.z);
                                if ((fl & 2u) != 0u) {
                                    if (res.y != 0.0f) acc = acc + (res.x * albedo);
                                    more_rays += fl & 1u;
                                }
                            }
                            light_k += 1;
                        }
                    }
                }
            } else if (!hit) {
                ret = (pend == PEND_CHILD_BG) ? background : black;
                state = ST_RETURN;
            } else {
                const Surface s = reconstruct(A.tree, cand);
                P = ray.o + (cand.t * ray.d);                                // kd_tree_simd.hpp:254
                hn = s.hit_normal; fn = s.face_normal; din = ray.d;
                hit_tri = s.tri; hit_mat = s.material;
                if (FORKS && A.tri_uv != nullptr && A.materials[hit_mat].kind == RTK_MAT_TEXTURE)   // texture_material: colour of this hit
                    albedo = sample_texture(A.textures + A.materials[hit_mat].texture, A.tri_uv + hit_tri, cand.u, cand.v, A.tex_pixels);
                state = ST_SHADE;
            }
        }
        // (added up here, in uniform control flow, bit by bit through ballots: four scalar popcounts)
        nrays_wave += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((more_rays & 1u) != 0u)) +
                      2u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((more_rays & 2u) != 0u)) +
                      4u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((more_rays & 4u) != 0u)) +
                      8u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((more_rays & 8u) != 0u));
        static_assert(kBurstMaxJobs + 1u <= 15u, "a lane's rays of one iteration fit four bits");
    }

    const uint32_t lane = fresh_lane();
#ifdef RTK_DEBUG_PHASES
    const Pixel dbg_p = my_pixel();
    const uint32_t lx = dbg_p.lx, ly = dbg_p.ly, px = dbg_p.px, py = dbg_p.py;
    if (valid && writer) {
        const unsigned long long ph_now = __builtin_readcyclecounter();
        const unsigned long long ph_rt1 = __builtin_amdgcn_s_memrealtime();
        const float vals[60] = {(float)(ph_now - ph_begin), (float)sx.c_trace, (float)sx.n_trace, (float)sx.n_steps,
                                (float)sx.n_small, (float)sx.t_small, (float)sx.c_small, (float)sx.n_big, (float)sx.t_big, (float)sx.c_big,
                                (float)(ph_begin - ph_entry), (float)(ph_first_trace - ph_begin), (float)(ph_after_first - ph_first_trace),
                                (float)(ph_now - ph_after_first), (float)sx.tally.chunks, (float)sx.tally.surv, (float)sx.tally.tris,
                                (float)(ph_rt0 & 0xFFFFFFull), (float)(ph_rt1 & 0xFFFFFFull), (float)blockIdx.x,
                                (float)ph_wait, ph_tr[0], ph_tr[1], ph_tr[2], ph_tr[3], ph_tr[4], ph_tr[5],
                                ph_kind[0], ph_kind[1], ph_kind[2], ph_kind[3], ph_kind[4], ph_kind[5], (float)sx.c_bund, (float)sx.c_list, ph_own[0],
                                (float)sx.tally.stg.n, (float)sx.tally.stg.w1, (float)sx.tally.stg.l1, (float)sx.tally.stg.w2, (float)sx.tally.stg.l2,
                                (float)sx.tally.stg.w3, (float)sx.tally.stg.l3, (float)sx.tally.stg.l4,
                                ph_job[1][0], ph_job[1][1], ph_job[1][2], ph_job[1][3], ph_job[1][4], ph_job[1][5], ph_job[1][6], ph_job[1][7],
                                ph_job[2][0], ph_job[2][1], ph_job[2][2], ph_job[2][3], ph_job[2][4], ph_job[2][5], ph_job[2][6], ph_job[2][7]};
        float v = 0.f;
        for (int i = 0; i < 60; ++i) v = (lane == (uint32_t)i) ? vals[i] : v;
        float *o = A.out + A.out_index(local_bucket, lx, ly, px, py) * 3;
        o[0] = v; o[1] = 0.f; o[2] = 0.f;
    }
#endif
    if (A.cost_out != nullptr && lane == 0u && gwave < A.n_units) {        // what this block cost, for the next frame's order
        // (a block that ran alone on one wave of a packed workgroup took about twice as long as it would with helpers)
        const unsigned long long dt = (__builtin_readcyclecounter() - cost_t0) >> (light ? 5 : 4);
        A.cost_out[gwave] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
    }
    {   // the longest pixel block of the frame: what the frame cannot be shorter than.  Only long blocks report, into one of 64
        // words (a single word takes ~88 atomics per microsecond: every block reporting would serialise the frame).
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - real_t0;
        if (lane == 0u && dt >= kCriticalMinTicks) atomicMax(A.counters + kCriticalWord + (gwave % (uint32_t)kRayCounterShards), dt);
    }
    if (SLICES > 1 && !light) group_post_exit(group_sh);
    const uint32_t total = nrays_wave;
    if (STATS && writer) flush_stats(st, 0u, A.counters);
    // one no-return atomic per pixel block, spread over 64 words (a single word saturates near 88 atomics/us)
    if (lane == 0u && writer) atomicAdd(A.counters + 8 + (gwave % (uint32_t)kRayCounterShards), (unsigned long long)total);
}

// ------------------------------------------------------------------------------------------------
// Two-pass frames, first pass: camera rays only.  One workgroup per 8x8 pixel block (owner wave + helpers, as in
// k_render).  Stores each pixel's candidate, writes the background for blocks no ray hits, and files every other
// block under a log-scale estimate of the work its shading will need, so that the second pass can start the most
// expensive blocks first (longest-processing-time-first: the frame no longer ends waiting for a few heavy blocks
// that happened to be dispatched last).
template <bool STATS, int SLICES>
__global__ __launch_bounds__(64 * SLICES, 8) void k_primary(RenderArgs A) {
    __shared__ GroupStorage<(SLICES > 1 ? SLICES : 1)> group_st;
    GroupShared *const group_sh = group_st.get();
    const uint32_t slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((threadIdx.x >> 6) + blockIdx.x) % (uint32_t)SLICES));
    if (slice != 0u) { group_helper_loop<SLICES>(A.tree, group_sh, slice); return; }
    SliceCtx sx = {group_sh, A.slice_min_tris, 0u, true, 0u, group_sh->bundles};

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gwave = blockIdx.x;
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint32_t local_bucket = gwave / bpb, sub = gwave % bpb;
    const uint32_t bucket = rank_bucket((uint32_t)A.rank, local_bucket, (uint32_t)A.world, A.skew_q);
    const uint32_t bx = (bucket % A.tiles_x) * A.bucket, by = (bucket / A.tiles_x) * A.bucket;
    const uint32_t lx = (sub % A.blocks_per_bucket_side) * 8u + (lane & 7u);
    const uint32_t ly = (sub / A.blocks_per_bucket_side) * 8u + (lane >> 3);
    const uint32_t px = bx + lx, py = by + ly;
    const bool valid = (bucket < A.n_buckets) & (lx < A.bucket) & (ly < A.bucket) & (px < A.width) & (py < A.height);
    const Ray ray = camera_ray(A, px, py, root_key(pcg_hash(A.seed), py * A.width + px, 0u));
    Stats st = {0, 0, 0, 0, 0, 0};
    const Cand c = trace<RTK_TRACE_WAVE, STATS, false, SLICES>(A.tree, nullptr, ray, true, valid, st, sx);
    group_post_exit(group_sh);

    const size_t pix = A.out_index(local_bucket, lx, ly, px, py);
    const bool hit = valid & (c.k != kMiss);
    const unsigned long long hit_mask = __builtin_amdgcn_ballot_w64(hit);
    if (hit_mask == 0ull) {                                        // nothing to shade: the block is finished here
        if (valid) {
            float *o = A.out + pix * 3;                             // (0 + background) / 1, render.hpp:68-72 with spp == 1
            o[0] = (0.0f + A.background[0]) / 1.0f; o[1] = (0.0f + A.background[1]) / 1.0f; o[2] = (0.0f + A.background[2]) / 1.0f;
        }
    } else {
        if (valid) A.prim[pix] = make_float4(c.t, c.u, c.v, __uint_as_float(c.k));
        // cost estimate: (trace rounds the block's shading will need) x (work of this block's primary trace)
        int kind = -1;
        if (hit) kind = A.materials[A.tree.shade[A.tree.tri_ids[c.k]].material].kind;
        const bool any_diffuse = wave_any(kind == RTK_MAT_DIFFUSE), any_mirror = wave_any(kind == RTK_MAT_REFLECTIVE),
                   any_glass = wave_any(kind == RTK_MAT_REFRACTIVE);
        const uint32_t lights = (uint32_t)A.n_lights + (uint32_t)A.diffuse_rays * 4u;
        uint32_t rounds = 1u;
        if (any_diffuse) rounds += lights;
        if (any_mirror) rounds += 1u + lights;
        if (any_glass) rounds += 4u * (1u + lights);
        const float cost = (float)rounds * (float)(sx.work + 16u) * (float)__popcll(hit_mask);
        uint32_t bin = (uint32_t)(__builtin_log2f(cost) * 2.0f);
        bin = bin > (uint32_t)(kCostBins - 1) ? (uint32_t)(kCostBins - 1) : bin;
        if (lane == 0u) {
            const uint32_t slot = atomicAdd(A.bin_count + bin, 1u);
            A.bin_list[(size_t)bin * A.tile_cap + slot] = gwave;
        }
    }
    const uint32_t total = wave_sum(valid ? 1u : 0u);
    if (STATS) flush_stats(st, 0u, A.counters);
    if (lane == 0u && total != 0u) atomicAdd(A.counters + 8 + (gwave % (uint32_t)kRayCounterShards), (unsigned long long)total);
}

// Two-pass frames, between the passes: concatenates the cost bins, most expensive first.  One workgroup per bin.
__global__ __launch_bounds__(256) void k_tile_order(RenderArgs A) {
    const uint32_t bin = (uint32_t)(kCostBins - 1) - blockIdx.x;
    uint32_t offset = 0;
    for (uint32_t b = bin + 1u; b < (uint32_t)kCostBins; ++b) offset += A.bin_count[b];
    const uint32_t n = A.bin_count[bin];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) A.tile_order[offset + i] = A.bin_list[(size_t)bin * A.tile_cap + i];
    if (bin == 0u && threadIdx.x == 0u) *A.n_listed = offset + n;
}

// ------------------------------------------------------------------------------------------------
// After the all-gather: [world][buckets_per_rank][bucket*bucket][3] -> frame [h][w][3].
__global__ __launch_bounds__(256) void k_assemble(AssembleArgs A) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)A.width * A.height) return;
    const uint32_t px = (uint32_t)(i % A.width), py = (uint32_t)(i / A.width);
    const uint32_t bucket = (py / A.bucket) * A.tiles_x + (px / A.bucket);
    uint32_t rank, local;
    bucket_owner(bucket, A.tiles_x, A.world, A.skew_q, rank, local);
    const size_t src = (((size_t)rank * A.buckets_per_rank + local) * A.bucket + (py % A.bucket)) * A.bucket + (px % A.bucket);
    A.rgb[i * 3 + 0] = A.gathered[src * 3 + 0];
    A.rgb[i * 3 + 1] = A.gathered[src * 3 + 1];
    A.rgb[i * 3 + 2] = A.gathered[src * 3 + 2];
}

// ------------------------------------------------------------------------------------------------
// The camera rays of one sample of every pixel (render.hpp:35-62), [h][w] row-major: one thread per pixel.
__global__ __launch_bounds__(256) void k_camera_rays(RenderArgs A, int sample, rtk_ray *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)A.width * A.height) return;
    const uint32_t px = (uint32_t)(i % A.width), py = (uint32_t)(i / A.width);
    const Ray r = camera_ray(A, px, py, root_key(pcg_hash(A.seed), py * A.width + px, (uint32_t)sample));
    float *o = reinterpret_cast<float *>(out + i);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

// write_ppm's quantisation (io/image/ppm.hpp:17-19): uint8(255.999 * clamp(c, 0, 1)), the product in double.
__global__ __launch_bounds__(256) void k_to_rgb8(const float *rgb, size_t n, uint8_t *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = rgb[i];
    const float k = (c < 0.0f) ? 0.0f : ((1.0f < c) ? 1.0f : c);           // std::clamp(c, 0.f, 1.f)
    out[i] = (uint8_t)(255.999 * (double)k);
}

}  // namespace dev

// ------------------------------------------------------------------------------------------------ launchers

namespace {

template <int MODE, bool STATS>
hipError_t launch_intersect_m(const dev::IntersectArgs &A, bool lds, size_t lds_bytes, hipStream_t s) {
    const unsigned blocks = (unsigned)((A.n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    // only the per-lane walk reads the node array from LDS: the wave-cooperative one fetches nodes and leaves through the scalar
    // cache (staging the array for it copied 6 KB per 256 rays for nothing)
    constexpr bool kNeedsNodes = (MODE == RTK_TRACE_LANE || MODE == RTK_TRACE_AUTO);
    if (lds && kNeedsNodes) hipLaunchKernelGGL((dev::k_intersect<MODE, STATS, true>), dim3(blocks), dim3(256), lds_bytes, s, A);
    else hipLaunchKernelGGL((dev::k_intersect<MODE, STATS, false>), dim3(blocks), dim3(256), 0, s, A);
    return hipGetLastError();
}

template <int MODE, bool STATS, bool FORKS, int SLICES>
hipError_t launch_render_m(const dev::RenderArgs &A, unsigned blocks, bool lds, size_t lds_bytes, hipStream_t s) {
    const unsigned threads = SLICES > 1 ? 64u * SLICES : 256u;
    constexpr bool kNeedsNodes = (MODE == RTK_TRACE_LANE || MODE == RTK_TRACE_AUTO);
    if (lds && kNeedsNodes) hipLaunchKernelGGL((dev::k_render<MODE, STATS, FORKS, true, SLICES>), dim3(blocks), dim3(threads), lds_bytes, s, A);
    else hipLaunchKernelGGL((dev::k_render<MODE, STATS, FORKS, false, SLICES>), dim3(blocks), dim3(threads), 0, s, A);
    return hipGetLastError();
}

template <int MODE, int SLICES>
hipError_t launch_render_mode(const dev::RenderArgs &A, unsigned blocks, bool stats, bool forks, bool lds, size_t lds_bytes,
                              hipStream_t s) {
    if (stats) return forks ? launch_render_m<MODE, true, true, SLICES>(A, blocks, lds, lds_bytes, s)
                            : launch_render_m<MODE, true, false, SLICES>(A, blocks, lds, lds_bytes, s);
    return forks ? launch_render_m<MODE, false, true, SLICES>(A, blocks, lds, lds_bytes, s)
                 : launch_render_m<MODE, false, false, SLICES>(A, blocks, lds, lds_bytes, s);
}

}  // namespace

hipError_t launch_intersect(const dev::IntersectArgs &A, int mode, bool stats, hipStream_t s) {
    const size_t lds_bytes = (size_t)A.tree.n_nodes * sizeof(DevNode);
    const bool lds = lds_bytes <= kMaxNodeLdsBytes;
    switch (mode) {
        case RTK_TRACE_LANE:
            return stats ? launch_intersect_m<RTK_TRACE_LANE, true>(A, lds, lds_bytes, s)
                         : launch_intersect_m<RTK_TRACE_LANE, false>(A, lds, lds_bytes, s);
        case RTK_TRACE_WAVE:
            return stats ? launch_intersect_m<RTK_TRACE_WAVE, true>(A, lds, lds_bytes, s)
                         : launch_intersect_m<RTK_TRACE_WAVE, false>(A, lds, lds_bytes, s);
        default:
            return stats ? launch_intersect_m<RTK_TRACE_AUTO, true>(A, lds, lds_bytes, s)
                         : launch_intersect_m<RTK_TRACE_AUTO, false>(A, lds, lds_bytes, s);
    }
}

hipError_t launch_render(const dev::RenderArgs &A, int mode, bool stats, bool forks, hipStream_t s, unsigned n_workgroups) {
#ifdef RTK_ONLY_LEAN_G4     // `make asm-lean`: the benchmark's kernel alone, for quick looks at its ISA (not a product build)
    hipLaunchKernelGGL((dev::k_render<RTK_TRACE_WAVE, false, false, false, 4>), dim3(A.n_units), dim3(256), 0, s, A);
    return hipGetLastError();
#else
    const size_t lds_bytes = (size_t)A.tree.n_nodes * sizeof(DevNode);
    const bool lds = lds_bytes <= kMaxNodeLdsBytes;
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint64_t waves = (uint64_t)A.buckets_per_rank * bpb;          // one per 8x8 pixel block
    if (waves == 0) return hipSuccess;
    if (waves > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const unsigned packed = (unsigned)((waves + 3) / 4);                  // 4 pixel blocks per workgroup
    switch (mode) {
        case RTK_TRACE_LANE: return launch_render_mode<RTK_TRACE_LANE, 1>(A, packed, stats, forks, lds, lds_bytes, s);
        case RTK_TRACE_WAVE: return launch_render_mode<RTK_TRACE_WAVE, 1>(A, packed, stats, forks, lds, lds_bytes, s);
        // (with a workgroup list the kernel's workgroups beyond the list return at once; launched all the same they are ~100,000
        // waves of a config-2 frame that only start and stop)
        case RTK_TRACE_GROUP4: return launch_render_mode<RTK_TRACE_WAVE, 4>(A, (n_workgroups != 0u && n_workgroups < waves) ? n_workgroups : (unsigned)waves,
                                                                            stats, forks, lds, lds_bytes, s);
        case RTK_TRACE_GROUP8: return launch_render_mode<RTK_TRACE_WAVE, 8>(A, (unsigned)waves, stats, forks, lds, lds_bytes, s);
        case RTK_TRACE_GROUP16: return launch_render_mode<RTK_TRACE_WAVE, 16>(A, (unsigned)waves, stats, forks, lds, lds_bytes, s);
        default: return launch_render_mode<RTK_TRACE_AUTO, 1>(A, packed, stats, forks, lds, lds_bytes, s);
    }
#endif
}

// A prior for the FIRST frame of a shape, when no frame has reported its blocks' costs yet.  One thread per 8x8 pixel block
// sorts it into one of three classes, and a second small kernel lays the launch list out class by class (no sort: ~10 us
// in front of the frame where k_order_by_cost needs 66):
//   1  some probe ray (the four corner pixels, the centre) enters the tree's root box -- first, a workgroup each
//   0  none does: background -- packed four to a workgroup, last
//   (2 is kept for a class to be started before both)
// Only the launch order depends on the classes, no result.  From the second frame on the measured costs take over.
namespace dev {
__global__ __launch_bounds__(256) void k_block_prior(RenderArgs A, uint8_t *cls, uint32_t *count /* [3], zeroed */) {
    const uint32_t unit = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = unit < A.n_units;
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint32_t local_bucket = (have ? unit : 0u) / bpb, sub = (have ? unit : 0u) % bpb;
    const uint32_t bucket = rank_bucket((uint32_t)A.rank, local_bucket, (uint32_t)A.world, A.skew_q);
    const uint32_t bx = (bucket % A.tiles_x) * A.bucket, by = (bucket / A.tiles_x) * A.bucket;
    const uint32_t x0 = bx + (sub % A.blocks_per_bucket_side) * 8u, y0 = by + (sub / A.blocks_per_bucket_side) * 8u;
    const bool in_frame = have && bucket < A.n_buckets && x0 < A.width && y0 < A.height;
    const uint32_t x1 = x0 + 7u < A.width ? x0 + 7u : A.width - 1u, y1 = y0 + 7u < A.height ? y0 + 7u : A.height - 1u;
    RenderArgs P = A;
    P.spp = 1;                                                                  // pixel centres: this is an estimate
    const float4 *root = reinterpret_cast<const float4 *>(A.tree.nodes);
    const float4 r0 = root[0], r1 = root[1];
    float t_min;
    const Ray centre = camera_ray(P, in_frame ? (x0 + x1) >> 1 : 0u, in_frame ? (y0 + y1) >> 1 : 0u, 0u);
    const bool centre_in = in_frame && slab(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, centre, t_min);
    bool enters = centre_in;
    if (in_frame) {
        const uint32_t px[4] = {x0, x1, x0, x1}, py[4] = {y0, y0, y1, y1};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const Ray r = camera_ray(P, px[i], py[i], 0u);
            enters = enters | slab(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r, t_min);
        }
    }
    // (Knowing what the centre ray HITS -- a mirror whose reflections fan out over the mesh is where config 2's longest blocks
    // are -- would give a better order still: the frame went from 0.37 to 0.29 ms with such blocks first.  But tracing the 32,400
    // centre rays, 64 to a wave, takes 0.24 ms: their bundles are 64 pixels wide and cull nothing.  Measured, not kept.)
    const uint32_t c = enters ? 1u : 0u;
    if (have) cls[unit] = (uint8_t)c;
#pragma unroll
    for (uint32_t q = 0; q < 3u; ++q) {                                        // one atomic per wave and class
        const unsigned long long m = __builtin_amdgcn_ballot_w64(have && c == q);
        if (m != 0ull && (int)__lane_id() == __builtin_ctzll(m)) atomicAdd(count + q, (uint32_t)__popcll(m));
    }
}

// order[] = the blocks class by class (2, 1, 0), wg_list / hdr as k_order_by_cost leaves them (RenderArgs::wg_list)
__global__ __launch_bounds__(256) void k_prior_layout(const uint8_t *cls, const uint32_t *count, uint32_t *cursor /* [3], zeroed */,
                                                      uint32_t *order, uint32_t *wg_list, uint32_t *hdr, uint32_t n, uint32_t pack) {
    const uint32_t unit = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = unit < n;
    const uint32_t c = have ? (uint32_t)cls[unit] : 3u;
    const uint32_t n2 = count[2], n1 = count[1], n0 = count[0];
    uint32_t pos = 0u;
#pragma unroll
    for (uint32_t q = 0; q < 3u; ++q) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(have && c == q);
        if (m == 0ull) continue;
        const int leader = __builtin_ctzll(m);
        uint32_t first = 0u;
        if ((int)__lane_id() == leader) first = atomicAdd(cursor + q, (uint32_t)__popcll(m));
        first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
        if (c == q) pos = first + (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1ull));
    }
    if (have) {
        const uint32_t at = c == 2u ? pos : (c == 1u ? n2 + pos : n2 + n1 + pos);
        order[at] = unit;
        if (c != 0u) wg_list[at] = unit;
        else if (pos % pack == 0u) wg_list[n2 + n1 + pos / pack] = 0x80000000u | at;
    }
    if (unit == 0u) { hdr[0] = n2 + n1 + (n0 + pack - 1u) / pack; hdr[1] = n; }
}
}  // namespace dev

hipError_t launch_block_prior(const dev::RenderArgs &A, uint8_t *cls, uint32_t *order, uint32_t *wg_list, uint32_t *hdr,
                              uint32_t *scratch /* [6], zeroed by the caller */, uint32_t pack, hipStream_t s) {
    if (A.n_units == 0u) return hipSuccess;
    // (`scratch` comes zeroed: it lies behind the frame's counters and is cleared with them -- a fill of its own, two in fact for 24
    // bytes, was 15 us in front of the first frame)
    const unsigned blocks = (A.n_units + 255u) / 256u;
    hipLaunchKernelGGL(dev::k_block_prior, dim3(blocks), dim3(256), 0, s, A, cls, scratch);
    hipLaunchKernelGGL(dev::k_prior_layout, dim3(blocks), dim3(256), 0, s, cls, scratch, scratch + 3, order, wg_list, hdr, A.n_units, pack);
    return hipGetLastError();
}

// Counting sort of the pixel blocks by last frame's cost, most expensive first.  One workgroup: a frame has tens of
// thousands of blocks, the whole job is two passes over a few hundred KB.
namespace dev {
__device__ __forceinline__ uint32_t cost_bin(uint32_t c) {                 // 8 bins per octave
    c |= 1u;
    const uint32_t e = 31u - (uint32_t)__clz((int)c);
    const uint32_t m = e >= 3u ? (c >> (e - 3u)) & 7u : (c << (3u - e)) & 7u;
    return e * 8u + m;                                                     // 0..255
}
__global__ __launch_bounds__(1024) void k_order_by_cost(const uint32_t *cost, uint8_t *bins, uint32_t *order, uint32_t *wg_list,
                                                        uint32_t *hdr, uint32_t n, uint32_t light_below, uint32_t floor_below,
                                                        uint32_t pack) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t start[256];
    __shared__ uint32_t wg_hist[2][256], wg_base[2][256];
    __shared__ uint32_t n_single_sh;
    const uint32_t light_bin = light_below > 0u ? cost_bin(light_below) : 0u;  // blocks in bins below it are packed `pack` to a workgroup
    // ... and of those, the ones below `floor_bin` need no order among themselves (background, a handful of nodes)
    const uint32_t floor_raw = floor_below > 0u ? cost_bin(floor_below) : 0u;
    const uint32_t floor_bin = floor_raw < light_bin ? floor_raw : light_bin;
    constexpr uint32_t kBatch = 8;                                          // loads in flight per thread: the kernel is one workgroup,
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) { hist[i] = 0u; wg_hist[0][i] = 0u; wg_hist[1][i] = 0u; }   // so memory latency, not bandwidth, is what it waits for
    __syncthreads();
    // the bin of every block is computed once and kept: `order` is a permutation even if cost[] changes under us
    for (uint32_t base = 0; base < n; base += blockDim.x * kBatch) {
        uint32_t c[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * blockDim.x + threadIdx.x;
            c[k] = i < n ? cost[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * blockDim.x + threadIdx.x;
            // most blocks of a frame are cheap and need no order among themselves: they all go to bin 0, counted once per wave
            uint32_t b = i < n ? cost_bin(c[k]) : 0u;
            if (b < floor_bin) b = 0u;
            if (i < n) bins[i] = (uint8_t)b;
            const unsigned long long cheap = __builtin_amdgcn_ballot_w64(i < n && b == 0u);
            if (i < n && b != 0u) atomicAdd(&hist[b], 1u);
            if (cheap != 0ull && (int)__lane_id() == __builtin_ctzll(cheap)) atomicAdd(&hist[0], (uint32_t)__popcll(cheap));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t acc = 0u;
        uint32_t singles = 0u;
        for (int b = 255; b >= 0; --b) {
            start[b] = acc; acc += hist[b];
            if (light_bin == 0u || (uint32_t)b >= light_bin) singles = acc;
        }
        n_single_sh = light_bin > 0u ? singles : n;                          // order[0, n_single): one workgroup each; the rest packed
        hdr[1] = n;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += blockDim.x * kBatch) {
        uint32_t b[kBatch], pos[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * blockDim.x + threadIdx.x;
            b[k] = i < n ? (uint32_t)bins[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * blockDim.x + threadIdx.x;
            const unsigned long long cheap = __builtin_amdgcn_ballot_w64(i < n && b[k] == 0u);
            pos[k] = 0u;
            if (i < n && b[k] != 0u) pos[k] = atomicAdd(&start[b[k]], 1u);
            if (cheap != 0ull) {
                const int leader = __builtin_ctzll(cheap);
                uint32_t first = 0u;
                if ((int)__lane_id() == leader) first = atomicAdd(&start[0], (uint32_t)__popcll(cheap));
                first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                if (i < n && b[k] == 0u) pos[k] = first + (uint32_t)__popcll(cheap & ((1ull << __lane_id()) - 1ull));
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t i = base + k * blockDim.x + threadIdx.x;
            if (i < n) order[pos[k]] = i;
        }
    }
    // The launch order of the WORKGROUPS: by how long each will run.  A packed workgroup runs its blocks one wave each, with
    // no helpers: it takes about twice as long as its most expensive block would take alone (8 bins = one octave), and
    // has to start as early as the single-block workgroups of that duration -- started behind all of them it is the
    // frame's tail.  wg_list[w] = block id, or 0x80000000 | index into order[] of the first of `pack` blocks.
    __threadfence_block();
    __syncthreads();
    const uint32_t n_single = n_single_sh;
    const uint32_t n_packed = (n - n_single + pack - 1u) / pack;
    const uint32_t n_wgs = n_single + n_packed;
    for (uint32_t w = threadIdx.x; w < n_wgs; w += blockDim.x) {
        const bool packed = w >= n_single;
        const uint32_t at = packed ? n_single + (w - n_single) * pack : w;
        uint32_t key = (uint32_t)bins[order[at]];
        if (packed && key != 0u) key = key + 8u < 255u ? key + 8u : 255u;
        atomicAdd(&wg_hist[packed ? 1 : 0][key], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t acc = 0u;
        for (int b = 255; b >= 0; --b) {
            wg_base[0][b] = acc; acc += wg_hist[0][b];
            wg_base[1][b] = acc; acc += wg_hist[1][b];
        }
        hdr[0] = n_wgs;
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < n_wgs; w += blockDim.x) {
        const bool packed = w >= n_single;
        const uint32_t at = packed ? n_single + (w - n_single) * pack : w;
        const uint32_t blk = order[at];
        uint32_t key = (uint32_t)bins[blk];
        if (packed && key != 0u) key = key + 8u < 255u ? key + 8u : 255u;
        const uint32_t pos = atomicAdd(&wg_base[packed ? 1 : 0][key], 1u);
        wg_list[pos] = packed ? (0x80000000u | at) : blk;
    }
}
}  // namespace dev

hipError_t launch_order_by_cost(const uint32_t *cost, uint8_t *bins, uint32_t *order, uint32_t *wg_list, uint32_t *hdr, uint32_t n,
                                uint32_t light_below, uint32_t floor_below, uint32_t pack, hipStream_t s) {
    if (n == 0u) return hipSuccess;
    if (pack == 0u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dev::k_order_by_cost, dim3(1), dim3(1024), 0, s, cost, bins, order, wg_list, hdr, n, light_below, floor_below, pack);
    return hipGetLastError();
}

// Two-pass frame (spp == 1): k_primary -> k_tile_order -> primed k_render, all on one stream.
hipError_t launch_twopass(const dev::RenderArgs &A, bool stats, bool forks, hipStream_t s) {
    const uint32_t bpb = A.blocks_per_bucket_side * A.blocks_per_bucket_side;
    const uint64_t tiles = (uint64_t)A.buckets_per_rank * bpb;
    if (tiles == 0) return hipSuccess;
    if (tiles > A.tile_cap) return hipErrorInvalidValue;
#ifdef RTK_ONLY_LEAN_G4
    return hipErrorNotSupported;
#else
    hipError_t e = hipMemsetAsync(A.bin_count, 0, (kCostBins + 1) * sizeof(uint32_t), s);      // + n_listed
    if (e != hipSuccess) return e;
    if (stats) hipLaunchKernelGGL((dev::k_primary<true, 4>), dim3((unsigned)tiles), dim3(256), 0, s, A);
    else hipLaunchKernelGGL((dev::k_primary<false, 4>), dim3((unsigned)tiles), dim3(256), 0, s, A);
    hipLaunchKernelGGL(dev::k_tile_order, dim3(kCostBins), dim3(256), 0, s, A);
    if (stats) {
        if (forks) hipLaunchKernelGGL((dev::k_render<RTK_TRACE_WAVE, true, true, false, 4, true>), dim3((unsigned)tiles), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((dev::k_render<RTK_TRACE_WAVE, true, false, false, 4, true>), dim3((unsigned)tiles), dim3(256), 0, s, A);
    } else {
        if (forks) hipLaunchKernelGGL((dev::k_render<RTK_TRACE_WAVE, false, true, false, 4, true>), dim3((unsigned)tiles), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((dev::k_render<RTK_TRACE_WAVE, false, false, false, 4, true>), dim3((unsigned)tiles), dim3(256), 0, s, A);
    }
    return hipGetLastError();
#endif
}

hipError_t launch_camera_rays(const dev::RenderArgs &A, int sample, rtk_ray *d_rays, hipStream_t s) {
    const size_t n = (size_t)A.width * A.height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dev::k_camera_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, sample, d_rays);
    return hipGetLastError();
}

hipError_t launch_to_rgb8(const float *d_rgb, size_t n, uint8_t *d_out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dev::k_to_rgb8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_rgb, n, d_out);
    return hipGetLastError();
}

hipError_t launch_assemble(const dev::AssembleArgs &A, hipStream_t s) {
    const size_t n = (size_t)A.width * A.height;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(dev::k_assemble, dim3(blocks), dim3(256), 0, s, A);
    return hipGetLastError();
}

}  // namespace rtk
