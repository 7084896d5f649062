// Device-side traversal and intersection for gfx950 (wave64).  Compile with -ffp-contract=off:
// every expression keeps the reference's operation order so IEEE add/mul/div/sqrt reproduce the
// CPU results bit for bit.  Reference paths are relative to
// /root/reference/include/raytracer/.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#include "rtk_internal.hpp"

namespace rtk {
namespace dev {

constexpr float kFltMax = 3.402823466e+38f;
constexpr uint32_t kMiss = 0xFFFFFFFFu;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }   // vec3.hpp:77-79
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }   // vec3.hpp:81-83
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }      // vec3.hpp:94-97
__device__ __forceinline__ V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  // vec3.hpp:119-122
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                     // vec3.hpp:124-131
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); } // vec3.hpp:85-91
__device__ __forceinline__ V3 normalized(V3 a) {                                                      // vec3.hpp:104-108
    const float inv_length = (1.0f / length(a));
    return V3{a.x * inv_length, a.y * inv_length, a.z * inv_length};
}

struct Ray {            // ray3<F>, ray3.hpp:5-15
    V3 o, d, inv;
};
__device__ __forceinline__ Ray make_ray(V3 o, V3 d) {
    Ray r;
    r.o = o; r.d = d;
    r.inv = V3{(1.0f / d.x), (1.0f / d.y), (1.0f / d.z)};                     // vec3.hpp:99-102
    return r;
}

struct Cand {           // kd_tree_simd.hpp:86-93 hit_candidate, with the leaf-ref index instead of (pack, lane)
    float t, u, v;
    uint32_t k;         // index into the leaf-ref arrays; kMiss = no hit
};

struct Stats {          // per-lane work counters (only in STATS kernels)
    uint32_t nodes, boxpass, leaves, tris, packets16, hits;
};

struct TreeView {
    const DevNode *nodes;
    const DevTri *tris;
    const uint32_t *tri_ids;
    const DevShade *shade;
    const DevNode *leaves;       // the leaves of `nodes` alone, in traversal order (a = first leaf ref, b = count)
    const DevNode *leaves_fast;  // RTK_TRAVERSAL_FAST: [8][n_leaves] the leaves front to back per direction octant; null = reference order
    uint32_t n_leaves;
    uint32_t n_nodes;
    float eps;
    int normalize;
    int bundle_cull;             // 1: leaves may be pre-culled against the wave's ray bundle (all vertex coordinates are small enough)
    int scalar_surv;             // 1: a culling pass's survivors are re-read through the scalar cache instead of v_readlane (leaf_range_bundle)
};

// Wave-uniform loads: address space 4 (constant) forces s_load_* through the scalar cache, so one
// fetch of a node / triangle serves all 64 rays of the wave.
typedef const uint32_t __attribute__((address_space(4))) *cptr_u32;
typedef const float __attribute__((address_space(4))) *cptr_f32;

// aabb3::intersect(ray), aabb3.hpp:74-90.  Explicit compare+select in the reference's order so that
// NaNs (0*inf when a direction component is 0) take the same path as std::minmax/max/min; the
// per-axis early return is an OR of the three "t_max < t_min" tests (once true it stays a miss).
__device__ __forceinline__ bool slab(const float lo0, const float lo1, const float lo2, const float hi0, const float hi1,
                                     const float hi2, const Ray &r, float &t_min_out) {
    float t_min = 0.0f, t_max = kFltMax;
    bool miss;
    {
        const float a = (lo0 - r.o.x) * r.inv.x, c = (hi0 - r.o.x) * r.inv.x;
        const float t1 = (c < a) ? c : a, t2 = (c < a) ? a : c;
        t_min = (t_min < t1) ? t1 : t_min;
        t_max = (t2 < t_max) ? t2 : t_max;
        miss = t_max < t_min;
    }
    {
        const float a = (lo1 - r.o.y) * r.inv.y, c = (hi1 - r.o.y) * r.inv.y;
        const float t1 = (c < a) ? c : a, t2 = (c < a) ? a : c;
        t_min = (t_min < t1) ? t1 : t_min;
        t_max = (t2 < t_max) ? t2 : t_max;
        miss = miss | (t_max < t_min);
    }
    {
        const float a = (lo2 - r.o.z) * r.inv.z, c = (hi2 - r.o.z) * r.inv.z;
        const float t1 = (c < a) ? c : a, t2 = (c < a) ? a : c;
        t_min = (t_min < t1) ? t1 : t_min;
        t_max = (t2 < t_max) ? t2 : t_max;
        miss = miss | (t_max < t_min);
    }
    t_min_out = t_min;
    return !miss;
}

// One lane of triangle_packet::intersect<cull,eps> (kd_tree_simd.hpp:25-60) followed by the winner rule of
// intersect_leaf (:266-302) and intersect (:222-226): within a leaf the earliest triangle with the
// smallest t wins, across leaves only a strictly smaller t replaces — i.e. a running strict '<'.
__device__ __forceinline__ void test_triangle(const float v0x, const float v0y, const float v0z, const float e1x,
                                              const float e1y, const float e1z, const float e2x, const float e2y,
                                              const float e2z, const Ray &r, const bool cull, const float eps,
                                              const uint32_t k, const bool enabled, Cand &best) {
    const float pvx = r.d.y * e2z - r.d.z * e2y;
    const float pvy = r.d.z * e2x - r.d.x * e2z;
    const float pvz = r.d.x * e2y - r.d.y * e2x;
    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
    bool m = enabled & (eps <= (cull ? det : __builtin_fabsf(det)));
    const float inv_det = (1.0f / det);
    const float tvx = r.o.x - v0x, tvy = r.o.y - v0y, tvz = r.o.z - v0z;
    const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv_det;
    m = m & (0.0f <= u) & (u <= 1.0f);
    const float qx = tvy * e1z - tvz * e1y;
    const float qy = tvz * e1x - tvx * e1z;
    const float qz = tvx * e1y - tvy * e1x;
    const float v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv_det;
    m = m & (0.0f <= v) & (u + v <= 1.0f);
    const float t = (e2x * qx + e2y * qy + e2z * qz) * inv_det;
    m = m & (eps < t) & (t < best.t);
    if (m) { best.t = t; best.u = u; best.v = v; best.k = k; }
}

// ------------------------------------------------------------------------------------------------
// Per-lane traversal: every lane walks the tree on its own.  The traversal order of the reference is
// ray-independent (child1 then child0, no near/far sort), so the LIFO stack of kd_tree_simd.hpp:191-214
// collapses to one index into the traversal-ordered node array: pass+inner -> n+1, otherwise -> skip.
// `n` and `best` are the complete traversal state, which is what lets the wave path hand over mid-tree.
template <bool STATS, bool LDS_NODES>
__device__ __forceinline__ void trace_lane_from(const TreeView &T, const DevNode *lds_nodes, const Ray &r, const bool cull,
                                                uint32_t n, Cand &best, Stats &st, const float exit_t = -1.0f) {
    const uint32_t end = T.n_nodes;
    for (;;) {
        uint32_t leaf_first = 0, leaf_count = 0;
        while (n < end) {
            const DevNode *np = LDS_NODES ? (lds_nodes + n) : (T.nodes + n);
            const float4 q0 = *reinterpret_cast<const float4 *>(np);
            const float4 q1 = *(reinterpret_cast<const float4 *>(np) + 1);
            const uint32_t a = __float_as_uint(q1.z), b = __float_as_uint(q1.w);
            float t_min;
            const bool box = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, t_min);
            const bool pass = box & !(best.t < t_min);                     // kd_tree_simd.hpp:202-205
            if (STATS) { st.nodes += 1; st.boxpass += pass ? 1u : 0u; }
            const bool inner = (b == DEV_INNER);
            if (pass & !inner) { leaf_first = a; leaf_count = b; n += 1; break; }
            n = (pass | !inner) ? n + 1 : a;
        }
        if (leaf_count == 0) break;
        if (STATS) { st.leaves += 1; st.tris += leaf_count; st.packets16 += (leaf_count + 15u) >> 4; }
        const float *tp = reinterpret_cast<const float *>(T.tris + leaf_first);
        for (uint32_t k = 0; k < leaf_count; ++k, tp += 9) {
            test_triangle(tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6], tp[7], tp[8], r, cull, T.eps, leaf_first + k,
                          true, best);
        }
        if (best.t <= exit_t) break;                                       // occlusion query answered (see trace())
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative traversal: the wave walks the traversal-ordered node array ONCE for its 64 rays.
// `n` is wave-uniform; each lane keeps `next`, the node it wants to visit next (>= n always).  A lane
// takes part in node n iff next == n.  Because skip targets nest, "some lane descends ? n+1 : skip" never
// jumps past a node another lane is waiting for.  Nodes and triangles are fetched with scalar loads.
// Lanes see exactly the node/triangle sequence they would see alone, so results are identical.
// Returns the lanes' `next` so the caller can continue per-lane (hand-over) if it left early.
__device__ __forceinline__ bool wave_any(const bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

struct TriS {      // one leaf reference held in SGPRs (wave-uniform)
    float v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z;
};
// One 32-byte scalar load (v0, e1, e2x, e2y) + one dword (e2z).  As ONE load the first eight floats cannot be split by
// the compiler into "needed now" and "needed later" halves: split, the later half was sunk to the end of the loop
// body, right in front of the wait, which exposed a full scalar-cache round trip in every iteration.
struct NodeS {     // one node held in SGPRs
    float lo0, lo1, lo2, hi0, hi1, hi2;
    uint32_t a, b;
};
typedef uint32_t uint8x_t __attribute__((ext_vector_type(8)));
typedef const uint8x_t __attribute__((address_space(4))) *cptr_u32x8;
__device__ __forceinline__ NodeS load_node_uniform(cptr_u32 np) {
    const uint8x_t q = *(cptr_u32x8)(const void __attribute__((address_space(4))) *)np;
    NodeS n;
    n.lo0 = __uint_as_float(q[0]); n.lo1 = __uint_as_float(q[1]); n.lo2 = __uint_as_float(q[2]);
    n.hi0 = __uint_as_float(q[3]); n.hi1 = __uint_as_float(q[4]); n.hi2 = __uint_as_float(q[5]);
    n.a = q[6]; n.b = q[7];
    return n;
}
typedef float float8_t __attribute__((ext_vector_type(8)));
typedef const float8_t __attribute__((address_space(4))) *cptr_f32x8;
__device__ __forceinline__ TriS load_tri_uniform(cptr_f32 tp) {
    const float8_t a = *(cptr_f32x8)(const void __attribute__((address_space(4))) *)tp;
    TriS t;
    t.v0x = a[0]; t.v0y = a[1]; t.v0z = a[2];
    t.e1x = a[3]; t.e1y = a[4]; t.e1z = a[5];
    t.e2x = a[6]; t.e2y = a[7]; t.e2z = tp[8];
    return t;
}

// one triangle against the wave's rays (the arithmetic of test_triangle, wave-level early outs between the stages)
#ifdef RTK_DEBUG_PHASES
struct StageTally { uint32_t n, w1, l1, w2, l2, w3, l3, l4; };   // triangles; waves / lanes alive after det+u_est, after u, after v; lanes accepted
#define RTK_STAGE_ARG , StageTally *stg = nullptr
#define RTK_STAGE(x) if (stg) { x; }
#else
#define RTK_STAGE_ARG
#define RTK_STAGE(x)
#endif
__device__ __forceinline__ void tri_step(const TriS &cur, const uint32_t k, const Ray &r, const bool cull, const float eps,
                                         const unsigned long long pass_mask, const uint32_t lane, Cand &best RTK_STAGE_ARG) {
    RTK_STAGE(stg->n += 1u)
    const float pvx = r.d.y * cur.e2z - r.d.z * cur.e2y;
    const float pvy = r.d.z * cur.e2x - r.d.x * cur.e2z;
    const float pvz = r.d.x * cur.e2y - r.d.y * cur.e2x;
    const float det = cur.e1x * pvx + cur.e1y * pvy + cur.e1z * pvz;
    unsigned long long m = pass_mask & __builtin_amdgcn_ballot_w64(eps <= (cull ? det : __builtin_fabsf(det)));
    if (m == 0ull) return;
    const float tvx = r.o.x - cur.v0x, tvy = r.o.y - cur.v0y, tvz = r.o.z - cur.v0z;
    const float un = tvx * pvx + tvy * pvy + tvz * pvz;
    // Most triangles of a leaf are missed by every ray at the `u` test, and the IEEE division that test needs is a
    // third of the work up to there.  A one-instruction reciprocal estimate (|error| <= 1 ulp) decides the clear
    // cases first: un*rcp(det) beyond [-1e-30, 1.00001] means the exactly rounded u = un * (1/det) is beyond [0, 1]
    // as well (the estimate and the exact product differ by < 2^-21 relative; NaNs and a flushed estimate fail both
    // comparisons and stay in).  Only if some lane is NOT clearly out does the wave run the exact arithmetic.
    const float u_est = un * __builtin_amdgcn_rcpf(det);
    m &= ~(__builtin_amdgcn_ballot_w64(u_est < -1.0e-30f) | __builtin_amdgcn_ballot_w64(1.00001f < u_est));
    if (m == 0ull) return;
    RTK_STAGE(stg->w1 += 1u; stg->l1 += (uint32_t)__popcll(m))
    const float inv_det = (1.0f / det);
    const float u = un * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= u) & __builtin_amdgcn_ballot_w64(u <= 1.0f);
    if (m == 0ull) return;
    RTK_STAGE(stg->w2 += 1u; stg->l2 += (uint32_t)__popcll(m))
    const float qx = tvy * cur.e1z - tvz * cur.e1y;
    const float qy = tvz * cur.e1x - tvx * cur.e1z;
    const float qz = tvx * cur.e1y - tvy * cur.e1x;
    const float v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= v) & __builtin_amdgcn_ballot_w64(u + v <= 1.0f);
    if (m == 0ull) return;
    RTK_STAGE(stg->w3 += 1u; stg->l3 += (uint32_t)__popcll(m))
    const float t = (cur.e2x * qx + cur.e2y * qy + cur.e2z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(eps < t) & __builtin_amdgcn_ballot_w64(t < best.t);
    if (m == 0ull) return;
    RTK_STAGE(stg->l4 += (uint32_t)__popcll(m))
    if ((m >> lane) & 1ull) { best.t = t; best.u = u; best.v = v; best.k = k; }
}

// Tests leaf references [lo, hi) of the leaf starting at `first` against the wave's rays (lanes with `pass`).
// Software-pipelined: the scalar loads of the next triangle are issued before the current one is tested, so scalar-cache /
// L2 latency overlaps the arithmetic.
__device__ __forceinline__ void leaf_range_wave(cptr_f32 tris, const uint32_t first, const uint32_t lo, const uint32_t hi,
                                                const Ray &r, const bool cull, const float eps, const bool pass, Cand &best) {
    if (lo >= hi) return;
    // The per-lane predicate of the reference's mask (`mask &= ...`, kd_tree_simd.hpp:33-57) is carried as a 64-bit
    // wave mask in SGPRs: every comparison is one v_cmp writing a lane mask, the ANDs and the "is anybody left"
    // early-outs run on the scalar unit.  Two triangles per turn, A and B: each is fetched into its own registers while
    // the other is tested, so nothing has to be moved from a "next" set into a "current" one.
    const unsigned long long pass_mask = __builtin_amdgcn_ballot_w64(pass);
    const uint32_t lane = __lane_id();
    cptr_f32 base = tris + (size_t)first * 9;
    const uint32_t last = hi - 1u;                                         // a prefetch past the range re-reads its last triangle
    uint32_t k = lo;
    TriS A = load_tri_uniform(base + (size_t)k * 9);
    for (;;) {
        const TriS B = load_tri_uniform(base + (size_t)(k + 1u < hi ? k + 1u : last) * 9);
        tri_step(A, first + k, r, cull, eps, pass_mask, lane, best);
        if (k + 1u >= hi) break;
        A = load_tri_uniform(base + (size_t)(k + 2u < hi ? k + 2u : last) * 9);
        tri_step(B, first + k + 1u, r, cull, eps, pass_mask, lane, best);
        if (k + 2u >= hi) break;
        k += 2u;
    }
}

// ------------------------------------------------------------------------------------------------
// Bundle culling.  The rays a wave traces together come in a few coherent classes (the camera rays of an 8x8 pixel block,
// the shadow rays of a surface patch towards one light, the reflections off a plane): of the hundreds of triangles in a
// leaf only a handful can be hit by ANY ray of a class.  So a leaf is first tested 64 triangles at a time, ONE TRIANGLE
// PER LANE, against each class's BUNDLE as a whole: the arithmetic of triangle_packet::intersect (kd_tree_simd.hpp:25-60)
// is evaluated in interval arithmetic over the box of the bundle's origins and directions, and a triangle is dropped when
// one of the tests of :33-57 fails for the whole interval.  Only the survivors go through the exact per-ray test
// (tri_step), in leaf order, their data broadcast from the lane that holds them (v_readlane, no memory access).
//
// Exactness.  The interval evaluation follows tri_step's expression tree operation by operation.  IEEE rounding is
// monotone, so for every float operation z = fl(x op y) of tri_step with x in [xl, xh], y in [yl, yh], z lies between the
// smallest and the largest of the op's values at the interval corners, computed in the same float arithmetic.  By induction
// every float the exact test computes for any ray of the bundle lies inside the interval computed here -- no error analysis,
// no epsilon.  (1/det is bounded through v_rcp_f32, |error| <= 1 ulp, widened by 8 ulp.)  A dropped triangle would have
// failed the exact test for every ray, so it could never have updated a candidate: results are bit-identical.
// Requirements, checked once per bundle / per accel: all operands finite and below 1e9 in magnitude, so that no
// intermediate overflows (inf - inf and 0 * inf would produce NaNs that v_min/v_max silently drop).
//
// A trace's bundles (at most kMaxBundles, one per ray class the caller names; further classes are merged into the last)
// live in LDS, kBundleFloats floats each: [0..11] the boxes below, [12..17] bounds of fl(1/d) per axis (lx,hx,ly,hy,lz,hz),
// [18] flags, [19] delta * |d|_1, [20..22] apex, [24..38] centre/radius form of the boxes for pencil bundles.  A culling pass
// reads them back with the same address in every lane and moves them to SGPRs (v_readfirstlane): neither register budget
// of the render kernels has room for them across a trace, and as VGPRs they crowd out the culling's own temporaries.
struct Bundle {
    float olx, oly, olz, ohx, ohy, ohz;   // box of the ray origins
    float dlx, dly, dlz, dhx, dhy, dhz;   // box of the ray directions
    uint32_t flags;
};
enum : uint32_t { BUNDLE_OK = 1u, BUNDLE_ALL_CULL = 2u, BUNDLE_INVX = 4u, BUNDLE_INVY = 8u, BUNDLE_INVZ = 16u, BUNDLE_PENCIL = 32u };
constexpr float kBundleLimit = 1.0e9f;   // |coordinate| bound under which the interval arithmetic cannot overflow
constexpr uint32_t kBundleMinTris = 4;   // leaves (or slices) smaller than this are tested triangle by triangle
#ifndef RTK_SPLIT_LEVELS
#define RTK_SPLIT_LEVELS 1
#endif
constexpr int kMaxBundles = 4;
constexpr int kBundleSplitLevels = RTK_SPLIT_LEVELS;   // 1: a wide class is cut in two; 2: and its halves once more (measured:
                                                       // config 2 unchanged, the diffuse-ray frames of configs 4 / 5 4-6 % slower)
constexpr float kBundleSplitRadius = 0.03f;   // a pencil whose direction box is wider than this (per axis, half width) is cut in two
constexpr int kBundleFloats = 40;       // see make_bundles for the layout
constexpr uint32_t kClsHasApex = 0x80000000u;   // bit of the caller's ray class: the `apex` passed along is meaningful

struct BundleSet {       // wave-uniform handle of the current trace's bundles
    const float *lds;    // [kMaxBundles][kBundleFloats]
    uint32_t n;          // 0: no culling for this trace
};

// Wave-wide min / max through DPP (quad_perm, row_shr within rows of 16, then row_bcast across rows; the classic gfx9
// reduction): the full result ends up in lane 63.  All 64 lanes must be active.  Written as v_min/v_max with the DPP move
// folded in (from the builtins the compiler makes v_mov + s_nop + v_mov_dpp + canonicalise + v_max: five slots per step).
template <bool MAX>
__device__ __forceinline__ float wave_minmax(float v) {
    // one chain: every step reads what the previous one wrote, two wait states apart (s_nop 1)
    if (MAX)
        asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" : "+v"(v));
    else
        asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Six minima and six maxima at once, each instruction a v_min/v_max with the DPP move folded in (the compiler emits
// v_mov_b32_dpp + s_nop + v_min for every step of wave_minmax: three issue slots where one will do).  The twelve chains are
// interleaved, so a register written by one instruction is read eleven instructions later: no DPP hazard inside the block;
// the s_nop in front covers a VALU write of an input just before it.  Lanes without a DPP source keep their value
// (bound_ctrl off), like min(v, v).  Results in lane 63.
#define RTK_DPP6(OP, CTRL) \
    OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" \
    OP " %3, %3, %3 " CTRL "\n\t" OP " %4, %4, %4 " CTRL "\n\t" OP " %5, %5, %5 " CTRL "\n\t"
#define RTK_DPP6B(OP, CTRL) \
    OP " %6, %6, %6 " CTRL "\n\t" OP " %7, %7, %7 " CTRL "\n\t" OP " %8, %8, %8 " CTRL "\n\t" \
    OP " %9, %9, %9 " CTRL "\n\t" OP " %10, %10, %10 " CTRL "\n\t" OP " %11, %11, %11 " CTRL "\n\t"
#define RTK_DPP_STEP12(CTRL) RTK_DPP6("v_min_f32_dpp", CTRL) RTK_DPP6B("v_max_f32_dpp", CTRL)
#define RTK_DPP_STEP6(CTRL) \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n\t" "v_min_f32_dpp %1, %1, %1 " CTRL "\n\t" "v_min_f32_dpp %2, %2, %2 " CTRL "\n\t" \
    "v_max_f32_dpp %3, %3, %3 " CTRL "\n\t" "v_max_f32_dpp %4, %4, %4 " CTRL "\n\t" "v_max_f32_dpp %5, %5, %5 " CTRL "\n\t"
#define RTK_DPP_ALL(STEP) \
    "s_nop 1\n\t" \
    STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
    STEP("row_shr:4 row_mask:0xf bank_mask:0xf") STEP("row_shr:8 row_mask:0xf bank_mask:0xf") \
    STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ float lane63(const float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// mins of a0..a5 -> a0..a5, maxes of b0..b5 -> b0..b5 (wave-uniform on return).  All 64 lanes must be active.
__device__ __forceinline__ void wave_min6_max6(float &a0, float &a1, float &a2, float &a3, float &a4, float &a5, float &b0, float &b1,
                                               float &b2, float &b3, float &b4, float &b5) {
    asm volatile(RTK_DPP_ALL(RTK_DPP_STEP12)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5));
    a0 = lane63(a0); a1 = lane63(a1); a2 = lane63(a2); a3 = lane63(a3); a4 = lane63(a4); a5 = lane63(a5);
    b0 = lane63(b0); b1 = lane63(b1); b2 = lane63(b2); b3 = lane63(b3); b4 = lane63(b4); b5 = lane63(b5);
}
// mins of a0..a2, maxes of b0..b2
__device__ __forceinline__ void wave_min3_max3(float &a0, float &a1, float &a2, float &b0, float &b1, float &b2) {
    asm volatile(RTK_DPP_ALL(RTK_DPP_STEP6) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2));
    a0 = lane63(a0); a1 = lane63(a1); a2 = lane63(a2); b0 = lane63(b0); b1 = lane63(b1); b2 = lane63(b2);
}

// bounds of the rays of the lanes with `in`
__device__ __forceinline__ Bundle make_bundle(const Ray &r, const bool cull, const bool in) {
    const float inf = __builtin_inff();
    Bundle B;
    B.olx = in ? r.o.x : inf; B.oly = in ? r.o.y : inf; B.olz = in ? r.o.z : inf;
    B.dlx = in ? r.d.x : inf; B.dly = in ? r.d.y : inf; B.dlz = in ? r.d.z : inf;
    B.ohx = in ? r.o.x : -inf; B.ohy = in ? r.o.y : -inf; B.ohz = in ? r.o.z : -inf;
    B.dhx = in ? r.d.x : -inf; B.dhy = in ? r.d.y : -inf; B.dhz = in ? r.d.z : -inf;
    wave_min6_max6(B.olx, B.oly, B.olz, B.dlx, B.dly, B.dlz, B.ohx, B.ohy, B.ohz, B.dhx, B.dhy, B.dhz);
    // a NaN component is invisible to min/max: any ray that is not finite and small switches the culling off
    const float L = kBundleLimit;
    const bool lane_ok = (__builtin_fabsf(r.o.x) <= L) & (__builtin_fabsf(r.o.y) <= L) & (__builtin_fabsf(r.o.z) <= L) &
                         (__builtin_fabsf(r.d.x) <= L) & (__builtin_fabsf(r.d.y) <= L) & (__builtin_fabsf(r.d.z) <= L);
    const bool ok = __builtin_amdgcn_ballot_w64(in & !lane_ok) == 0ull;
    B.flags = (ok ? BUNDLE_OK : 0u) | ((__builtin_amdgcn_ballot_w64(in & !cull) == 0ull) ? BUNDLE_ALL_CULL : 0u);
    return B;
}

// The direction box of the lanes with `in`, reduced to what a split needs: its widest axis, the half width and the middle.
struct DirSpread { float widest, mid; uint32_t axis; };
__device__ __forceinline__ DirSpread dir_spread(const Ray &r, const bool in) {
    const float inf = __builtin_inff();
    float lx = in ? r.d.x : inf, ly = in ? r.d.y : inf, lz = in ? r.d.z : inf;
    float hx = in ? r.d.x : -inf, hy = in ? r.d.y : -inf, hz = in ? r.d.z : -inf;
    wave_min3_max3(lx, ly, lz, hx, hy, hz);
    DirSpread S;
    S.widest = (hx - lx) * 0.5f; S.mid = (lx + hx) * 0.5f; S.axis = 0u;
    const float wy = (hy - ly) * 0.5f, wz = (hz - lz) * 0.5f;
    if (S.widest < wy) { S.widest = wy; S.axis = 1u; S.mid = (ly + hy) * 0.5f; }
    if (S.widest < wz) { S.widest = wz; S.axis = 2u; S.mid = (lz + hz) * 0.5f; }
    return S;                                            // (NaN directions: the comparisons fail, nothing is split)
}

// Pencil bundles.  Camera rays leave one point; shadow rays END in one (the light): the LINES of such a bundle pass through
// a common apex C, up to rounding.  The edge tests of the triangle only depend on the line, so they can be evaluated as if
// every ray started in C -- a bundle with a point origin, whose bounds are tight (linear in the direction box) no matter
// how far apart the real origins lie (the two sides of a silhouette, a floor seen at a grazing angle).  The price:
// tri_step's floats for the real origin differ from the exact value of that formulation by rounding errors, which the
// culling bounds explicitly (pencil_misses).  The apex is only a HINT from the caller (or found by itself when all origins
// coincide): every lane measures how far its line really is from C, and that distance `delta` enters the margins, so a
// wrong hint makes the culling loose, never wrong.
__device__ __forceinline__ float pencil_delta_lane(const Ray &r, const float cx, const float cy, const float cz) {
    const float c0 = cx - r.o.x, c1 = cy - r.o.y, c2 = cz - r.o.z;
    const float dd = (r.d.x * r.d.x + r.d.y * r.d.y) + r.d.z * r.d.z;
    const float k = ((c0 * r.d.x + c1 * r.d.y) + c2 * r.d.z) / dd;
    const float w0 = c0 - k * r.d.x, w1 = c1 - k * r.d.y, w2 = c2 - k * r.d.z;
    const float wn = (__builtin_fabsf(w0) + __builtin_fabsf(w1)) + __builtin_fabsf(w2);
    const float cn = (__builtin_fabsf(c0) + __builtin_fabsf(c1)) + __builtin_fabsf(c2);
    const float dl = wn + 1.9073486e-06f * cn;                              // 2^-19 |C - o|_1 covers the rounding of w itself
    return (dd >= 1.0e-30f) ? dl : __builtin_inff();                          // (a NaN stays a NaN and fails the `<` of the caller)
}

// Bounds + pencil data of the rays of the lanes with `in`, written to bundle slot `n` of `lds`.  Returns the flags (0 = not OK).
__device__ __forceinline__ uint32_t emit_bundle(const Ray &r, const bool cull, const bool in, const bool hinted, float cx, float cy,
                                                float cz, float *lds, const uint32_t n) {
    const Bundle B = make_bundle(r, cull, in);
    if ((B.flags & BUNDLE_OK) == 0u) return 0u;
    // fl(1/d) per axis: 1/d is monotone decreasing on either side of zero, and so is its rounding; directions with a
    // tiny or zero component anywhere in the bundle give no bound on that axis (and could overflow the products)
    uint32_t flags = B.flags;
    if ((1.0e-30f < B.dlx) | (B.dhx < -1.0e-30f)) flags |= BUNDLE_INVX;
    if ((1.0e-30f < B.dly) | (B.dhy < -1.0e-30f)) flags |= BUNDLE_INVY;
    if ((1.0e-30f < B.dlz) | (B.dhz < -1.0e-30f)) flags |= BUNDLE_INVZ;
    // apex: the caller's hint, or the origin itself when all origins coincide
    bool pencil = hinted;
    if ((B.olx == B.ohx) & (B.oly == B.ohy) & (B.olz == B.ohz)) { cx = B.olx; cy = B.oly; cz = B.olz; pencil = true; }
    float delta = 0.0f;
    if (pencil) {
        const float dl = pencil_delta_lane(r, cx, cy, cz);
        const bool bad = in & !(dl < kBundleLimit);                        // NaN-safe: such a lane switches the pencil off
        delta = wave_minmax<true>(in ? dl : 0.0f) * 1.00001f;
        const float L = kBundleLimit;
        pencil = (__builtin_amdgcn_ballot_w64(bad) == 0ull) & (__builtin_fabsf(cx) <= L) & (__builtin_fabsf(cy) <= L) &
                 (__builtin_fabsf(cz) <= L);
    }
    if (pencil) flags |= BUNDLE_PENCIL;
    // centre / radius of the direction and origin boxes for pencil_misses; the radii are inflated (1 + 1e-6, plus
    // 2^-22 (|lo| + |hi|)) so that centre +- radius covers the box in exact arithmetic
    constexpr float kInfl = 1.000001f, kAbs = 2.3841858e-07f;
    const float dcx = (B.dlx + B.dhx) * 0.5f, dcy = (B.dly + B.dhy) * 0.5f, dcz = (B.dlz + B.dhz) * 0.5f;
    const float rdx = (B.dhx - B.dlx) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.dlx) + __builtin_fabsf(B.dhx));
    const float rdy = (B.dhy - B.dly) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.dly) + __builtin_fabsf(B.dhy));
    const float rdz = (B.dhz - B.dlz) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.dlz) + __builtin_fabsf(B.dhz));
    if (__lane_id() == 0u) {
        float4 *q = reinterpret_cast<float4 *>(lds + n * (uint32_t)kBundleFloats);
        q[0] = make_float4(B.olx, B.oly, B.olz, B.ohx);
        q[1] = make_float4(B.ohy, B.ohz, B.dlx, B.dly);
        q[2] = make_float4(B.dlz, B.dhx, B.dhy, B.dhz);
        q[3] = make_float4(1.0f / B.dhx, 1.0f / B.dlx, 1.0f / B.dhy, 1.0f / B.dly);
        const float ocx = (B.olx + B.ohx) * 0.5f, ocy = (B.oly + B.ohy) * 0.5f, ocz = (B.olz + B.ohz) * 0.5f;
        const float rox = (B.ohx - B.olx) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.olx) + __builtin_fabsf(B.ohx));
        const float roy = (B.ohy - B.oly) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.oly) + __builtin_fabsf(B.ohy));
        const float roz = (B.ohz - B.olz) * 0.5f * kInfl + kAbs * (__builtin_fabsf(B.olz) + __builtin_fabsf(B.ohz));
        const float Dmx = __builtin_fabsf(dcx) + rdx, Dmy = __builtin_fabsf(dcy) + rdy, Dmz = __builtin_fabsf(dcz) + rdz;
        const float dD1 = (delta * ((Dmx + Dmy) + Dmz)) * 1.00001f;                   // delta |d|_1
        q[4] = make_float4(1.0f / B.dhz, 1.0f / B.dlz, __uint_as_float(flags), dD1);
        q[5] = make_float4(cx, cy, cz, 0.0f);
        q[6] = make_float4(dcx, dcy, dcz, rdx);
        q[7] = make_float4(rdy, rdz, ocx, ocy);
        q[8] = make_float4(ocz, rox, roy, roz);
        q[9] = make_float4(Dmx, Dmy, Dmz, (Dmx + Dmy) + Dmz);
    }
    return flags;
}

// Splits the active lanes by `cls` into at most kMaxBundles bundles (classes beyond that join the last one), writes them
// to `lds` and tells every lane which bundle its ray belongs to.  A class whose directions spread widely -- the two sides
// of a silhouette, reflections fanning out over a mesh, the occlusion queries of hits scattered over it -- is cut along
// the widest axis of its direction box, and its parts once more, while slots are left: the culling works with the UNION
// of the bundles' bounds, and four narrow boxes around clusters of rays cover far less than one wide box around them all.
// (The partition lives in `cidx` itself; any partition is a valid one, the bounds are made from whatever it is.)
// Returns the number of bundles, 0 when culling is off for this trace.
__device__ __forceinline__ uint32_t make_bundles(const Ray &r, const bool cull, const bool active, const uint32_t cls,
                                                 const V3 apex, float *lds, uint32_t &cidx) {
    const uint32_t lane = __lane_id();
    unsigned long long rem = __builtin_amdgcn_ballot_w64(active);
    uint32_t n = 0u;
    cidx = 0u;
    while (rem != 0ull) {
        const int first = __builtin_ctzll(rem);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cls, first);
        const bool mine = ((rem >> lane) & 1ull) != 0ull;
        const bool in = mine & ((n == (uint32_t)(kMaxBundles - 1)) | (cls == c));
        const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(apex.x), first));
        const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(apex.y), first));
        const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(apex.z), first));
        const bool hinted = (c & kClsHasApex) != 0u;
        rem &= ~__builtin_amdgcn_ballot_w64(in);
        // slots this class may use: all that are left, minus one for the classes still waiting
        const uint32_t room = (uint32_t)kMaxBundles - n - (rem != 0ull ? 1u : 0u);
        uint32_t cnt = 1u;
        if (in) cidx = n;
        for (int level = 0; level < kBundleSplitLevels && cnt < room; ++level) {
            const uint32_t cnt0 = cnt;
            for (uint32_t b = 0u; b < cnt0 && cnt < room; ++b) {
                const bool sel = in & (cidx == n + b);
                const DirSpread S = dir_spread(r, sel);
                if (!(S.widest > kBundleSplitRadius)) continue;
                const float dv = S.axis == 0u ? r.d.x : (S.axis == 1u ? r.d.y : r.d.z);
                const bool high = sel & !(dv <= S.mid);
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(high);
                if (hm == 0ull || hm == __builtin_amdgcn_ballot_w64(sel)) continue;       // all on one side
                if (high) cidx = n + cnt;
                cnt += 1u;
            }
        }
        for (uint32_t b = 0u; b < cnt; ++b)
            if (emit_bundle(r, cull, in & (cidx == n + b), hinted, cx, cy, cz, lds, n + b) == 0u) return 0u;
        n += cnt;
    }
    return n;
}

// Bundle images are read back from LDS with the same address in every lane and moved to SGPRs.
__device__ __forceinline__ float4 lds_uniform4(const float4 *p) {
    const float4 v = *p;
    return make_float4(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.y))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.z))),
                       __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.w))));
}
__device__ __forceinline__ uint32_t bundle_flags(const float *lds, const uint32_t k) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(lds[k * (uint32_t)kBundleFloats + 18u]));
}
struct BundleRegs {      // the interval form: boxes + bounds of 1/d
    Bundle b;
    float ilx, ihx, ily, ihy, ilz, ihz;
};
__device__ __forceinline__ BundleRegs load_bundle(const float *lds, const uint32_t k, const bool with_inv) {
    const float4 *q = reinterpret_cast<const float4 *>(lds + k * (uint32_t)kBundleFloats);
    const float4 a = lds_uniform4(q), b = lds_uniform4(q + 1), c = lds_uniform4(q + 2);
    BundleRegs R;
    R.b.olx = a.x; R.b.oly = a.y; R.b.olz = a.z; R.b.ohx = a.w;
    R.b.ohy = b.x; R.b.ohz = b.y; R.b.dlx = b.z; R.b.dly = b.w;
    R.b.dlz = c.x; R.b.dhx = c.y; R.b.dhy = c.z; R.b.dhz = c.w;
    R.b.flags = bundle_flags(lds, k);
    R.ilx = R.ihx = R.ily = R.ihy = R.ilz = R.ihz = 0.0f;
    if (with_inv) {
        const float4 d = lds_uniform4(q + 3), e = lds_uniform4(q + 4);
        R.ilx = d.x; R.ihx = d.y; R.ily = d.z; R.ihy = d.w; R.ilz = e.x; R.ihz = e.y;
    }
    return R;
}
struct PencilRegs {      // the centre / radius form (make_bundles)
    float cx, cy, cz, dD1;
    float dcx, dcy, dcz, rdx, rdy, rdz, ocx, ocy, ocz, rox, roy, roz, D1;
    uint32_t flags;
};
__device__ __forceinline__ PencilRegs load_pencil(const float *lds, const uint32_t k) {
    const float4 *q = reinterpret_cast<const float4 *>(lds + k * (uint32_t)kBundleFloats);
    const float4 e = lds_uniform4(q + 4), f = lds_uniform4(q + 5), g = lds_uniform4(q + 6), h = lds_uniform4(q + 7),
                 i = lds_uniform4(q + 8), j = lds_uniform4(q + 9);
    PencilRegs P;
    P.flags = __float_as_uint(e.z); P.dD1 = e.w;
    P.cx = f.x; P.cy = f.y; P.cz = f.z;
    P.dcx = g.x; P.dcy = g.y; P.dcz = g.z; P.rdx = g.w;
    P.rdy = h.x; P.rdz = h.y; P.ocx = h.z; P.ocy = h.w;
    P.ocz = i.x; P.rox = i.y; P.roy = i.z; P.roz = i.w;
    P.D1 = j.w;
    return P;
}

struct Iv { float lo, hi; };
__device__ __forceinline__ Iv iv_add(const Iv a, const Iv b) { return Iv{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ Iv iv_sub(const Iv a, const Iv b) { return Iv{a.lo - b.hi, a.hi - b.lo}; }
__device__ __forceinline__ Iv iv_scale(const Iv a, const float s) {           // interval * number
    const float p = a.lo * s, q = a.hi * s;
    return Iv{__builtin_fminf(p, q), __builtin_fmaxf(p, q)};
}
__device__ __forceinline__ Iv iv_mul(const Iv a, const Iv b) {                 // interval * interval
    const float p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
    return Iv{__builtin_fminf(__builtin_fminf(p, q), __builtin_fminf(r, s)), __builtin_fmaxf(__builtin_fmaxf(p, q), __builtin_fmaxf(r, s))};
}

// true: no ray of the bundle can pass triangle_packet::intersect for this lane's triangle (see "Bundle culling")
__device__ __forceinline__ bool bundle_misses(const Bundle &B, const float eps, const float v0x, const float v0y, const float v0z,
                                              const float e1x, const float e1y, const float e1z, const float e2x,
                                              const float e2y, const float e2z) {
    // (staged with scheduling barriers like pencil_misses: interleaved, the intervals need more registers than the kernels have)
    const Iv dx = {B.dlx, B.dhx}, dy = {B.dly, B.dhy}, dz = {B.dlz, B.dhz};
    // pvec = d x e2, det = e1 . pvec                                             (tri_step: pvx, pvy, pvz, det)
    const Iv pvx = iv_sub(iv_scale(dy, e2z), iv_scale(dz, e2y));
    const Iv pvy = iv_sub(iv_scale(dz, e2x), iv_scale(dx, e2z));
    const Iv pvz = iv_sub(iv_scale(dx, e2y), iv_scale(dy, e2x));
    const Iv det = iv_add(iv_add(iv_scale(pvx, e1x), iv_scale(pvy, e1y)), iv_scale(pvz, e1z));
    const bool all_cull = (B.flags & BUNDLE_ALL_CULL) != 0u;
    // eps <= det (culling rays) / eps <= |det| (the others), kd_tree_simd.hpp:33-38.  Comparisons are written so that a
    // NaN bound keeps the triangle.
    const bool none = (det.hi < eps) & (all_cull | (-eps < det.lo));              // no ray passes the determinant test
    const bool pos = !(det.hi < eps);                                             // some ray may have det >= eps
    const bool neg = !all_cull & !(-eps < det.lo);                                // some non-culling ray may have det <= -eps
    // one sign of det is left: normalise to det > 0 (x * inv_det == (-x) * (-inv_det) exactly, and fl(1/-d) == -fl(1/d))
    const float dh = neg ? -det.lo : det.hi;
    const float dl = __builtin_fmaxf(neg ? -det.hi : det.lo, eps);                // rays with a smaller determinant fail the test above
    // fl(1/det) for det in [dl, dh]: v_rcp_f32 is within 1 ulp of 1/x, fl(1/x) within 1/2; 8 ulp of slack
    const float il = __builtin_amdgcn_rcpf(dh) * 0.999999f, ih = __builtin_amdgcn_rcpf(dl) * 1.000001f;
    __builtin_amdgcn_sched_barrier(0);
    // tvec = o - v0, un = tvec . pvec
    const Iv tvx = {B.olx - v0x, B.ohx - v0x}, tvy = {B.oly - v0y, B.ohy - v0y}, tvz = {B.olz - v0z, B.ohz - v0z};
    Iv un = iv_add(iv_add(iv_mul(tvx, pvx), iv_mul(tvy, pvy)), iv_mul(tvz, pvz));
    if (neg) un = Iv{-un.hi, -un.lo};
    const float u_hi = __builtin_fmaxf(un.hi * il, un.hi * ih), u_lo = __builtin_fminf(un.lo * il, un.lo * ih);
    bool out = (u_hi < 0.0f) | (1.0f < u_lo);                                     // 0 <= u <= 1           (kd_tree_simd.hpp:47)
    __builtin_amdgcn_sched_barrier(0);
    // qvec = tvec x e1, vn = d . qvec, tn = e2 . qvec
    const Iv qx = iv_sub(iv_scale(tvy, e1z), iv_scale(tvz, e1y));
    const Iv qy = iv_sub(iv_scale(tvz, e1x), iv_scale(tvx, e1z));
    const Iv qz = iv_sub(iv_scale(tvx, e1y), iv_scale(tvy, e1x));
    __builtin_amdgcn_sched_barrier(0);
    Iv vn = iv_add(iv_add(iv_mul(dx, qx), iv_mul(dy, qy)), iv_mul(dz, qz));
    if (neg) vn = Iv{-vn.hi, -vn.lo};
    const float v_hi = __builtin_fmaxf(vn.hi * il, vn.hi * ih), v_lo = __builtin_fminf(vn.lo * il, vn.lo * ih);
    out = out | (v_hi < 0.0f) | (1.0f < u_lo + v_lo);                             // 0 <= v, u + v <= 1    (:54)
    __builtin_amdgcn_sched_barrier(0);
    Iv tn = iv_add(iv_add(iv_scale(qx, e2x), iv_scale(qy, e2y)), iv_scale(qz, e2z));
    if (neg) tn = Iv{-tn.hi, -tn.lo};
    const float t_hi = __builtin_fmaxf(tn.hi * il, tn.hi * ih);
    out = out | (t_hi <= eps);                                                    // eps < t               (:57)
    // both signs possible: the bundle straddles the triangle's plane, keep; huge determinants: no bound on 1/det, keep
    return none | (out & !(pos & neg) & (dh < 1.0e30f));
}

// true: no ray of the PENCIL bundle can pass triangle_packet::intersect for this lane's triangle.
// With o - C = alpha d + w, w perpendicular to d, |w| <= delta (pencil_delta_lane), in exact arithmetic
//     un = (o - v0).(d x e2) = d.A + w.(d x e2),   A  = e2 x (C - v0)
//     vn = d.((o - v0) x e1) = d.Bv + d.(w x e1),  Bv = (C - v0) x e1
//     det = d.Dv,  Dv = e2 x e1;     tn = e2.((o - v0) x e1) = -(o - v0).Dv
// are linear in d (tn: in o), so over the direction box dc +- rd (origin box oc +- ro) their ranges are centre +- |X|.r exactly.
// tri_step's floats differ from these by rounding: forward error analysis of its expression tree (products rounded once, sums
// of three products twice more) gives |fl(un) - un| <= 6.1 u T_un, T_un = sum_a |tv_a| (|d_b||e2_c| + |d_c||e2_b|), the same with
// T_vn = sum_a |d_a| (|tv_b||e1_c| + |tv_c||e1_b|), T_tn = sum_a |e2_a| (|tv_b||e1_c| + |tv_c||e1_b|), and 5.1 u T_det,
// T_det = sum_a |e1_a| (|d_b||e2_c| + |d_c||e2_b|), u = 2^-24; this function's own evaluation of d.A etc. errs by as much with
// |C - v0| in place of |tv|.  All of it is covered by M = 2^-20 (T(o) + T(C)) plus the delta terms (the T's themselves bounded
// through norms: sum_a x_a (y_b z_c + y_c z_b) <= |x|_1 |y|_1 max|z|).  The tests then ask for
// a clear margin: fl(un) < 0 without underflow => u < 0; fl(un) >= fl(det)(1 + 4u) => u > 1; likewise v and u + v;
// fl(tn) <= 0 => t <= 0 < eps.  tests/cull_model.py mirrors this function line by line; tests/test_bundle_cull_model.py hammers it.
// (The evaluation is staged, one linear functional after the other with scheduling barriers in between: left to itself
// the scheduler interleaves all of them and needs ~150 VGPRs, which the render kernels do not have.)
__device__ __forceinline__ bool pencil_misses(const PencilRegs &R, const float eps, const float v0x, const float v0y, const float v0z,
                                              const float e1x, const float e1y, const float e1z, const float e2x,
                                              const float e2y, const float e2z) {
    constexpr float kT = 9.5367432e-07f;                                         // 2^-20
    constexpr float kDet = 4.7683716e-07f;                                       // 2^-21
    // The T's are bounded through norms instead of being summed out: sum_a x_a (y_b z_c + y_c z_b) <= |x|_1 |y|_1 max|z|.
    const float a1x = __builtin_fabsf(e1x), a1y = __builtin_fabsf(e1y), a1z = __builtin_fabsf(e1z);
    const float a2x = __builtin_fabsf(e2x), a2y = __builtin_fabsf(e2y), a2z = __builtin_fabsf(e2z);
    const float E11 = (a1x + a1y) + a1z, E21 = (a2x + a2y) + a2z;
    const float E1m = __builtin_fmaxf(__builtin_fmaxf(a1x, a1y), a1z), E2m = __builtin_fmaxf(__builtin_fmaxf(a2x, a2y), a2z);
    // ---- det = d.Dv
    const float Dx = e2y * e1z - e2z * e1y, Dy = e2z * e1x - e2x * e1z, Dz = e2x * e1y - e2y * e1x;                  // e2 x e1
    const float M_det = kT * ((E11 * R.D1) * E2m) + 1.0e-30f;
    const float c_det = (R.dcx * Dx + R.dcy * Dy) + R.dcz * Dz;
    const float r_det = (__builtin_fabsf(Dx) * R.rdx + __builtin_fabsf(Dy) * R.rdy) + __builtin_fabsf(Dz) * R.rdz;
    const float detH = (c_det + r_det) + M_det, detL = (c_det - r_det) - M_det;
    const bool all_cull = (R.flags & BUNDLE_ALL_CULL) != 0u;
    const bool none = (detH < eps) & (all_cull | (-eps < detL));                 // no ray passes the determinant test
    const bool pos = !(detH < eps);
    const bool neg = !all_cull & !(-eps < detL);
    const float s = neg ? -1.0f : 1.0f;                                          // normalise to positive determinants
    const float dH = neg ? -detL : detH;
    const float rcp = 1.0f / dH;
    const float slack_d = kDet * dH;
    __builtin_amdgcn_sched_barrier(0);
    // ---- tn = -(o - v0).Dv over the origin box
    const float tcx = R.ocx - v0x, tcy = R.ocy - v0y, tcz = R.ocz - v0z;
    const float TV1 = ((__builtin_fabsf(tcx) + R.rox) + (__builtin_fabsf(tcy) + R.roy)) + (__builtin_fabsf(tcz) + R.roz);   // |tv|_1 bound
    const float M_tn = kT * ((E21 * TV1) * E1m) + 1.0e-30f;
    const float c_tn = -s * ((tcx * Dx + tcy * Dy) + tcz * Dz);
    const float r_tn = (__builtin_fabsf(Dx) * R.rox + __builtin_fabsf(Dy) * R.roy) + __builtin_fabsf(Dz) * R.roz;
    bool out = ((c_tn + r_tn) + M_tn) < 0.0f;                                     // t <= 0
    __builtin_amdgcn_sched_barrier(0);
    // ---- un = d.A (+ delta term), un - det = d.X1
    const float cvx = R.cx - v0x, cvy = R.cy - v0y, cvz = R.cz - v0z;
    const float TS = TV1 + ((__builtin_fabsf(cvx) + __builtin_fabsf(cvy)) + __builtin_fabsf(cvz));                   // |tv|_1 + |C - v0|_1
    const float M_un = kT * ((TS * R.D1) * E2m) + R.dD1 * E21 + 1.0e-30f;
    const float M_vn = kT * ((TS * R.D1) * E1m) + R.dD1 * E11 + 1.0e-30f;
    const float Ax = e2y * cvz - e2z * cvy, Ay = e2z * cvx - e2x * cvz, Az = e2x * cvy - e2y * cvx;                  // e2 x (C - v0)
    const float c_un = s * ((R.dcx * Ax + R.dcy * Ay) + R.dcz * Az);
    const float r_un = (__builtin_fabsf(Ax) * R.rdx + __builtin_fabsf(Ay) * R.rdy) + __builtin_fabsf(Az) * R.rdz;
    const float unH = (c_un + r_un) + M_un;
    out = out | ((unH < 0.0f) & ((-unH) * rcp >= 1.0e-30f));                      // u < 0 for every ray (and no underflow to -0)
    const float X1x = Ax - Dx, X1y = Ay - Dy, X1z = Az - Dz;
    const float c_x1 = s * ((R.dcx * X1x + R.dcy * X1y) + R.dcz * X1z);
    const float r_x1 = (__builtin_fabsf(X1x) * R.rdx + __builtin_fabsf(X1y) * R.rdy) + __builtin_fabsf(X1z) * R.rdz;
    out = out | (((c_x1 - r_x1) - (M_un + M_det) - slack_d) > 0.0f);              // u > 1
    __builtin_amdgcn_sched_barrier(0);
    // ---- vn = d.Bv (+ delta term), un + vn - det = d.X2
    const float Bx = cvy * e1z - cvz * e1y, By = cvz * e1x - cvx * e1z, Bz = cvx * e1y - cvy * e1x;                  // (C - v0) x e1
    const float c_vn = s * ((R.dcx * Bx + R.dcy * By) + R.dcz * Bz);
    const float r_vn = (__builtin_fabsf(Bx) * R.rdx + __builtin_fabsf(By) * R.rdy) + __builtin_fabsf(Bz) * R.rdz;
    const float vnH = (c_vn + r_vn) + M_vn;
    out = out | ((vnH < 0.0f) & ((-vnH) * rcp >= 1.0e-30f));                      // v < 0
    const float X2x = X1x + Bx, X2y = X1y + By, X2z = X1z + Bz;
    const float c_x2 = s * ((R.dcx * X2x + R.dcy * X2y) + R.dcz * X2z);
    const float r_x2 = (__builtin_fabsf(X2x) * R.rdx + __builtin_fabsf(X2y) * R.rdy) + __builtin_fabsf(X2z) * R.rdz;
    out = out | (((c_x2 - r_x2) - ((M_un + M_vn) + M_det) - slack_d) > 0.0f);     // u + v > 1
    return none | (out & !(pos & neg) & (dH < 1.0e30f));
}

// Leaf references [lo, hi) of the leaf starting at `first`, bundle-culled 64 at a time (one triangle per lane), the
// survivors tested exactly in leaf order.  `cidx` = the bundle of this lane's ray.
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
#ifdef RTK_DEBUG_PHASES
struct CullTally { uint32_t chunks, surv, tris; unsigned long long c_cull, c_surv; StageTally stg; };
#define RTK_TALLY_ARG , CullTally &tally
#define RTK_TALLY_PASS , tally
#else
#define RTK_TALLY_ARG
#define RTK_TALLY_PASS
#endif
__device__ __forceinline__ void leaf_range_bundle(const float *tris, const uint32_t first, const uint32_t lo, const uint32_t hi,
                                                  const Ray &r, const bool cull, const float eps, const bool pass,
                                                  const uint32_t cidx, const BundleSet &BS, const bool scalar_surv, Cand &best RTK_TALLY_ARG) {
    const unsigned long long pass_mask = __builtin_amdgcn_ballot_w64(pass);
    const uint32_t lane = __lane_id();
    for (uint32_t base = lo; base < hi; base += 64u) {
#ifdef RTK_DEBUG_PHASES
        const unsigned long long pc0 = __builtin_readcyclecounter();
#endif
        const uint32_t cnt = hi - base < 64u ? hi - base : 64u;
        const bool have = lane < cnt;
        const F3 *tp = reinterpret_cast<const F3 *>(tris + (size_t)(first + base + (have ? lane : 0u)) * 9);
        const F3 v0 = tp[0], e1 = tp[1], e2 = tp[2];
        bool keep = false;
        for (uint32_t k = 0; k < BS.n; ++k) {
            if (__builtin_amdgcn_ballot_w64(pass & (cidx == k)) == 0ull) continue;       // no ray of this bundle is in the leaf
            if ((bundle_flags(BS.lds, k) & BUNDLE_PENCIL) != 0u) {
                const PencilRegs R = load_pencil(BS.lds, k);
                keep = keep | !pencil_misses(R, eps, v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z);
            } else {
                const BundleRegs R = load_bundle(BS.lds, k, false);
                keep = keep | !bundle_misses(R.b, eps, v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z);
            }
        }
        unsigned long long surv = __builtin_amdgcn_ballot_w64(have & keep);
#ifdef RTK_DEBUG_PHASES
        tally.chunks += 1u; tally.surv += (uint32_t)__popcll(surv); tally.tris += cnt;
#endif
#ifdef RTK_DEBUG_PHASES
        const unsigned long long ps0 = __builtin_readcyclecounter();
        tally.c_cull += ps0 - pc0;
#endif
        // the survivors, in leaf order.  Their data is broadcast from the lane that holds it (nine v_readlane), or -- scalar_surv,
        // the streaming kernels -- comes back through the scalar cache (one 32-byte + one 4-byte s_load each, the next
        // survivor's issued before the current one is tested): the readlanes are VALU issue slots, which a lone wave runs out
        // of first (configs 3 / 4: -5 %), but under the megakernel's full occupancy the scalar cache is the scarcer (config 2: +9 %).
        if (scalar_surv) {
            if (surv != 0ull) {
                cptr_f32 tb = (cptr_f32)(const void *)tris + (size_t)(first + base) * 9;
                int j = __builtin_ctzll(surv);
                surv &= surv - 1ull;
                TriS cur = load_tri_uniform(tb + (size_t)j * 9);
                for (;;) {
                    const int jn = surv != 0ull ? __builtin_ctzll(surv) : j;
                    const TriS nxt = load_tri_uniform(tb + (size_t)jn * 9);
#ifdef RTK_DEBUG_PHASES
                    tri_step(cur, first + base + (uint32_t)j, r, cull, eps, pass_mask, lane, best, &tally.stg);
#else
                    tri_step(cur, first + base + (uint32_t)j, r, cull, eps, pass_mask, lane, best);
#endif
                    if (surv == 0ull) break;
                    surv &= surv - 1ull;
                    cur = nxt; j = jn;
                }
            }
        } else {
        while (surv != 0ull) {
            const int j = __builtin_ctzll(surv);
            surv &= surv - 1ull;
            TriS cur;
            cur.v0x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0.x), j));
            cur.v0y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0.y), j));
            cur.v0z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0.z), j));
            cur.e1x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1.x), j));
            cur.e1y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1.y), j));
            cur.e1z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1.z), j));
            cur.e2x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2.x), j));
            cur.e2y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2.y), j));
            cur.e2z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2.z), j));
#ifdef RTK_DEBUG_PHASES
            tri_step(cur, first + base + (uint32_t)j, r, cull, eps, pass_mask, lane, best, &tally.stg);
#else
            tri_step(cur, first + base + (uint32_t)j, r, cull, eps, pass_mask, lane, best);
#endif
        }
        }
#ifdef RTK_DEBUG_PHASES
        tally.c_surv += __builtin_readcyclecounter() - ps0;
#endif
    }
}

// leaf references [lo, hi): bundle-culled when the trace has bundles and the range is worth a 64-wide pass
__device__ __forceinline__ void leaf_range(const TreeView &T, const uint32_t first, const uint32_t lo, const uint32_t hi,
                                           const Ray &r, const bool cull, const bool pass, const uint32_t cidx,
                                           const BundleSet &BS, Cand &best RTK_TALLY_ARG) {
    if (lo >= hi) return;
    if (BS.n != 0u && hi - lo >= kBundleMinTris)
        leaf_range_bundle(reinterpret_cast<const float *>(T.tris), first, lo, hi, r, cull, T.eps, pass, cidx, BS, T.scalar_surv != 0, best RTK_TALLY_PASS);
    else
        leaf_range_wave((cptr_f32)(const void *)T.tris, first, lo, hi, r, cull, T.eps, pass, best);
}

// Workgroup-cooperative leaves (SLICES > 1).  A workgroup of SLICES waves serves ONE 8x8 pixel block: wave 0 (the
// owner) holds the 64 rays, runs the shading state machine and walks the tree; waves 1..SLICES-1 are helpers that
// sleep at a workgroup barrier until the owner reaches a leaf with at least `min_tris` triangles.  The owner then
// publishes (leaf range, per-lane best_t, pass/cull masks) in LDS, every wave tests one contiguous SLICES-th of the
// leaf, and the owner merges the winners in slice order with a strict '<' — exactly the sequential "earliest
// triangle with the smallest t" rule.  This divides the longest dependency chain of a frame (one wave grinding
// through a 500-triangle leaf) by SLICES without replicating traversal or shading work.
struct GroupShared {
    float4 ray_o[64];            // origin xyz, w = the lane's bundle index; rewritten only when the owner starts a new ray
    float4 ray_d[64];            // direction xyz
    float best_t[64];            // per-lane best t before the leaf
    unsigned long long pass_mask, cull_mask;
    uint32_t first, count;       // leaf references [first, first+count)
    uint32_t kind;               // 0 = leaf, 1 = exit
    uint32_t ray_gen;            // bumped whenever ray_o/ray_d change
    uint32_t n_bundles;          // bundles of the rays in ray_o/ray_d (0 = no culling)
    uint32_t pad[3];
    float bundles[kMaxBundles * kBundleFloats];    // written by make_bundles at the start of the owner's trace
    float4 result[][64];         // [SLICES][64] winners of slices 1..SLICES-1: t,u,v,k (storage: GroupStorage<SLICES>)
};
static_assert(offsetof(GroupShared, bundles) % 16 == 0, "bundle images are read with 16-byte LDS loads");
template <int SLICES>
struct alignas(16) GroupStorage {
    // GroupShared, result[SLICES][64], then one private bundle area per wave (helpers that trace on their own, GROUP_EXTRA)
    unsigned char raw[sizeof(GroupShared) + (size_t)SLICES * 64 * sizeof(float4) + (size_t)SLICES * kMaxBundles * kBundleFloats * sizeof(float)];
    __device__ __forceinline__ GroupShared *get() { return reinterpret_cast<GroupShared *>(raw); }
};
template <int SLICES>
__device__ __forceinline__ float *group_private_bundles(GroupShared *sh, const uint32_t slice) {
    return reinterpret_cast<float *>(&sh->result[SLICES][0]) + slice * (uint32_t)(kMaxBundles * kBundleFloats);
}
// commands the owner posts: a leaf to slice, "we are done", or a job for the kernel's own helper service (k_render: the
// occlusion queries of the other lights, kernels.hip)
enum : uint32_t { GROUP_LEAF = 0, GROUP_EXIT = 1, GROUP_EXTRA = 2 };
struct NoExtraService {
    __device__ __forceinline__ void operator()(GroupShared *, uint32_t) const {}
};

struct SliceCtx {
    GroupShared *sh;     // LDS (nullptr when SLICES == 1)
    uint32_t min_tris;   // leaves with fewer triangles are tested by the owner alone
    uint32_t ray_gen;    // owner: generation of the rays currently in LDS
    bool rays_dirty;     // owner: the current ray is not in LDS yet
    uint32_t work;       // wave-uniform tally of nodes stepped + triangles iterated (a cost estimate for scheduling)
    float *bundle_lds;   // [kMaxBundles * kBundleFloats] floats of LDS for this trace's bundles: sh->bundles when SLICES > 1,
                         // a wave-private area otherwise; nullptr = no bundle culling / leaf-list traversal
    bool rebundle = true;  // make new bundles when half of a trace's occlusion queries are answered (trace_list).  Pays where
                         // a trace is long (the megakernel's critical blocks: config 2 -5 %), not in the streaming pipeline's
                         // sorted queues (configs 3 / 4: +4 %)
#ifdef RTK_DEBUG_PHASES
    // diagnostic (tools/phase_times.py): cycles and counts of the owner's walk by phase
    unsigned long long c_small = 0, c_big = 0, c_trace = 0, c_bund = 0, c_list = 0;
    uint32_t n_steps = 0, n_small = 0, n_big = 0, t_small = 0, t_big = 0, n_trace = 0;
    CullTally tally = {0u, 0u, 0u, 0ull, 0ull, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};   // owner's own chunks / survivors / triangles seen by the bundle culling
#endif
};
#ifdef RTK_DEBUG_PHASES
#define RTK_SX_TALLY , sx.tally
#else
#define RTK_SX_TALLY
#endif

// Slice `slice` of a leaf of `count` triangles: contiguous (the merge relies on it).  A leaf with at least one full
// 64-triangle pass per wave is cut at multiples of 64, so that the bundle culling, which reads 64 triangles at a time, does
// not run on mostly empty passes (537 triangles: 9 passes in all instead of 4 x 3); a smaller leaf is cut evenly -- every
// wave then has one (partly filled) pass and a SLICES-th of the survivors, which is what its time is made of.
template <int SLICES>
__device__ __forceinline__ void slice_range(const uint32_t count, const uint32_t slice, uint32_t &lo, uint32_t &hi) {
    if (count >= 64u * (uint32_t)SLICES) {
        const uint32_t nc = (count + 63u) >> 6;
        lo = ((nc * slice) / (uint32_t)SLICES) << 6;
        hi = ((nc * (slice + 1u)) / (uint32_t)SLICES) << 6;
        lo = lo < count ? lo : count;
        hi = hi < count ? hi : count;
    } else {
        const uint32_t per = (count + (uint32_t)SLICES - 1u) / (uint32_t)SLICES;
        lo = slice * per < count ? slice * per : count;
        hi = lo + per < count ? lo + per : count;
    }
}

// helper waves: serve leaf slices (and the kernel's extra jobs) until the owner posts GROUP_EXIT
template <int SLICES, typename Extra = NoExtraService>
__device__ __forceinline__ void group_helper_loop(const TreeView &T, GroupShared *sh, const uint32_t slice, const Extra extra = Extra()) {
    const uint32_t lane = __lane_id();
    uint32_t my_gen = 0xFFFFFFFFu;
    Ray r;
    r.o = mk(0.f, 0.f, 0.f); r.d = mk(0.f, 0.f, 0.f); r.inv = mk(0.f, 0.f, 0.f);
    uint32_t cidx = 0u;
    BundleSet BS = {sh->bundles, 0u};
#ifdef RTK_DEBUG_PHASES
    CullTally tally = {0u, 0u, 0u, 0ull, 0ull, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
#endif
    for (;;) {
        __syncthreads();                                                   // B1: a command is posted
        const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->kind);
        if (kind == GROUP_EXIT) break;
        if (kind == GROUP_EXTRA) {
            extra(sh, slice);
            my_gen = 0xFFFFFFFFu;                                          // ray_o / ray_d carried the job, not rays
            __syncthreads();                                               // B2
            continue;
        }
        const uint32_t gen = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->ray_gen);
        if (gen != my_gen) {
            const float4 o = sh->ray_o[lane], d = sh->ray_d[lane];
            r.o = mk(o.x, o.y, o.z); r.d = mk(d.x, d.y, d.z);
            cidx = __float_as_uint(o.w);
            BS.n = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->n_bundles);
            my_gen = gen;
        }
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->first);
        const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->count);
        const unsigned long long pm = sh->pass_mask, cm = sh->cull_mask;
        Cand mine;
        mine.t = sh->best_t[lane]; mine.u = 0.f; mine.v = 0.f; mine.k = kMiss;
        uint32_t lo, hi;
        slice_range<SLICES>(count, slice, lo, hi);
        leaf_range(T, first, lo, hi, r, ((cm >> lane) & 1ull) != 0ull, ((pm >> lane) & 1ull) != 0ull, cidx, BS, mine RTK_TALLY_PASS);
        sh->result[slice][lane] = make_float4(mine.t, mine.u, mine.v, __uint_as_float(mine.k));
        __syncthreads();                                                   // B2: results are in LDS
    }
}

__device__ __forceinline__ void group_post_exit(GroupShared *sh) {
    if (__lane_id() == 0u) sh->kind = GROUP_EXIT;
    __syncthreads();
}

// One leaf for the lanes with `pass`: tested by this wave alone, or (SLICES > 1, leaf of at least sx.min_tris triangles) cut
// into SLICES contiguous ranges, one per wave of the workgroup, and merged in range order with a strict '<'.
template <int SLICES>
__device__ __forceinline__ void process_leaf(const TreeView &T, const uint32_t a, const uint32_t b, const Ray &r, const bool cull,
                                             const bool pass, const uint32_t cidx, const BundleSet &BS, Cand &best, SliceCtx &sx) {
    if (SLICES > 1 && b >= sx.min_tris) {
        GroupShared *sh = sx.sh;
        const uint32_t lane = __lane_id();
        if (sx.rays_dirty) {
            sh->ray_o[lane] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(cidx));
            sh->ray_d[lane] = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
            if (lane == 0u) sh->n_bundles = BS.n;                          // the bundles themselves are in sh->bundles already
            sx.ray_gen += 1u;
            sx.rays_dirty = false;
        }
        sh->best_t[lane] = best.t;
        const unsigned long long pm = __builtin_amdgcn_ballot_w64(pass), cm = __builtin_amdgcn_ballot_w64(cull);
        if (lane == 0u) {
            sh->pass_mask = pm; sh->cull_mask = cm;
            sh->first = a; sh->count = b; sh->kind = GROUP_LEAF; sh->ray_gen = sx.ray_gen;
        }
        __syncthreads();                                       // B1: helpers start on their slices
        uint32_t lo0, hi0;
        slice_range<SLICES>(b, 0u, lo0, hi0);
        leaf_range(T, a, lo0, hi0, r, cull, pass, cidx, BS, best RTK_SX_TALLY);
        __syncthreads();                                       // B2: helper results are in LDS
#pragma unroll
        for (int s = 1; s < SLICES; ++s) {
            const float4 c = sh->result[s][lane];
            if (c.x < best.t) { best.t = c.x; best.u = c.y; best.v = c.z; best.k = __float_as_uint(c.w); }
        }
    } else {
        leaf_range(T, a, 0u, b, r, cull, pass, cidx, BS, best RTK_SX_TALLY);
    }
}

// ------------------------------------------------------------------------------------------------
// Leaf-list traversal.  A lane visits a leaf iff the leaf's own box test passes (`box hit && !(best_t < t_min)`,
// kd_tree_simd.hpp:200-205) at the moment the walk reaches it: the tests of its ancestors are IMPLIED.  A child box is
// a subset of its parent's (aabb3::split, aabb3.hpp:43-60: one plane moved to the midpoint), the slab test is built from
// monotone float operations (subtract, multiply by the same 1/d, compare-select), so per axis the child's [t1, t2] lies inside
// the parent's, t_min(child) >= t_min(parent), and a hit child means a hit parent; the candidate's t only shrinks, so
// `best_t >= t_min(child)` now implies `best_t_then >= t_min(parent)` earlier.  (A NaN from 0 * inf only ever drops a constraint
// of the node it occurs in; the parent's constraint on that axis is then no stronger.)  Leaves are visited in traversal order,
// so every lane sees exactly the leaf sequence of the stack walk -- without the inner nodes.
// The list is first cut down, 64 leaves at a time (one leaf per lane), to the leaves a BUNDLE can touch at all (interval
// slab test, same monotonicity argument as the triangle culling); each survivor then costs one exact 64-ray box test.
__device__ __forceinline__ bool bundle_may_hit_box(const BundleRegs &R, const float lo0, const float lo1, const float lo2,
                                                   const float hi0, const float hi1, const float hi2) {
    const Bundle &B = R.b;
    float t_min_lo = 0.0f, t_max_hi = kFltMax;
    bool miss = false;
    if ((B.flags & BUNDLE_INVX) != 0u) {
        const Iv inv = {R.ilx, R.ihx};
        const Iv a = iv_mul(Iv{lo0 - B.ohx, lo0 - B.olx}, inv), c = iv_mul(Iv{hi0 - B.ohx, hi0 - B.olx}, inv);
        t_min_lo = __builtin_fmaxf(t_min_lo, __builtin_fminf(a.lo, c.lo));
        t_max_hi = __builtin_fminf(t_max_hi, __builtin_fmaxf(a.hi, c.hi));
        miss = t_max_hi < t_min_lo;
    }
    if ((B.flags & BUNDLE_INVY) != 0u) {
        const Iv inv = {R.ily, R.ihy};
        const Iv a = iv_mul(Iv{lo1 - B.ohy, lo1 - B.oly}, inv), c = iv_mul(Iv{hi1 - B.ohy, hi1 - B.oly}, inv);
        t_min_lo = __builtin_fmaxf(t_min_lo, __builtin_fminf(a.lo, c.lo));
        t_max_hi = __builtin_fminf(t_max_hi, __builtin_fmaxf(a.hi, c.hi));
        miss = miss | (t_max_hi < t_min_lo);
    }
    if ((B.flags & BUNDLE_INVZ) != 0u) {
        const Iv inv = {R.ilz, R.ihz};
        const Iv a = iv_mul(Iv{lo2 - B.ohz, lo2 - B.olz}, inv), c = iv_mul(Iv{hi2 - B.ohz, hi2 - B.olz}, inv);
        t_min_lo = __builtin_fmaxf(t_min_lo, __builtin_fminf(a.lo, c.lo));
        t_max_hi = __builtin_fminf(t_max_hi, __builtin_fmaxf(a.hi, c.hi));
        miss = miss | (t_max_hi < t_min_lo);
    }
    return !miss;
}

#ifndef RTK_REBUNDLE_MIN_LEAVES
#define RTK_REBUNDLE_MIN_LEAVES 2
#endif
constexpr uint32_t kRebundleMinLeaves = RTK_REBUNDLE_MIN_LEAVES;   // new bundles cost about one 64-triangle pass: only with leaves left to use them on
constexpr uint32_t kListMaxLeaves = 512;    // larger trees keep the hierarchical walk

template <int SLICES>
__device__ __forceinline__ void trace_list(const TreeView &T, const Ray &r, const bool cull, const bool active, Cand &best,
                                           SliceCtx &sx, const float exit_t, uint32_t cidx, BundleSet BS, const uint32_t cls,
                                           const V3 apex) {
    const uint32_t lane = __lane_id();
    bool live = active;                                                    // lanes still looking for their closest hit
    // RTK_TRAVERSAL_FAST (rtk.h): the leaf list of the octant most of the wave's rays travel in, front to back.  Any order visits
    // a superset of what each ray needs for its closest t (a leaf is skipped only when a nearer hit is already known); only
    // which of several triangles at exactly that t wins depends on the order.
    const DevNode *leaf_list = T.leaves;
    if (T.leaves_fast != nullptr) {
        const uint32_t half = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active));
        uint32_t oct = 0u;
        if (2u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active & (r.d.x < 0.0f))) > half) oct |= 1u;
        if (2u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active & (r.d.y < 0.0f))) > half) oct |= 2u;
        if (2u * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active & (r.d.z < 0.0f))) > half) oct |= 4u;
        leaf_list = T.leaves_fast + (size_t)oct * T.n_leaves;
    }
    uint32_t bundled = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active));   // live lanes when the bundles were last made
    for (uint32_t base = 0; base < T.n_leaves; base += 64u) {
#ifdef RTK_DEBUG_PHASES
        const unsigned long long pl0 = __builtin_readcyclecounter();
#endif
        const bool have = base + lane < T.n_leaves;
        const float4 *lp = reinterpret_cast<const float4 *>(leaf_list + (have ? base + lane : 0u));
        const float4 q0 = lp[0], q1 = lp[1];
        bool cand = false;
#pragma unroll
        for (int k = 0; k < kMaxBundles; ++k) {
            if ((uint32_t)k < BS.n) {
                const BundleRegs R = load_bundle(BS.lds, (uint32_t)k, true);
                cand = cand | bundle_may_hit_box(R, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
            }
        }
        unsigned long long cm = __builtin_amdgcn_ballot_w64(have & cand);
#ifdef RTK_DEBUG_PHASES
        sx.n_steps += (uint32_t)__popcll(cm);
        sx.c_list += __builtin_readcyclecounter() - pl0;
#endif
        while (cm != 0ull) {
            const int j = __builtin_ctzll(cm);
            cm &= cm - 1ull;
            const float lo0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q0.x), j));
            const float lo1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q0.y), j));
            const float lo2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q0.z), j));
            const float hi0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q0.w), j));
            const float hi1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q1.x), j));
            const float hi2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q1.y), j));
            float t_min;
            const bool box = slab(lo0, lo1, lo2, hi0, hi1, hi2, r, t_min);
            const bool pass = live & box & !(best.t < t_min);              // kd_tree_simd.hpp:202-205
            if (!wave_any(pass)) continue;
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(q1.z), j);
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(q1.w), j);
            sx.work += b + 2u;
#ifdef RTK_DEBUG_PHASES
            const unsigned long long ph0 = __builtin_readcyclecounter();
#endif
            process_leaf<SLICES>(T, a, b, r, cull, pass, cidx, BS, best, sx);
#ifdef RTK_DEBUG_PHASES
            {
                const unsigned long long ph1 = __builtin_readcyclecounter();
                if (SLICES > 1 && b >= sx.min_tris) { sx.c_big += ph1 - ph0; sx.n_big += 1u; sx.t_big += b; }
                else { sx.c_small += ph1 - ph0; sx.n_small += 1u; sx.t_small += b; }
            }
#endif
            // occlusion queries: a lane whose hit already answers the query stops; when nobody is left the walk ends
            if (best.t <= exit_t) live = false;
            const uint32_t n_live = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
            if (n_live == 0u) return;
            // occlusion queries thin out as they are answered: once half of the rays the bundles were made for are gone, bundles
            // of the remaining ones are tighter (the candidate leaves found with the old bundles stay a valid superset)
            if (sx.rebundle && n_live * 2u <= bundled && (uint32_t)__popcll(cm) >= kRebundleMinLeaves) {
#ifdef RTK_DEBUG_PHASES
                const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
                BS.n = make_bundles(r, cull, live, cls, apex, sx.bundle_lds, cidx);
#ifdef RTK_DEBUG_PHASES
                sx.c_bund += __builtin_readcyclecounter() - pb0;
#endif
                bundled = n_live;
                sx.rays_dirty = true;                                       // helpers re-read the bundles and the lanes' bundle indices
            }
        }
    }
}

// The stack walk of kd_tree_simd.hpp:191-228 for the 64 rays of a wave at once (see "Wave-cooperative traversal" above).
template <bool STATS, int SLICES>
__device__ __forceinline__ uint32_t trace_wave(const TreeView &T, const Ray &r, const bool cull, const bool active,
                                               Cand &best, Stats &st, const uint32_t min_lanes, SliceCtx &sx,
                                               const float exit_t, const uint32_t cidx, const BundleSet &BS) {
    const uint32_t end = T.n_nodes;
    uint32_t next = active ? 0u : end;
    uint32_t n = 0;
    cptr_u32 nodes = (cptr_u32)(const void *)T.nodes;
    NodeS cur = load_node_uniform(nodes);
    uint32_t have = 0u;                                                    // `cur` holds node `have`
    while (n < end) {
        // ONE 32-byte scalar load per node (split into "a,b now, box later" by the compiler it cost two serial scalar-cache
        // round trips per step), and the descend-successor n+1 is requested before node n is tested: only a jump to a
        // skip target waits for memory.
        if (have != n) cur = load_node_uniform(nodes + (size_t)n * 8);
        const uint32_t n_ahead = n + 1u < end ? n + 1u : n;
        const NodeS ahead = load_node_uniform(nodes + (size_t)n_ahead * 8);
        const float lo0 = cur.lo0, lo1 = cur.lo1, lo2 = cur.lo2, hi0 = cur.hi0, hi1 = cur.hi1, hi2 = cur.hi2;
        const uint32_t a = cur.a, b = cur.b;
        cur = ahead; have = n_ahead;
        const bool part = (next == n);
        const unsigned long long part_mask = __builtin_amdgcn_ballot_w64(part);
        if (part_mask == 0ull) {                                           // nobody is waiting here
            n = (b == DEV_INNER) ? a : n + 1;
            continue;
        }
        if ((uint32_t)__popcll(part_mask) < min_lanes) break;              // too few rays agree: hand over to per-lane
        float t_min;
        const bool box = slab(lo0, lo1, lo2, hi0, hi1, hi2, r, t_min);
        const bool pass = part & box & !(best.t < t_min);
        if (STATS) { st.nodes += part ? 1u : 0u; st.boxpass += pass ? 1u : 0u; }
        const bool any_pass = wave_any(pass);
        sx.work += (any_pass && b != DEV_INNER) ? b + 2u : 2u;
#ifdef RTK_DEBUG_PHASES
        sx.n_steps += 1u;
#endif
        if (b == DEV_INNER) {
            if (part) next = pass ? n + 1 : a;
            n = any_pass ? n + 1 : a;
        } else {
            if (part) next = n + 1;
            if (any_pass) {
                if (STATS && pass) { st.leaves += 1; st.tris += b; st.packets16 += (b + 15u) >> 4; }
#ifdef RTK_DEBUG_PHASES
                const unsigned long long ph0 = __builtin_readcyclecounter();
#endif
                process_leaf<SLICES>(T, a, b, r, cull, pass, cidx, BS, best, sx);
#ifdef RTK_DEBUG_PHASES
                {
                    const unsigned long long ph1 = __builtin_readcyclecounter();
                    if (SLICES > 1 && b >= sx.min_tris) { sx.c_big += ph1 - ph0; sx.n_big += 1u; sx.t_big += b; }
                    else { sx.c_small += ph1 - ph0; sx.n_small += 1u; sx.t_small += b; }
                }
#endif
                // occlusion queries: a lane whose hit already answers the query stops; when nobody is left the walk ends
                if (best.t <= exit_t) next = end;
                if (__builtin_amdgcn_ballot_w64(next < end) == 0ull) break;
            }
            n = n + 1;
        }
    }
    return next;
}

// Closest hit for the wave's rays.  MODE: RTK_TRACE_LANE, RTK_TRACE_WAVE or RTK_TRACE_AUTO
// (wave-cooperative while at least `kAutoMinLanes` rays share the node, then per-lane from where each ray stands).
constexpr uint32_t kAutoMinLanes = 12;

// `exit_t` (per lane, default "never"): the caller only wants to know whether the closest hit has t <= exit_t
// (is_occluded, render.hpp:110-131, for scenes without transmissive materials).  The lane then stops at the end of
// the first leaf that gives it such a hit: what it has evaluated up to there is a PREFIX of what the reference
// evaluates (same order, same pruning), the reference's closest hit can only be nearer, so the answer is the same.
// `cls` (per lane): the caller's name for the coherent class the lane's ray belongs to (camera rays, shadow rays towards
// light k, ...); with kClsHasApex set, `apex` is a point the lines of the class's rays (should) pass through.  Both only steer
// the bundle culling, never a result.
// The root box test of a trace for callers that want to know, before they set anything else up, whether any ray enters the
// tree at all (most traces of a frame do not: background).  Lanes that fail it can never hit anything.
__device__ __forceinline__ bool enters_root(const TreeView &T, const Ray &r, const bool active) {
    cptr_u32 nodes = (cptr_u32)(const void *)T.nodes;
    const NodeS root = load_node_uniform(nodes);
    float t_min;
    return active & slab(root.lo0, root.lo1, root.lo2, root.hi0, root.hi1, root.hi2, r, t_min);
}

// ROOT_DONE: the caller has applied enters_root() already and passes only those lanes as `active` (never with STATS: the
// counting build charges the root node to every ray).
template <int MODE, bool STATS, bool LDS_NODES, int SLICES = 1, bool ROOT_DONE = false>
__device__ __forceinline__ Cand trace(const TreeView &T, const DevNode *lds_nodes, const Ray &r, const bool cull,
                                      const bool active, Stats &st, SliceCtx &sx, const uint32_t auto_min = kAutoMinLanes,
                                      const float exit_t = -1.0f, const uint32_t cls = 0u, const V3 apex = V3{0.f, 0.f, 0.f}) {
    static_assert(!(ROOT_DONE && STATS), "the counting build walks every node, the root included");
    Cand best;
    best.t = kFltMax; best.u = 0.0f; best.v = 0.0f; best.k = kMiss;
    sx.rays_dirty = true;
    if (MODE == RTK_TRACE_LANE) {
        trace_lane_from<STATS, LDS_NODES>(T, lds_nodes, r, cull, active ? 0u : T.n_nodes, best, st, exit_t);
    } else if (MODE == RTK_TRACE_WAVE) {
        if (wave_any(active)) {
            // root box first: most traces of a frame end here (background), before anything is spent on bundles; rays that
            // miss it take no further part (they cannot hit anything, and would only widen the bundles)
            const bool in = (ROOT_DONE || STATS) ? active : enters_root(T, r, active);
            if (wave_any(in)) {
                BundleSet BS = {sx.bundle_lds, 0u};
                uint32_t cidx = 0u;
#ifdef RTK_DEBUG_PHASES
                const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
                if (T.bundle_cull != 0 && sx.bundle_lds != nullptr) BS.n = make_bundles(r, cull, in, cls, apex, sx.bundle_lds, cidx);
#ifdef RTK_DEBUG_PHASES
                sx.c_bund += __builtin_readcyclecounter() - pb0;
#endif
                if (!STATS && BS.n != 0u && T.n_leaves <= kListMaxLeaves) trace_list<SLICES>(T, r, cull, in, best, sx, exit_t, cidx, BS, cls, apex);
                else (void)trace_wave<STATS, SLICES>(T, r, cull, in, best, st, 1u, sx, exit_t, cidx, BS);
            }
        }
    } else {
        const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
        if (am != 0ull) {
            uint32_t next = active ? 0u : T.n_nodes;
            const BundleSet none = {nullptr, 0u};
            if ((uint32_t)__popcll(am) >= auto_min) next = trace_wave<STATS, 1>(T, r, cull, active, best, st, auto_min, sx, exit_t, 0u, none);
            trace_lane_from<STATS, LDS_NODES>(T, lds_nodes, r, cull, next, best, st, exit_t);
        }
    }
    if (STATS && best.k != kMiss) st.hits += 1;
    return best;
}

// Hit reconstruction, kd_tree_simd.hpp:234-263.
struct Surface {
    V3 hit_normal, face_normal;
    uint32_t tri, mesh, material;
    float w;
};
__device__ __forceinline__ Surface reconstruct(const TreeView &T, const Cand &c) {
    Surface s;
    s.tri = T.tri_ids[c.k];
    const DevShade *sh = T.shade + s.tri;
    const float4 a = *reinterpret_cast<const float4 *>(sh);
    const float4 b = *(reinterpret_cast<const float4 *>(sh) + 1);
    const float4 cc = *(reinterpret_cast<const float4 *>(sh) + 2);
    const float4 d = *(reinterpret_cast<const float4 *>(sh) + 3);
    const V3 n0 = mk(a.x, a.y, a.z), n1 = mk(a.w, b.x, b.y), n2 = mk(b.z, b.w, cc.x);
    s.face_normal = mk(cc.y, cc.z, cc.w);
    s.mesh = __float_as_uint(d.x);
    s.material = __float_as_uint(d.y);
    s.w = 1.0f - c.u - c.v;
    V3 hn = (c.u * n1 + c.v * n2) + s.w * n0;
    if (T.normalize) hn = normalized(hn);                                  // kd_tree_simd.hpp:250 vs kd_tree.hpp:140
    s.hit_normal = hn;
    return s;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

}  // namespace dev
}  // namespace rtk
