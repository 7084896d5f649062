// Device-side traversal and intersection for gfx950 (wave64).  Compile with -ffp-contract=off:
// every expression keeps the reference's operation order so IEEE add/mul/div/sqrt reproduce the
// CPU results bit for bit.  Reference paths are relative to
// /root/reference/include/raytracer/.
#pragma once

#include <hip/hip_runtime.h>

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
    uint32_t n_nodes;
    float eps;
    int normalize;
#ifdef RTK_DEBUG_KHIST
    unsigned long long *khist;   // diagnostic: leaf visits by number of participating rays (see tools/khist.py)
#endif
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
__device__ __forceinline__ void tri_step(const TriS &cur, const uint32_t k, const Ray &r, const bool cull, const float eps,
                                         const unsigned long long pass_mask, const uint32_t lane, Cand &best) {
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
    const float inv_det = (1.0f / det);
    const float u = un * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= u) & __builtin_amdgcn_ballot_w64(u <= 1.0f);
    if (m == 0ull) return;
    const float qx = tvy * cur.e1z - tvz * cur.e1y;
    const float qy = tvz * cur.e1x - tvx * cur.e1z;
    const float qz = tvx * cur.e1y - tvy * cur.e1x;
    const float v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= v) & __builtin_amdgcn_ballot_w64(u + v <= 1.0f);
    if (m == 0ull) return;
    const float t = (cur.e2x * qx + cur.e2y * qy + cur.e2z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(eps < t) & __builtin_amdgcn_ballot_w64(t < best.t);
    if (m == 0ull) return;
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

// ---- Vector-broadcast variant (experimental, -DRTK_VEC_TREE; measured slower than the scalar-load variant) -------------
// Scalar loads complete out of order, so a wave can only wait for ALL of them (lgkmcnt(0)): the prefetch can never be
// more than one triangle deep, and the 225 KB of leaf references stream through a 16 KB scalar cache, so most fetches
// pay an L2 round trip (measured in situ: ~550 cycles per triangle step, 950 per node step, against ~150 cycles of
// arithmetic).  Vector loads with the same address in every lane (one request, broadcast by the TA) return IN order:
// vmcnt lets triangle k be consumed while k+1 and k+2 are in flight.  The loop is unrolled by three so that the three
// register sets rotate without moves.  `vz` is a per-lane zero the compiler cannot see through; without it the loads
// would be turned back into scalar loads.
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint32_t opaque_lane_zero() {
    uint32_t z;
    asm("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ TriS load_tri_bcast(const float *tp, const uint32_t vz) {
    const char *q = reinterpret_cast<const char *>(tp) + vz;
    const f32x4_a4 a = *reinterpret_cast<const f32x4_a4 *>(q);
    const f32x4_a4 b = *reinterpret_cast<const f32x4_a4 *>(q + 16);
    TriS t;
    t.v0x = a.x; t.v0y = a.y; t.v0z = a.z; t.e1x = a.w;
    t.e1y = b.x; t.e1z = b.y; t.e2x = b.z; t.e2y = b.w;
    t.e2z = *reinterpret_cast<const float *>(q + 32);
    return t;
}

// One triangle against the wave's rays: the arithmetic of test_triangle with wave-level early outs between the stages
// (the per-lane predicate is a 64-bit mask in SGPRs, see leaf_range_wave).
__device__ __forceinline__ void tri_step_wave(const TriS &cur, const uint32_t k, const Ray &r, const bool cull, const float eps,
                                              const unsigned long long pass_mask, const uint32_t lane, Cand &best) {
    const float pvx = r.d.y * cur.e2z - r.d.z * cur.e2y;
    const float pvy = r.d.z * cur.e2x - r.d.x * cur.e2z;
    const float pvz = r.d.x * cur.e2y - r.d.y * cur.e2x;
    const float det = cur.e1x * pvx + cur.e1y * pvy + cur.e1z * pvz;
    unsigned long long m = pass_mask & __builtin_amdgcn_ballot_w64(eps <= (cull ? det : __builtin_fabsf(det)));
    if (m == 0ull) return;
    const float inv_det = (1.0f / det);
    const float tvx = r.o.x - cur.v0x, tvy = r.o.y - cur.v0y, tvz = r.o.z - cur.v0z;
    const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= u) & __builtin_amdgcn_ballot_w64(u <= 1.0f);
    if (m == 0ull) return;
    const float qx = tvy * cur.e1z - tvz * cur.e1y;
    const float qy = tvz * cur.e1x - tvx * cur.e1z;
    const float qz = tvx * cur.e1y - tvy * cur.e1x;
    const float v = (r.d.x * qx + r.d.y * qy + r.d.z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(0.0f <= v) & __builtin_amdgcn_ballot_w64(u + v <= 1.0f);
    if (m == 0ull) return;
    const float t = (cur.e2x * qx + cur.e2y * qy + cur.e2z * qz) * inv_det;
    m &= __builtin_amdgcn_ballot_w64(eps < t) & __builtin_amdgcn_ballot_w64(t < best.t);
    if (m == 0ull) return;
    if ((m >> lane) & 1ull) { best.t = t; best.u = u; best.v = v; best.k = k; }
}

__device__ __forceinline__ void leaf_range_bcast(const float *tris, const uint32_t vz, const uint32_t first, const uint32_t lo,
                                                 const uint32_t hi, const Ray &r, const bool cull, const float eps,
                                                 const bool pass, Cand &best) {
    if (lo >= hi) return;
    const unsigned long long pass_mask = __builtin_amdgcn_ballot_w64(pass);
    const uint32_t lane = __lane_id();
    const float *base = tris + (size_t)first * 9;
    const uint32_t last = hi - 1u;                                         // prefetches past the range re-read its last triangle
    uint32_t k = lo;
    TriS A = load_tri_bcast(base + (size_t)k * 9, vz);
    TriS B = load_tri_bcast(base + (size_t)(k + 1u < hi ? k + 1u : last) * 9, vz);
    TriS C = load_tri_bcast(base + (size_t)(k + 2u < hi ? k + 2u : last) * 9, vz);
    for (;;) {
        tri_step_wave(A, first + k, r, cull, eps, pass_mask, lane, best);
        if (k + 1u >= hi) break;
        A = load_tri_bcast(base + (size_t)(k + 3u < hi ? k + 3u : last) * 9, vz);
        tri_step_wave(B, first + k + 1u, r, cull, eps, pass_mask, lane, best);
        if (k + 2u >= hi) break;
        B = load_tri_bcast(base + (size_t)(k + 4u < hi ? k + 4u : last) * 9, vz);
        tri_step_wave(C, first + k + 2u, r, cull, eps, pass_mask, lane, best);
        if (k + 3u >= hi) break;
        C = load_tri_bcast(base + (size_t)(k + 5u < hi ? k + 5u : last) * 9, vz);
        k += 3u;
    }
}

#ifndef RTK_VEC_TREE
#define RTK_LEAF_RANGE(T_, tris_, first_, lo_, hi_, r_, cull_, pass_, best_) \
    leaf_range_wave((cptr_f32)(const void *)(T_).tris, first_, lo_, hi_, r_, cull_, (T_).eps, pass_, best_)
#else
#define RTK_LEAF_RANGE(T_, tris_, first_, lo_, hi_, r_, cull_, pass_, best_) \
    leaf_range_bcast(reinterpret_cast<const float *>((T_).tris), vz, first_, lo_, hi_, r_, cull_, (T_).eps, pass_, best_)
#endif

// Workgroup-cooperative leaves (SLICES > 1).  A workgroup of SLICES waves serves ONE 8x8 pixel block: wave 0 (the
// owner) holds the 64 rays, runs the shading state machine and walks the tree; waves 1..SLICES-1 are helpers that
// sleep at a workgroup barrier until the owner reaches a leaf with at least `min_tris` triangles.  The owner then
// publishes (leaf range, per-lane best_t, pass/cull masks) in LDS, every wave tests one contiguous SLICES-th of the
// leaf, and the owner merges the winners in slice order with a strict '<' — exactly the sequential "earliest
// triangle with the smallest t" rule.  This divides the longest dependency chain of a frame (one wave grinding
// through a 500-triangle leaf) by SLICES without replicating traversal or shading work.
struct GroupShared {
    float4 ray_o[64];            // origin xyz (w unused); rewritten only when the owner starts a new ray
    float4 ray_d[64];            // direction xyz
    float best_t[64];            // per-lane best t before the leaf
    unsigned long long pass_mask, cull_mask;
    uint32_t first, count;       // leaf references [first, first+count)
    uint32_t kind;               // 0 = leaf, 1 = exit
    uint32_t ray_gen;            // bumped whenever ray_o/ray_d change
    uint32_t pad[2];
    float4 result[][64];         // [SLICES][64] winners of slices 1..SLICES-1: t,u,v,k (storage: GroupStorage<SLICES>)
};
template <int SLICES>
struct alignas(16) GroupStorage {
    unsigned char raw[sizeof(GroupShared) + (size_t)SLICES * 64 * sizeof(float4)];
    __device__ __forceinline__ GroupShared *get() { return reinterpret_cast<GroupShared *>(raw); }
};
enum : uint32_t { GROUP_LEAF = 0, GROUP_EXIT = 1 };

struct SliceCtx {
    GroupShared *sh;     // LDS (nullptr when SLICES == 1)
    uint32_t min_tris;   // leaves with fewer triangles are tested by the owner alone
    uint32_t ray_gen;    // owner: generation of the rays currently in LDS
    bool rays_dirty;     // owner: the current ray is not in LDS yet
    uint32_t work;       // wave-uniform tally of nodes stepped + triangles iterated (a cost estimate for scheduling)
#ifdef RTK_DEBUG_PHASES
    // diagnostic (tools/phase_times.py): cycles and counts of the owner's walk by phase
    unsigned long long c_small = 0, c_big = 0, c_trace = 0;
    uint32_t n_steps = 0, n_small = 0, n_big = 0, t_small = 0, t_big = 0, n_trace = 0;
#endif
};

// helper waves: serve leaf slices until the owner posts GROUP_EXIT
template <int SLICES>
__device__ __forceinline__ void group_helper_loop(const TreeView &T, GroupShared *sh, const uint32_t slice) {
    const uint32_t vz = opaque_lane_zero();
    (void)vz;
    const uint32_t lane = __lane_id();
    uint32_t my_gen = 0xFFFFFFFFu;
    Ray r;
    r.o = mk(0.f, 0.f, 0.f); r.d = mk(0.f, 0.f, 0.f); r.inv = mk(0.f, 0.f, 0.f);
    for (;;) {
        __syncthreads();                                                   // B1: a command is posted
        const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->kind);
        if (kind == GROUP_EXIT) break;
        const uint32_t gen = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->ray_gen);
        if (gen != my_gen) {
            const float4 o = sh->ray_o[lane], d = sh->ray_d[lane];
            r.o = mk(o.x, o.y, o.z); r.d = mk(d.x, d.y, d.z);
            my_gen = gen;
        }
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->first);
        const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->count);
        const unsigned long long pm = sh->pass_mask, cm = sh->cull_mask;
        Cand mine;
        mine.t = sh->best_t[lane]; mine.u = 0.f; mine.v = 0.f; mine.k = kMiss;
        const uint32_t lo = (count * slice) / (uint32_t)SLICES, hi = (count * (slice + 1u)) / (uint32_t)SLICES;
        RTK_LEAF_RANGE(T, tris, first, lo, hi, r, ((cm >> lane) & 1ull) != 0ull, ((pm >> lane) & 1ull) != 0ull, mine);
        sh->result[slice][lane] = make_float4(mine.t, mine.u, mine.v, __uint_as_float(mine.k));
        __syncthreads();                                                   // B2: results are in LDS
    }
}

__device__ __forceinline__ void group_post_exit(GroupShared *sh) {
    if (__lane_id() == 0u) sh->kind = GROUP_EXIT;
    __syncthreads();
}

template <bool STATS, int SLICES>
__device__ __forceinline__ uint32_t trace_wave(const TreeView &T, const Ray &r, const bool cull, const bool active,
                                               Cand &best, Stats &st, const uint32_t min_lanes, SliceCtx &sx,
                                               const float exit_t = -1.0f) {
    const uint32_t end = T.n_nodes;
    uint32_t next = active ? 0u : end;
    uint32_t n = 0;
    const uint32_t vz = opaque_lane_zero();
    (void)vz;
#ifndef RTK_VEC_TREE
    cptr_u32 nodes = (cptr_u32)(const void *)T.nodes;
    NodeS cur = load_node_uniform(nodes);
    uint32_t have = 0u;                                                    // `cur` holds node `have`
#else
    // node n is read with two broadcast vector loads; node n+1 (where the walk goes when somebody descends, and
    // after every leaf) is requested before node n is used, so only the jumps to a skip target wait for memory
    const char *nodes_b = reinterpret_cast<const char *>(T.nodes);
    float4 q0 = *reinterpret_cast<const float4 *>(nodes_b + vz);
    float4 q1 = *reinterpret_cast<const float4 *>(nodes_b + vz + 16);
    uint32_t have = 0u;                                                    // q0/q1 hold node `have`
#endif
    while (n < end) {
#ifndef RTK_VEC_TREE
        // ONE 32-byte scalar load per node (split into "a,b now, box later" by the compiler it cost two serial scalar-cache
        // round trips per step), and the descend-successor n+1 is requested before node n is tested: only a jump to a
        // skip target waits for memory.
        if (have != n) cur = load_node_uniform(nodes + (size_t)n * 8);
        const uint32_t n_ahead = n + 1u < end ? n + 1u : n;
        const NodeS ahead = load_node_uniform(nodes + (size_t)n_ahead * 8);
        const float lo0 = cur.lo0, lo1 = cur.lo1, lo2 = cur.lo2, hi0 = cur.hi0, hi1 = cur.hi1, hi2 = cur.hi2;
        const uint32_t a = cur.a, b = cur.b;
        cur = ahead; have = n_ahead;
#else
        if (have != n) {
            const char *np = nodes_b + (size_t)n * 32;
            q0 = *reinterpret_cast<const float4 *>(np + vz);
            q1 = *reinterpret_cast<const float4 *>(np + vz + 16);
        }
        const uint32_t n_ahead = n + 1u < end ? n + 1u : n;
        const char *np1 = nodes_b + (size_t)n_ahead * 32;
        const float4 p0 = *reinterpret_cast<const float4 *>(np1 + vz);
        const float4 p1 = *reinterpret_cast<const float4 *>(np1 + vz + 16);
        const float lo0 = q0.x, lo1 = q0.y, lo2 = q0.z, hi0 = q0.w, hi1 = q1.x, hi2 = q1.y;
        const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q1.z));
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q1.w));
        q0 = p0; q1 = p1; have = n_ahead;
#endif
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
#ifdef RTK_DEBUG_KHIST
                const uint32_t k = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(pass));
                if (STATS && __lane_id() == 0u) {
                    const uint32_t kb = k <= 2u ? 0u : k <= 4u ? 1u : k <= 8u ? 2u : k <= 16u ? 3u : k <= 32u ? 4u : 5u;
                    const uint32_t tb = b < 12u ? 0u : b < 64u ? 1u : b < 192u ? 2u : 3u;
                    atomicAdd(T.khist + tb * 8u + kb, (unsigned long long)b);
                    atomicAdd(T.khist + 32u + tb * 8u + kb, 1ull);
                }
#endif
#ifdef RTK_DEBUG_PHASES
                const unsigned long long ph0 = __builtin_readcyclecounter();
#endif
                if (SLICES > 1 && b >= sx.min_tris) {
                    GroupShared *sh = sx.sh;
                    const uint32_t lane = __lane_id();
                    if (sx.rays_dirty) {
                        sh->ray_o[lane] = make_float4(r.o.x, r.o.y, r.o.z, 0.f);
                        sh->ray_d[lane] = make_float4(r.d.x, r.d.y, r.d.z, 0.f);
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
                    RTK_LEAF_RANGE(T, tris, a, 0u, b / (uint32_t)SLICES, r, cull, pass, best);
                    __syncthreads();                                       // B2: helper results are in LDS
#pragma unroll
                    for (int s = 1; s < SLICES; ++s) {
                        const float4 c = sh->result[s][lane];
                        if (c.x < best.t) { best.t = c.x; best.u = c.y; best.v = c.z; best.k = __float_as_uint(c.w); }
                    }
                } else {
                    RTK_LEAF_RANGE(T, tris, a, 0u, b, r, cull, pass, best);
                }
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
template <int MODE, bool STATS, bool LDS_NODES, int SLICES = 1>
__device__ __forceinline__ Cand trace(const TreeView &T, const DevNode *lds_nodes, const Ray &r, const bool cull,
                                      const bool active, Stats &st, SliceCtx &sx, const uint32_t auto_min = kAutoMinLanes,
                                      const float exit_t = -1.0f) {
    Cand best;
    best.t = kFltMax; best.u = 0.0f; best.v = 0.0f; best.k = kMiss;
    sx.rays_dirty = true;
    if (MODE == RTK_TRACE_LANE) {
        trace_lane_from<STATS, LDS_NODES>(T, lds_nodes, r, cull, active ? 0u : T.n_nodes, best, st, exit_t);
    } else if (MODE == RTK_TRACE_WAVE) {
        if (wave_any(active)) (void)trace_wave<STATS, SLICES>(T, r, cull, active, best, st, 1u, sx, exit_t);
    } else {
        const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
        if (am != 0ull) {
            uint32_t next = active ? 0u : T.n_nodes;
            if ((uint32_t)__popcll(am) >= auto_min) next = trace_wave<STATS, 1>(T, r, cull, active, best, st, auto_min, sx, exit_t);
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
