// Kernel argument blocks and launchers shared by kernels.hip and api.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "rtk_internal.hpp"

namespace rtk {

constexpr int kMaxRayDepth = 16;                 // frame-stack capacity of the render kernel
constexpr int kRayCounterShards = 64;              // k_render spreads its per-wave ray-count atomics over this many words
constexpr int kCriticalWord = 8 + kRayCounterShards;   // [kRayCounterShards] longest pixel block of the frame in s_memrealtime ticks (10 ns), sharded
constexpr unsigned long long kCriticalMinTicks = 1000;   // blocks shorter than 10 us do not report (one atomic per block would serialise the frame)
constexpr int kCounterWords = 8 + 2 * kRayCounterShards;
constexpr int kCostBins = 64;                      // two-pass frames: log-scale cost bins for longest-first scheduling
constexpr uint32_t kSliceMinTrisDefault = 33;    // GROUP modes: leaves of at least two 64-triangle chunks are cut across the waves
constexpr size_t kMaxNodeLdsBytes = 48 * 1024;   // node arrays up to this size are staged in LDS (1536 nodes)

}  // namespace rtk

#include "trace.hip.hpp"

namespace rtk {
namespace dev {

// Which rank renders which bucket (tile/bucket.hpp:7-21's row-major bucket list dealt to GPUs instead of threads).
// Round robin: bucket i -> rank i % world, the i / world-th bucket of that rank.  When a row of buckets is a whole number of
// rounds (tiles_x % world == 0: 3840 / 24 = 160 buckets per row on 8 ranks) round robin hands every rank the same COLUMNS of
// every row, and a rank's share is as uneven as the picture is from left to right (config 5's shape: 32.8 M ... 38.9 M rays per
// rank).  Those frames are dealt diagonally instead: bucket (bx, by) -> rank (bx + by) % world; with q = tiles_x / world buckets
// per rank and row it is that rank's (by * q + bx / world)-th.  skew_q = q selects it (0 = round robin).  Host mirrors:
// api.hip frame_geom / rank_bucket, parallel.py BucketLayout.
__host__ __device__ inline uint32_t rank_bucket(uint32_t rank, uint32_t local, uint32_t world, uint32_t skew_q) {
    if (skew_q == 0u) return rank + local * world;
    const uint32_t by = local / skew_q, m = local % skew_q;
    const uint32_t c0 = (rank + world - by % world) % world;
    return by * (skew_q * world) + c0 + m * world;
}
__host__ __device__ inline void bucket_owner(uint32_t bucket, uint32_t tiles_x, uint32_t world, uint32_t skew_q, uint32_t &rank, uint32_t &local) {
    if (skew_q == 0u) { rank = bucket % world; local = bucket / world; return; }
    const uint32_t bx = bucket % tiles_x, by = bucket / tiles_x;
    rank = (bx + by) % world; local = by * skew_q + bx / world;
}

struct IntersectArgs {
    TreeView tree;
    const rtk_ray *rays;
    rtk_hit *out;
    size_t n;
    const uint32_t *perm;             // lane i takes ray perm[i] and writes hit perm[i] (repack.hip); null = identity
    uint32_t raster_w;                // != 0 (perm == null): the batch is rows of raster_w rays; waves take 8x8 blocks of it (k_raster_probe)
    const uint32_t *verdict;          // non-null: a launch made BEFORE the host knows the probe's verdict (repack.hpp words 15-17, on the
                                      // device): it does nothing if the batch is to be sorted, and takes the raster width from there
    int cull;
    unsigned long long *counters;
};

struct RenderArgs {
    TreeView tree;
    const DevMaterial *materials;
    const DevTexture *textures;       // procedural textures (null / unused when the scene has none)
    const DevTriUv *tri_uv;           // per-triangle uvs, only when the scene has textures
    const uint8_t *tex_pixels;        // RGB bytes of the bitmap textures (DevTexture::bmp), null without them
    const DevLight *lights;
    int n_lights;
    int has_refractive;
    float cam_pos[3];
    float cam_mat[9];
    float background[3];
    uint32_t width, height;
    float aspect;
    float tan_half_fov;               // tanf(float(radians) / 2.0f), render.hpp:55-57
    int spp, max_depth, diffuse_rays;
    int sample_begin, sample_end;     // this pass renders samples [sample_begin, sample_end) of every pixel (progressive accumulation)
    uint32_t seed;
    float shadow_bias, reflection_bias, refraction_bias;
    // bucket sharding
    uint32_t bucket, tiles_x, tiles_y, n_buckets, blocks_per_bucket_side;
    uint32_t buckets_per_rank;
    uint32_t skew_q;                  // rank_bucket(): 0 = buckets dealt round robin, else diagonally with tiles_x / world per rank and row
    int rank, world;
    uint32_t slice_min_tris;          // GROUP modes: smallest leaf that is split across the workgroup's waves
    int compact;                      // 1: out is [buckets_per_rank][bucket][bucket][3]; 0: out is [h][w][3]
    float *out;
    unsigned long long *counters;
    // two-pass frames (RTK_TRACE_TWOPASS): k_primary fills these, k_tile_order sorts, the primed k_render consumes
    float4 *prim;                     // [output pixels] primary-ray candidate t,u,v,k
    uint32_t *bin_count;              // [kCostBins] pixel blocks per cost bin
    uint32_t *bin_list;               // [kCostBins][tile_cap] pixel-block ids per bin
    uint32_t *tile_order;             // [tile_cap] pixel blocks with work, most expensive first
    uint32_t *n_listed;               // number of entries in tile_order
    uint32_t tile_cap;
    const uint32_t *only_if;          // non-null: the kernel does nothing unless this word is non-zero (overflow fallback)
    // frame-to-frame scheduling feedback (api.hip "cost feedback"): every pixel block reports what it cost; the next
    // frame of the same shape starts its blocks most-expensive-first (results do not depend on the order)
    uint32_t n_units;                 // pixel blocks (8x8) of this rank = buckets_per_rank * blocks_per_bucket_side^2
    uint32_t *cost_out;               // [n_units] or null
    float width_f, height_f, spp_f, gi_div_f;   // (float)width, height, spp, diffuse_rays + 1: wave-uniform, converted once on the host
                                      // (converted in the kernel they sit in vector registers and are spilled per lane)
    const uint32_t *order_in;         // permutation of the pixel blocks, or null = natural order
    const uint32_t *order_hdr;        // {n_workgroups, n}: see launch_order_by_cost
    const uint32_t *wg_list;          // [n_workgroups] a block id, or 0x80000000 | index into order_in of the first block of a packed workgroup
    int shadow_exit;                  // occlusion queries may stop at the first answering hit (no transmissive material; trace())
    // RTK_TRAVERSAL_FAST on a scene with transmissive materials: `occl` is the tree without the transmissive triangles, and an occlusion
    // query of the streaming pipeline (k_shadow) is ONE any-hit query against it instead of is_occluded's stepping loop (rtk.h)
    int occl_on;
    TreeView occl;

    __device__ __forceinline__ size_t out_index(uint32_t local_bucket, uint32_t lx, uint32_t ly, uint32_t px,
                                                uint32_t py) const {
        return compact ? (((size_t)local_bucket * bucket + ly) * bucket + lx) : ((size_t)py * width + px);
    }
};

struct AssembleArgs {
    const float *gathered;
    float *rgb;
    uint32_t width, height, bucket, tiles_x, world, buckets_per_rank, skew_q;
};

}  // namespace dev

hipError_t launch_intersect(const dev::IntersectArgs &A, int mode, bool stats, hipStream_t s);
// n_workgroups (GROUP modes with a workgroup list): how many workgroups the list holds, when the host knows; 0 = one per pixel block
hipError_t launch_render(const dev::RenderArgs &A, int mode, bool stats, bool forks, hipStream_t s, unsigned n_workgroups = 0);
// order[0..n) = the pixel blocks sorted by cost[], most expensive first (one workgroup, counting sort over 256 log-scale bins;
// blocks cheaper than floor_below are not ordered among themselves).  Blocks costlier than light_below (units of 16 cycles;
// 0 = all) get a workgroup each, the rest are packed `pack` to a workgroup; wg_list[0..hdr[0]) = the workgroups by expected
// duration, longest first (see RenderArgs::wg_list); hdr[1] = n
hipError_t launch_order_by_cost(const uint32_t *cost, uint8_t *bins /* [n] scratch */, uint32_t *order, uint32_t *wg_list /* [n] */,
                                uint32_t *hdr /* [2] */, uint32_t n, uint32_t light_below, uint32_t floor_below, uint32_t pack,
                                hipStream_t s);
// first frame of a shape: the launch list (order, wg_list, hdr as launch_order_by_cost leaves them) from the camera rays alone:
// blocks whose centre ray hits a reflective / refractive surface, then blocks that enter the tree, then background blocks `pack`
// to a workgroup.  cls [n_units] and scratch [6] are work space.
hipError_t launch_block_prior(const dev::RenderArgs &A, uint8_t *cls, uint32_t *order, uint32_t *wg_list, uint32_t *hdr,
                              uint32_t *scratch, uint32_t pack, hipStream_t s);
hipError_t launch_twopass(const dev::RenderArgs &A, bool stats, bool forks, hipStream_t s);
hipError_t launch_assemble(const dev::AssembleArgs &A, hipStream_t s);
hipError_t launch_camera_rays(const dev::RenderArgs &A, int sample, rtk_ray *d_rays, hipStream_t s);
hipError_t launch_to_rgb8(const float *d_rgb, size_t n, uint8_t *d_out, hipStream_t s);

}  // namespace rtk
