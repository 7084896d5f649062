// Argument blocks of the streaming ("wavefront") frame pipeline (stream.hip).
#pragma once

#include "kernels.hpp"

namespace rtk {
namespace dev {

// One node of a sample's ray tree = one ray (camera ray or secondary ray).  48 bytes, three 16-byte slots.
struct RayRec {
    float o[3];
    uint32_t parent;        // node id of the ray that spawned this one (unused for camera rays)
    float d[3];
    uint32_t pixel;         // output index (in pixels) of the sample this ray belongs to
    uint32_t key;           // RNG key (position in the ray tree, common.hip.hpp)
    uint32_t info;          // bit 0: valid, bit 1: a miss is worth the background colour (else black)
    uint32_t pad[2];
};
static_assert(sizeof(RayRec) == 48, "RayRec is three 16-byte slots");
constexpr uint32_t kRayValid = 1u, kRayMissBackground = 2u;

// What color_hit made of the ray (render/render.hpp:133-308).  32 bytes.
struct NodeRes {
    float value[3];         // the ray's colour once known (leaves: at once; inner nodes: after k_combine)
    uint32_t kind;          // NODE_*
    uint32_t first_child;   // children are contiguous node ids
    uint32_t aux;           // NODE_REFR: fresnel (float bits); NODE_DIFF: index of the shading point
    uint32_t n_children;
    uint32_t pad;
};
static_assert(sizeof(NodeRes) == 32, "NodeRes is two 16-byte slots");
enum : uint32_t {
    NODE_LEAF = 0,          // value is final: miss colour, depth limit, constant material
    NODE_PASS = 1,          // reflective / total internal reflection: the value of the single child
    NODE_REFR = 2,          // refractive: fresnel * child[1] + (1 - fresnel) * child[0]   (child 0 = refraction ray)
    NODE_DIFF = 3,          // diffuse: (sum of GI children, then unoccluded lights in order) / (diffuse_rays + 1)
    NODE_TEX = 4            // texture material: unoccluded lights in order times the sampled colour kept in `value`
};

struct HitRec {             // 32 B: a diffuse shading point waiting for its light loop
    float P[3];
    uint32_t node;
    float ncos[3];          // the normal the cosine law uses (hit_normal if smooth_shading, else face_normal)
    uint32_t mat;
};
static_assert(sizeof(HitRec) == 32, "HitRec is two 16-byte stores");

// control block (u32 words)
constexpr uint32_t kLevels = kMaxRayDepth + 2;
constexpr uint32_t kCtrlNodeCount = 0;                      // [kLevels] ray-tree nodes per depth level
constexpr uint32_t kCtrlHitCount = kLevels;                 // [kLevels] shading points per depth level
constexpr uint32_t kCtrlTicket = 2 * kLevels;               // [2][kLevels] dynamic work-unit tickets (path, shadow)
constexpr uint32_t kCtrlOverflow = 4 * kLevels;             // [1] set when a queue ran out of space: the frame is redone by the megakernel
constexpr uint32_t kCtrlWords = 4 * kLevels + 4;

// Coherence sort (deep levels of forking ray trees): rays are binned by direction octant (major) and a 16^3 grid cell
// of their origin inside the scene box; shading points by the cell of their position.  Counting sort, all sizes on
// the device.  Only the ORDER in which 64-ray work units are formed changes — never a result.
constexpr uint32_t kSortBins = 8u * 4096u;

struct StreamWs {
    RayRec *rays;           // [node_cap]
    NodeRes *nodes;         // [node_cap]
    HitRec *hits;           // [hit_cap]
    float2 *contrib;        // [hit_cap * n_lights] {contribution, unoccluded}
    float *sumbuf;          // [pixels * 3] running sample sum (spp > 1 only)
    uint32_t *ctrl;
    uint32_t node_cap, hit_cap;
    uint32_t *node_bins;    // [kSortBins] histogram -> offsets -> cursors of the next level's rays
    uint32_t *hit_bins;     // [kSortBins] same for this level's shading points
    uint32_t *node_order;   // [node_cap] node ids of the current level in sorted order
    uint32_t *hit_order;    // [hit_cap] hit ids of the current level in sorted order
};

// Samples of a frame are independent ray trees.  A BATCH of consecutive samples is traced together, level by level, through one
// set of queues (one k_path / k_shadow / k_combine launch per level for the whole batch: fewer launches and level tails, and
// the coherence sort has several samples' rays to make its 64-ray units from); the depth-0 k_combine adds a pixel's samples
// in sample order.  Up to kStreamLanes batches are in flight at once, each on its own HIP stream with its own queues, so that
// the tail of one batch's level overlaps the other batches' work; only the depth-0 k_combine of a batch is ordered behind the
// previous batch's, by an event: the running pixel sums stay in sample order.  HBM is what pays: ~120 B per ray-tree node.
constexpr int kStreamLanes = 8;          // upper bound; the default is 4 (api.hip rtk_knobs)

struct StreamArgs {
    RenderArgs r;
    StreamWs ws;
    const uint32_t *lane_overflow[kStreamLanes];   // the overflow words of all lanes in use (k_combine at depth 0 emits nothing if any is set)
    uint32_t n_lanes;
    uint32_t level;
    int sample;             // first sample of this launch's batch
    uint32_t n_batch;       // samples traced together: the ray trees of samples [sample, sample + n_batch) share the queues
    uint32_t n_root;        // level-0 nodes of ONE sample: 64 per 8x8 pixel block of this rank
    uint32_t n_level0;      // = n_root * n_batch: sample b's camera ray of pixel slot i is node b * n_root + i
    uint32_t auto_min_lanes; // RTK_TRACE_AUTO: leave the wave-cooperative walk when fewer rays than this share a node
    uint32_t nodes_sorted;   // k_path: this level's nodes are taken through ws.node_order
    uint32_t hits_sorted;    // k_shadow: this level's shading points are taken through ws.hit_order
    uint32_t bin_children;   // k_path: histogram the rays it spawns (the next level will be sorted)
    uint32_t bin_hits;       // k_path: histogram the shading points it appends
    uint32_t key_dirs;       // ray_sort_key: 1 = spend six of the fifteen key bits on the direction (frames with diffuse rays)
    float grid_lo[3], grid_scale[3];   // cell = clamp((p - lo) * scale, 0, 15)
};

}  // namespace dev

// Two side streams per sample lane for the k_shadow kernels (even / odd depth levels), with the events that order them.
struct StreamSide {
    hipStream_t stream[2];
    hipEvent_t ready[2];     // recorded on the lane's stream when a level's shading points are sorted: k_shadow may start
    hipEvent_t done[2];      // recorded on the side stream behind its last k_shadow of the sample
};
// `wait_before_emit` (may be null): the depth-0 k_combine waits for it (the previous sample's `done`); `done` (may be null) is
// recorded behind it.  `side` (may be null): run the k_shadow kernels there.
// `slices`: waves per 64-ray work unit of the queue-driven levels (4: an owner and three helpers that share its large leaves; 2; 1: no helpers)
hipError_t launch_stream_sample(const dev::StreamArgs &base, bool stats, int deep_level, int deep_mode, int sort_from_level,
                                hipStream_t s, hipEvent_t wait_before_emit, hipEvent_t done, const StreamSide *side, int slices = 4);
// ORs the lanes' overflow words into lane 0's, then zeroes the ray counters if it is set (the megakernel that redoes the frame counts from scratch)
hipError_t launch_stream_overflow_reset(const dev::StreamArgs &S, hipStream_t s);

}  // namespace rtk
