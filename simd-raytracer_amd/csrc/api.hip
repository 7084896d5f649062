// C-ABI of the rtk engine (include/rtk.h).  Owns device buffers, validates operands against what the
// kernels and their grids assume, and never lets an exception cross the boundary.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "kernels.hpp"
#include "stream.hpp"
#include "repack.hpp"

namespace rtk {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

namespace {

int fail(int code, const std::string &msg) { set_error(msg); return code; }

int hip_fail(hipError_t e, const char *what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return RTK_ERR_HIP;
}

#define RTK_HIP(call)                                        \
    do {                                                     \
        const hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #call);    \
    } while (0)

template <typename T>
int upload(const std::vector<T> &src, T **dst) {
    *dst = nullptr;
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(dst), bytes));
    if (!src.empty()) RTK_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RTK_OK;
}

}  // namespace
}  // namespace rtk

// Environment knobs (all optional).  Read ONCE, when an accel is built: nothing on the launch path
// calls getenv.
struct rtk_knobs {
    uint32_t slice_min_tris = rtk::kSliceMinTrisDefault;   // RTK_SLICE_MIN_TRIS
    bool shadow_exit = true;                                // RTK_SHADOW_EARLY_EXIT
    bool bundle_cull = true;                                // RTK_BUNDLE_CULL
    bool auto_trials = true;                                // RTK_AUTO_TRIALS
    bool cost_feedback = true;                              // RTK_COST_FEEDBACK
    unsigned resort_every = 16;                             // RTK_COST_RESORT_EVERY
    uint32_t light_cycles = 140000u;                        // RTK_LIGHT_BELOW_CYCLES
    uint32_t order_floor_cycles = 20000u;                   // RTK_ORDER_FLOOR_CYCLES
    bool batch_scalar_surv = false;                         // RTK_BATCH_SCALAR_SURV: the same in the batched intersect
    bool stream_scalar_surv = true;                         // RTK_STREAM_SCALAR_SURV: survivors through the scalar cache in the streaming kernels
    bool repack = true;                                     // RTK_REPACK: RTK_TRACE_AUTO may sort large incoherent ray batches
    bool raster_tiles = true;                               // RTK_RASTER_TILES: coherent batches that are rows of camera rays go to the waves as 8x8 blocks
    int repack_skip_bits = 6;                               // RTK_REPACK_SKIP_BITS: low key bits left unsorted when <= 3 dimensions vary (0..14)
    bool repack_full_bounds = false;                        // RTK_REPACK_FULL_BOUNDS: key cells from the bounds of all rays, not of a sample
    int repack_skip_bits2 = 14;                             // RTK_REPACK_SKIP_BITS2: the same when <= 2 dimensions vary (0..22)
    bool repack_dirs3 = false;                              // RTK_REPACK_DIRS3: directions enter the sort keys as three components even when all rays share an origin
    int repack_trace = -1;                                  // RTK_REPACK_TRACE: strategy for a sorted batch (0 auto, 1 lane, 2 wave; default: by the probe)
    size_t group8_below = 9000;                             // RTK_GROUP8_BELOW_BLOCKS
    int stream_node_factor = 0;                             // RTK_STREAM_NODE_FACTOR (0 = default)
    int stream_deep_level = 99, stream_deep_mode = RTK_TRACE_AUTO;   // RTK_STREAM_DEEP_LEVEL / _MODE
    uint32_t auto_min_lanes = 12;                           // RTK_AUTO_MIN_LANES
    int stream_sort_from = -1;                              // RTK_STREAM_SORT_FROM (-1 = default)
    bool stream_debug = false;                              // RTK_STREAM_DEBUG
    bool stream_side = true;                                // RTK_STREAM_SIDE: k_shadow on side streams
    int stream_lanes = 4;                                   // RTK_STREAM_LANES: batches of a frame in flight at once (1..kStreamLanes); 8 measured no faster
    int stream_slices = 0;                                  // RTK_STREAM_SLICES: waves per work unit of the streaming levels (1, 2, 4; 0 = by the tree's leaf sizes)
    int stream_batch = 0;                                   // RTK_STREAM_BATCH: samples traced together per launch (stream.hpp; 0 = the pass split evenly over the lanes)
    int stream_mem_gb = 96;                                 // RTK_STREAM_MEM_GB: budget for the queues of all batches in flight
    int stream_side_below = 2;                              // RTK_STREAM_SIDE_BELOW: k_shadow on side streams while at most this many samples are in flight
    bool first_frame_prior = true;                          // RTK_FIRST_FRAME_PRIOR: launch order of a shape's first frame from k_block_prior
    bool fast_occluders = true;                             // RTK_FAST_OCCLUDERS: RTK_TRAVERSAL_FAST answers occlusion through transmissive surfaces from the opaque triangles alone
    bool traversal_fast = false;                            // RTK_TRAVERSAL_FAST: front-to-back leaf order (rtk.h; NOT the parity mode)

    static rtk_knobs from_env() {
        rtk_knobs k;
        auto geti = [](const char *name, long &out) { const char *e = std::getenv(name); if (!e || !*e) return false; out = std::atol(e); return true; };
        long v;
        if (geti("RTK_SLICE_MIN_TRIS", v) && v > 0) k.slice_min_tris = uint32_t(v);
        if (geti("RTK_SHADOW_EARLY_EXIT", v)) k.shadow_exit = v != 0;
        if (geti("RTK_BUNDLE_CULL", v)) k.bundle_cull = v != 0;
        if (geti("RTK_AUTO_TRIALS", v)) k.auto_trials = v != 0;
        if (geti("RTK_COST_FEEDBACK", v)) k.cost_feedback = v != 0;
        if (geti("RTK_COST_RESORT_EVERY", v) && v > 0) k.resort_every = unsigned(v);
        if (geti("RTK_LIGHT_BELOW_CYCLES", v) && v >= 0) k.light_cycles = uint32_t(v);
        if (geti("RTK_ORDER_FLOOR_CYCLES", v) && v >= 0) k.order_floor_cycles = uint32_t(v);
        if (geti("RTK_REPACK", v)) k.repack = v != 0;
        if (geti("RTK_RASTER_TILES", v)) k.raster_tiles = v != 0;
        if (geti("RTK_REPACK_FULL_BOUNDS", v)) k.repack_full_bounds = v != 0;
        if (geti("RTK_REPACK_SKIP_BITS2", v) && v >= 0 && v <= 22) k.repack_skip_bits2 = int(v);
        if (geti("RTK_REPACK_DIRS3", v)) k.repack_dirs3 = v != 0;
        if (geti("RTK_REPACK_SKIP_BITS", v) && v >= 0 && v <= 14) k.repack_skip_bits = int(v);
        if (geti("RTK_STREAM_SCALAR_SURV", v)) k.stream_scalar_surv = v != 0;
        if (geti("RTK_BATCH_SCALAR_SURV", v)) k.batch_scalar_surv = v != 0;
        if (geti("RTK_REPACK_TRACE", v) && (v == RTK_TRACE_AUTO || v == RTK_TRACE_WAVE || v == RTK_TRACE_LANE)) k.repack_trace = int(v);
        if (geti("RTK_GROUP8_BELOW_BLOCKS", v) && v >= 0) k.group8_below = size_t(v);
        if (geti("RTK_STREAM_NODE_FACTOR", v) && v >= 1) k.stream_node_factor = int(v);
        if (geti("RTK_STREAM_DEEP_LEVEL", v)) k.stream_deep_level = int(v);
        if (geti("RTK_STREAM_DEEP_MODE", v)) k.stream_deep_mode = int(v);
        if (geti("RTK_AUTO_MIN_LANES", v) && v > 0 && v <= 64) k.auto_min_lanes = uint32_t(v);
        if (geti("RTK_STREAM_SORT_FROM", v)) k.stream_sort_from = int(v);
        if (geti("RTK_STREAM_DEBUG", v)) k.stream_debug = v != 0;
        if (geti("RTK_STREAM_SIDE", v)) k.stream_side = v != 0;
        if (geti("RTK_TRAVERSAL_FAST", v)) k.traversal_fast = v != 0;
        if (geti("RTK_FIRST_FRAME_PRIOR", v)) k.first_frame_prior = v != 0;
        if (geti("RTK_STREAM_SLICES", v) && (v == 0 || v == 1 || v == 2 || v == 4)) k.stream_slices = int(v);
        if (geti("RTK_STREAM_BATCH", v) && v >= 0 && v <= 4096) k.stream_batch = int(v);
        if (geti("RTK_STREAM_MEM_GB", v) && v >= 1 && v <= 256) k.stream_mem_gb = int(v);
        if (geti("RTK_FAST_OCCLUDERS", v)) k.fast_occluders = v != 0;
        if (geti("RTK_STREAM_SIDE_BELOW", v) && v >= 0) k.stream_side_below = int(v);
        if (geti("RTK_STREAM_LANES", v) && v >= 1 && v <= rtk::dev::kStreamLanes) k.stream_lanes = int(v);
        return k;
    }
};

struct rtk_accel {
    rtk_knobs knobs;
    bool coords_small = false;        // every leaf-reference coordinate is below kBundleLimit: bundle culling cannot overflow
    rtk_scene scene;                  // private copy: the caller may free its scene (kd_tree_simd.hpp:106-107 copies too)
    rtk::HostTree tree;
    rtk_accel_params params;
    bool has_refractive = false;
    // device residency (lazy: built on first compute call so host-only use needs no GPU)
    bool on_device = false;
    int device = -1;
    rtk::DevNode *d_nodes = nullptr;
    rtk::DevNode *d_leaves = nullptr;
    // RTK_TRAVERSAL_FAST on a scene with transmissive materials: the tree again with the opaque triangles only (occlusion queries, k_shadow)
    rtk::DevNode *d_occl_nodes = nullptr, *d_occl_leaves = nullptr;
    rtk::DevTri *d_occl_tris = nullptr;
    uint32_t *d_occl_ids = nullptr;
    uint32_t occl_n_leaves = 0;
    bool occl_on = false;
    rtk::DevNode *d_leaves_fast = nullptr;    // RTK_TRAVERSAL_FAST: 8 front-to-back orders of the leaves (null in the parity mode)
    bool fast_traversal = false;
    // Streaming pipeline: waves per 64-ray work unit.  Helper waves pay where a ray meets large leaves (hw11/scene8, a
    // triangle reference sits in a leaf of 258 on average: 23.3 ms with three helpers, 36.6 without) and cost where it does not
    // (hw15/scene2, 109: 64.1 ms with, 47.8 without -- the helpers' wave slots are worth more as owners of further units).
    int stream_slices_auto = 4;
    rtk::DevTri *d_tris = nullptr;
    uint32_t *d_tri_ids = nullptr;
    rtk::DevShade *d_shade = nullptr;
    rtk::DevMaterial *d_materials = nullptr;
    rtk::DevLight *d_lights = nullptr;
    rtk::DevTexture *d_textures = nullptr;
    rtk::DevTriUv *d_tri_uv = nullptr;
    uint8_t *d_tex_pixels = nullptr;
    unsigned long long *d_counters = nullptr;     // 8 x u64 in rtk_counters order + kRayCounterShards ray-count shards
    // streaming-pipeline workspace (grown on demand)
    // one per sample lane (stream.hpp kStreamLanes); `ws` = lane 0; all lanes share lane 0's sumbuf
    rtk::dev::StreamWs ws = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr, nullptr, nullptr, nullptr};
    rtk::dev::StreamWs ws_lane[rtk::dev::kStreamLanes] = {};
    int ws_lanes = 0;
    hipStream_t lane_stream[rtk::dev::kStreamLanes] = {};
    hipEvent_t lane_done[rtk::dev::kStreamLanes] = {};
    hipEvent_t lane_fork = nullptr;
    rtk::StreamSide lane_side[rtk::dev::kStreamLanes] = {};     // k_shadow side streams of every lane
    size_t ws_pixels = 0, ws_lights = 0, ws_nodes = 0;
    bool ws_sum = false;
    // two-pass workspace
    float4 *tp_prim = nullptr;
    uint32_t *tp_bins = nullptr;       // [kCostBins] counts, [1] n_listed
    uint32_t *tp_bin_list = nullptr, *tp_order = nullptr;
    size_t tp_pixels = 0, tp_tiles = 0;
    // cost feedback (megakernel frames): per-pixel-block cost of the last frame of shape fb_sig, and the order made from it
    uint32_t *fb_cost = nullptr, *fb_order = nullptr;
    uint8_t *fb_bins = nullptr;
    size_t fb_units = 0;
    uint64_t fb_sig[4] = {0, 0, 0, 0};
    // RTK_TRACE_AUTO on forking scenes: which engine is faster for the current shape (render_device_impl)
    hipEvent_t trial_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t trial_sig[3] = {0, 0, 0};
    int trial_state = 0;
    // number of workgroups in fb_order's workgroup list, read back behind the sort that made it (pinned host word + event)
    uint32_t *fb_nwgs_host = nullptr;
    hipEvent_t fb_nwgs_ev = nullptr;
    bool fb_nwgs_pending = false, fb_nwgs_known = false;
    bool fb_valid = false;           // fb_cost holds the costs of a frame of shape fb_sig
    bool fb_order_valid = false;     // fb_order was made from such costs
    // ray repacking workspace (batched intersect, repack.hip)
    uint32_t *rp_bounds = nullptr, *rp_keys = nullptr, *rp_idx = nullptr;
    void *rp_temp = nullptr;
    uint32_t *rp_host = nullptr;     // pinned: the probe's words as the host sees them
    hipEvent_t rp_probe_ev = nullptr;
    hipEvent_t rp_done = nullptr;    // recorded behind the k_intersect that walks rp_idx: the next repack on any stream waits for it
    bool rp_in_use = false;
    size_t rp_temp_bytes = 0, rp_cap = 0;
    unsigned fb_age = 0;             // frames rendered with the current order
    hipStream_t last_stream = nullptr;
    uint64_t last_primary = 0;
    bool last_stats = false;
    std::mutex mu;
};

namespace rtk {
namespace {

int ensure_device(rtk_accel *a) {
    if (a->on_device) {
        RTK_HIP(hipSetDevice(a->device));
        return RTK_OK;
    }
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(RTK_ERR_NO_DEVICE, "no usable HIP device (this engine has no CPU path)");
    int dev = a->params.device;
    if (dev < 0) RTK_HIP(hipGetDevice(&dev));
    if (dev >= count) return fail(RTK_ERR_INVALID, "device ordinal out of range");
    RTK_HIP(hipSetDevice(dev));
    a->device = dev;
    int rc;
    if ((rc = upload(a->tree.dev_nodes, &a->d_nodes)) != RTK_OK) return rc;
    if ((rc = upload(a->tree.dev_leaves, &a->d_leaves)) != RTK_OK) return rc;
    if (a->fast_traversal && (rc = upload(a->tree.dev_leaves_fast, &a->d_leaves_fast)) != RTK_OK) return rc;
    if ((rc = upload(a->tree.dev_tris, &a->d_tris)) != RTK_OK) return rc;
    if ((rc = upload(a->tree.dev_tri_ids, &a->d_tri_ids)) != RTK_OK) return rc;
    if ((rc = upload(a->tree.dev_shade, &a->d_shade)) != RTK_OK) return rc;
    if (a->fast_traversal && a->has_refractive && a->knobs.fast_occluders) {
        // occlusion through transmissive surfaces as ONE query against what is not transmissive (rtk.h, RTK_TRAVERSAL_FAST): the same
        // nodes, their leaves without the transmissive triangles; leaves left empty drop out of the leaf list
        const HostTree &t = a->tree;
        std::vector<DevNode> nodes = t.dev_nodes, leaves;
        std::vector<DevTri> tris;
        std::vector<uint32_t> ids;
        for (DevNode &n : nodes) {
            if (n.b == DEV_INNER) continue;
            const uint32_t first = uint32_t(tris.size());
            for (uint32_t r = n.a; r < n.a + n.b; ++r) {
                const uint32_t m = t.dev_shade[t.dev_tri_ids[r]].material;
                if (m < a->scene.materials.size() && a->scene.materials[m].kind == RTK_MAT_REFRACTIVE) continue;
                tris.push_back(t.dev_tris[r]); ids.push_back(t.dev_tri_ids[r]);
            }
            n.a = first; n.b = uint32_t(tris.size()) - first;
            if (n.b != 0u) leaves.push_back(n);
        }
        if (tris.empty()) { tris.push_back(t.dev_tris.empty() ? DevTri{} : t.dev_tris[0]); ids.push_back(0u); }      // (nothing opaque: keep the pointers valid)
        if (leaves.empty()) { DevNode n = nodes.empty() ? DevNode{} : nodes[0]; n.a = 0u; n.b = 0u; leaves.push_back(n); }
        if ((rc = upload(nodes, &a->d_occl_nodes)) != RTK_OK) return rc;
        if ((rc = upload(leaves, &a->d_occl_leaves)) != RTK_OK) return rc;
        if ((rc = upload(tris, &a->d_occl_tris)) != RTK_OK) return rc;
        if ((rc = upload(ids, &a->d_occl_ids)) != RTK_OK) return rc;
        a->occl_n_leaves = uint32_t(leaves.size());
        a->occl_on = true;
    }
    if ((rc = upload(a->scene.materials, &a->d_materials)) != RTK_OK) return rc;
    if ((rc = upload(a->scene.lights, &a->d_lights)) != RTK_OK) return rc;
    if (!a->scene.textures.empty()) {
        if ((rc = upload(a->scene.textures, &a->d_textures)) != RTK_OK) return rc;
        if ((rc = upload(a->tree.dev_tri_uv, &a->d_tri_uv)) != RTK_OK) return rc;
        if (!a->scene.tex_pixels.empty() && (rc = upload(a->scene.tex_pixels, &a->d_tex_pixels)) != RTK_OK) return rc;
    }
    // (+ 4 words behind the counters: the six cursors of the first-frame prior, zeroed by the same fill as the counters)
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->d_counters), (kCounterWords + 4) * sizeof(unsigned long long)));
    RTK_HIP(hipMemset(a->d_counters, 0, (kCounterWords + 4) * sizeof(unsigned long long)));
    a->on_device = true;
    return RTK_OK;
}

dev::TreeView tree_view(const rtk_accel *a) {
    dev::TreeView t;
    t.nodes = a->d_nodes; t.tris = a->d_tris; t.tri_ids = a->d_tri_ids; t.shade = a->d_shade;
    t.leaves = a->d_leaves; t.n_leaves = static_cast<uint32_t>(a->tree.dev_leaves.size());
    t.leaves_fast = a->fast_traversal ? a->d_leaves_fast : nullptr;
    t.n_nodes = static_cast<uint32_t>(a->tree.dev_nodes.size());
    t.eps = a->params.eps;
    t.normalize = a->params.normalize_hit_normal;
    t.bundle_cull = (a->knobs.bundle_cull && a->coords_small) ? 1 : 0;
    t.scalar_surv = 0;
    return t;
}

// (Re)allocates the streaming workspace for `pixels` output pixels.  Allocation synchronises the device, so it only
// happens when a larger frame (or more lights / multi-sample) is requested than ever before on this accel.
// The streams the streaming pipeline's lanes (and their k_shadow side kernels) run on belong to the PROCESS and are made once
// per device, in the order the first accel needs them.  HIP deals a process's streams to a handful of hardware queues; lanes
// that land on one queue take turns instead of overlapping, and which queue a stream gets depends on how many were made before
// it.  With streams made (and destroyed) per accel, the second accel of a process ran the same kernels on the same rays up to
// 60 % slower (hw11/scene8 23 -> 37 ms, hw15/scene2 47 -> 58 ms per pass; GPU_MAX_HW_QUEUES=2: 72 ms): its lanes shared queues.
struct LaneStreams {
    hipStream_t lane[rtk::dev::kStreamLanes] = {};
    hipStream_t side[rtk::dev::kStreamLanes][2] = {};
};
std::mutex g_lane_mu;
LaneStreams g_lane_streams[16];

hipError_t lane_streams_for(int device, int lanes, LaneStreams **out) {
    if (device < 0 || device >= 16) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_lane_mu);
    LaneStreams &L = g_lane_streams[device];
    for (int j = 0; j < lanes && j < rtk::dev::kStreamLanes; ++j) {
        hipError_t e = hipSuccess;
        if (j > 0 && !L.lane[j]) e = hipStreamCreateWithFlags(&L.lane[j], hipStreamNonBlocking);
        for (int par = 0; par < 2 && e == hipSuccess; ++par)
            if (!L.side[j][par]) e = hipStreamCreateWithFlags(&L.side[j][par], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = &L;
    return hipSuccess;
}

void free_stream_ws(rtk_accel *a) {
    (void)hipFree(a->ws.sumbuf);
    for (int j = 0; j < dev::kStreamLanes; ++j) {
        dev::StreamWs &w = a->ws_lane[j];
        (void)hipFree(w.rays); (void)hipFree(w.nodes); (void)hipFree(w.hits); (void)hipFree(w.contrib); (void)hipFree(w.ctrl);
        (void)hipFree(w.node_bins); (void)hipFree(w.hit_bins); (void)hipFree(w.node_order); (void)hipFree(w.hit_order);
        w = dev::StreamWs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr, nullptr, nullptr, nullptr};
    }
    a->ws = a->ws_lane[0];
    a->ws_lanes = 0; a->ws_pixels = 0; a->ws_nodes = 0; a->ws_lights = 0; a->ws_sum = false;
}

int ensure_stream_ws(rtk_accel *a, size_t pixels, size_t nodes, size_t lights, bool need_sum, int lanes) {
    if (lights == 0) lights = 1;
    if (lanes < 1) lanes = 1;
    if (pixels <= a->ws_pixels && nodes <= a->ws_nodes && lights <= a->ws_lights && (!need_sum || a->ws_sum) && lanes <= a->ws_lanes) return RTK_OK;
    const size_t np = pixels > a->ws_pixels ? pixels : a->ws_pixels;
    const size_t nn = nodes > a->ws_nodes ? nodes : a->ws_nodes;
    const size_t nl = lights > a->ws_lights ? lights : a->ws_lights;
    const int nlanes = lanes > a->ws_lanes ? lanes : a->ws_lanes;
    const bool sum = need_sum || a->ws_sum;
    if (nn > 0xFFFFFFF0ull) return fail(RTK_ERR_INVALID, "frame too large for the streaming pipeline's 32-bit node ids");
    RTK_HIP(hipDeviceSynchronize());
    free_stream_ws(a);
    const size_t nh = nn / 2 + 64;                            // every shading point belongs to a distinct node
    float *sumbuf = nullptr;
    if (sum) RTK_HIP(hipMalloc(reinterpret_cast<void **>(&sumbuf), np * 3 * sizeof(float)));
    for (int j = 0; j < nlanes; ++j) {
        dev::StreamWs &w = a->ws_lane[j];
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.rays), nn * sizeof(dev::RayRec)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.nodes), nn * sizeof(dev::NodeRes)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.hits), nh * sizeof(dev::HitRec)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.contrib), nh * nl * sizeof(float2)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.ctrl), dev::kCtrlWords * sizeof(uint32_t)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.node_bins), dev::kSortBins * sizeof(uint32_t)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.hit_bins), dev::kSortBins * sizeof(uint32_t)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.node_order), nn * sizeof(uint32_t)));
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&w.hit_order), nh * sizeof(uint32_t)));
        w.sumbuf = sumbuf;
        w.node_cap = uint32_t(nn); w.hit_cap = uint32_t(nh);
        if (!a->lane_done[j]) RTK_HIP(hipEventCreateWithFlags(&a->lane_done[j], hipEventDisableTiming));
        for (int par = 0; par < 2; ++par) {
            if (!a->lane_side[j].ready[par]) RTK_HIP(hipEventCreateWithFlags(&a->lane_side[j].ready[par], hipEventDisableTiming));
            if (!a->lane_side[j].done[par]) RTK_HIP(hipEventCreateWithFlags(&a->lane_side[j].done[par], hipEventDisableTiming));
        }
    }
    if (!a->lane_fork) RTK_HIP(hipEventCreateWithFlags(&a->lane_fork, hipEventDisableTiming));
    {   // the process's lane streams (see LaneStreams); the events that order them stay this accel's own
        LaneStreams *L = nullptr;
        RTK_HIP(lane_streams_for(a->device, nlanes, &L));
        for (int j = 0; j < nlanes; ++j) {
            a->lane_stream[j] = L->lane[j];
            a->lane_side[j].stream[0] = L->side[j][0]; a->lane_side[j].stream[1] = L->side[j][1];
        }
    }
    a->ws = a->ws_lane[0];
    a->ws_lanes = nlanes; a->ws_pixels = np; a->ws_nodes = nn; a->ws_lights = nl; a->ws_sum = sum;
    return RTK_OK;
}

int ensure_twopass_ws(rtk_accel *a, size_t pixels, size_t tiles) {
    if (pixels <= a->tp_pixels && tiles <= a->tp_tiles) return RTK_OK;
    const size_t np = pixels > a->tp_pixels ? pixels : a->tp_pixels, nt = tiles > a->tp_tiles ? tiles : a->tp_tiles;
    RTK_HIP(hipDeviceSynchronize());
    (void)hipFree(a->tp_prim); (void)hipFree(a->tp_bins); (void)hipFree(a->tp_bin_list); (void)hipFree(a->tp_order);
    a->tp_prim = nullptr; a->tp_bins = nullptr; a->tp_bin_list = nullptr; a->tp_order = nullptr;
    a->tp_pixels = 0; a->tp_tiles = 0;
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->tp_prim), np * sizeof(float4)));
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->tp_bins), (kCostBins + 1) * sizeof(uint32_t)));
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->tp_bin_list), size_t(kCostBins) * nt * sizeof(uint32_t)));
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->tp_order), nt * sizeof(uint32_t)));
    a->tp_pixels = np; a->tp_tiles = nt;
    return RTK_OK;
}

bool valid_mode(int m) { return m == RTK_TRACE_AUTO || m == RTK_TRACE_LANE || m == RTK_TRACE_WAVE; }
bool valid_batch_mode(int m) { return valid_mode(m) || m == RTK_TRACE_REPACK; }
bool valid_frame_mode(int m) { return valid_mode(m) || m == RTK_TRACE_GROUP4 || m == RTK_TRACE_GROUP8 || m == RTK_TRACE_GROUP16 ||
           m == RTK_TRACE_STREAM || m == RTK_TRACE_TWOPASS; }

struct FrameGeom {
    uint32_t width, height, bucket, tiles_x, tiles_y, n_buckets, blocks_side, buckets_per_rank;
    uint32_t skew_q;                   // kernels.hpp rank_bucket(): 0 = round robin, else tiles_x / world (diagonal deal)
    int rank, world;
    int sample_begin, sample_end;      // this call renders samples [sample_begin, sample_end) (rtk_render_params.sample_begin/_count)
};

int frame_geom(const rtk_accel *a, const rtk_render_params *p, FrameGeom &g) {
    if (!a || !p) return fail(RTK_ERR_INVALID, "null accel or params");
    const int64_t w = p->width > 0 ? p->width : a->scene.width;
    const int64_t h = p->height > 0 ? p->height : a->scene.height;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) return fail(RTK_ERR_INVALID, "image size must be in [1, 65536]");
    if (p->spp < 1) return fail(RTK_ERR_INVALID, "spp must be >= 1");
    if (p->max_ray_depth < 0 || p->max_ray_depth > kMaxRayDepth)
        return fail(RTK_ERR_INVALID, "max_ray_depth must be in [0, 16]");
    if (p->diffuse_rays < 0 || p->diffuse_rays > 32767) return fail(RTK_ERR_INVALID, "diffuse_rays must be in [0, 32767]");
    if (!valid_frame_mode(p->trace_mode)) return fail(RTK_ERR_INVALID, "unknown trace_mode");
    if (p->sample_begin < 0 || p->sample_count < 0 || p->sample_begin >= p->spp ||
        int64_t(p->sample_begin) + p->sample_count > p->spp)
        return fail(RTK_ERR_INVALID, "sample_begin / sample_count must select samples inside [0, spp)");
    if (p->sample_count == 0 && p->sample_begin != 0) return fail(RTK_ERR_INVALID, "sample_count == 0 means all samples: sample_begin must be 0");
    g.sample_begin = p->sample_begin;
    g.sample_end = p->sample_count == 0 ? p->spp : p->sample_begin + p->sample_count;
    g.world = p->world_size > 1 ? p->world_size : 1;
    g.rank = p->world_size > 1 ? p->rank : 0;
    if (g.rank < 0 || g.rank >= g.world) return fail(RTK_ERR_INVALID, "rank must be in [0, world_size)");
    g.width = uint32_t(w); g.height = uint32_t(h);
    g.bucket = uint32_t(a->scene.bucket_size > 0 ? a->scene.bucket_size : 64);
    g.tiles_x = (g.width + g.bucket - 1) / g.bucket;
    g.tiles_y = (g.height + g.bucket - 1) / g.bucket;
    g.n_buckets = g.tiles_x * g.tiles_y;
    g.blocks_side = (g.bucket + 7) / 8;
    g.buckets_per_rank = (g.n_buckets + uint32_t(g.world) - 1) / uint32_t(g.world);
    g.skew_q = (g.world > 1 && g.tiles_x % uint32_t(g.world) == 0u) ? g.tiles_x / uint32_t(g.world) : 0u;
    return RTK_OK;
}

}  // namespace
}  // namespace rtk

using namespace rtk;

extern "C" {

int rtk_abi_version(void) { return RTK_ABI_VERSION; }

const char *rtk_last_error(void) { return g_last_error.c_str(); }

int rtk_device_count(int *count) {
    if (!count) return fail(RTK_ERR_INVALID, "null count");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return RTK_OK;
}

// ---------------------------------------------------------------- scene

int rtk_scene_create(const rtk_scene_desc *desc, rtk_scene **out) {
    if (!desc || !out) return fail(RTK_ERR_INVALID, "null desc or out");
    *out = nullptr;
    try {
        rtk_scene *s = new rtk_scene();
        std::string err;
        const int rc = scene_from_desc(*desc, *s, err);
        if (rc != RTK_OK) { delete s; return fail(rc, err); }
        *out = s;
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_scene_load_crtscene(const char *path, rtk_scene **out) {
    if (!path || !out) return fail(RTK_ERR_INVALID, "null path or out");
    *out = nullptr;
    try {
        rtk_scene *s = new rtk_scene();
        std::string err;
        const int rc = scene_from_crtscene(path, *s, err);
        if (rc != RTK_OK) { delete s; return fail(rc, err); }
        *out = s;
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_scene_get_info(const rtk_scene *s, rtk_scene_info *info) {
    if (!s || !info) return fail(RTK_ERR_INVALID, "null scene or info");
    info->n_meshes = int32_t(s->meshes.size());
    info->n_materials = int32_t(s->materials.size());
    info->n_lights = int32_t(s->lights.size());
    info->n_vertices = s->n_vertices;
    info->n_triangles = s->n_triangles;
    info->width = s->width; info->height = s->height; info->bucket_size = s->bucket_size;
    info->n_textures = int32_t(s->textures.size());
    info->n_uv_vertices = 0;
    for (const HostMesh &m : s->meshes) info->n_uv_vertices += int32_t(m.uvs.size() / 2);
    info->n_bitmap_bytes = int32_t(s->tex_pixels.size());
    return RTK_OK;
}

int rtk_scene_get_arrays(const rtk_scene *s, int32_t *mesh_material, int32_t *mesh_nverts, int32_t *mesh_ntris,
                         float *vertices, uint32_t *indices, int32_t *mat_kind, float *mat_albedo, float *mat_ior,
                         int32_t *mat_smooth, float *light_pos, float *light_intensity, float *cam_pos, float *cam_mat,
                         float *background) {
    if (!s) return fail(RTK_ERR_INVALID, "null scene");
    size_t vo = 0, to = 0;
    for (size_t m = 0; m < s->meshes.size(); ++m) {
        const HostMesh &hm = s->meshes[m];
        if (mesh_material) mesh_material[m] = hm.material;
        if (mesh_nverts) mesh_nverts[m] = int32_t(hm.vertices.size());
        if (mesh_ntris) mesh_ntris[m] = int32_t(hm.indices.size() / 3);
        if (vertices) for (size_t i = 0; i < hm.vertices.size(); ++i) {
            vertices[(vo + i) * 3] = hm.vertices[i].x; vertices[(vo + i) * 3 + 1] = hm.vertices[i].y; vertices[(vo + i) * 3 + 2] = hm.vertices[i].z;
        }
        if (indices && !hm.indices.empty()) std::memcpy(indices + to * 3, hm.indices.data(), hm.indices.size() * sizeof(uint32_t));
        vo += hm.vertices.size(); to += hm.indices.size() / 3;
    }
    for (size_t i = 0; i < s->materials.size(); ++i) {
        if (mat_kind) mat_kind[i] = s->materials[i].kind;
        if (mat_albedo) std::memcpy(mat_albedo + i * 3, s->materials[i].albedo, 3 * sizeof(float));
        if (mat_ior) mat_ior[i] = s->materials[i].ior;
        if (mat_smooth) mat_smooth[i] = s->materials[i].smooth;
    }
    for (size_t i = 0; i < s->lights.size(); ++i) {
        if (light_pos) std::memcpy(light_pos + i * 3, s->lights[i].pos, 3 * sizeof(float));
        if (light_intensity) light_intensity[i] = s->lights[i].intensity;
    }
    if (cam_pos) std::memcpy(cam_pos, s->cam_pos, sizeof(s->cam_pos));
    if (cam_mat) std::memcpy(cam_mat, s->cam_mat, sizeof(s->cam_mat));
    if (background) std::memcpy(background, s->background, sizeof(s->background));
    return RTK_OK;
}

int rtk_scene_get_textures(const rtk_scene *s, int32_t *mat_texture, int32_t *mesh_has_uvs, float *uvs, int32_t *tex_kind,
                           float *tex_color_a, float *tex_color_b, float *tex_param) {
    if (!s) return fail(RTK_ERR_INVALID, "null scene");
    for (size_t i = 0; i < s->materials.size(); ++i) if (mat_texture) mat_texture[i] = s->materials[i].texture;
    size_t uo = 0;
    for (size_t m = 0; m < s->meshes.size(); ++m) {
        const HostMesh &hm = s->meshes[m];
        if (mesh_has_uvs) mesh_has_uvs[m] = hm.uvs.empty() ? 0 : 1;
        if (uvs && !hm.uvs.empty()) std::memcpy(uvs + uo, hm.uvs.data(), hm.uvs.size() * sizeof(float));
        uo += hm.uvs.size();
    }
    for (size_t i = 0; i < s->textures.size(); ++i) {
        if (tex_kind) tex_kind[i] = s->textures[i].kind;
        const bool bmp = s->textures[i].kind == RTK_TEX_BITMAP;
        const float zero[3] = {0.f, 0.f, 0.f};
        if (tex_color_a) std::memcpy(tex_color_a + i * 3, bmp ? zero : s->textures[i].a, 3 * sizeof(float));
        if (tex_color_b) std::memcpy(tex_color_b + i * 3, s->textures[i].b, 3 * sizeof(float));
        if (tex_param) tex_param[i] = s->textures[i].param;
    }
    return RTK_OK;
}

int rtk_scene_get_bitmaps(const rtk_scene *s, int32_t *tex_bitmap, uint8_t *tex_pixels) {
    if (!s) return fail(RTK_ERR_INVALID, "null scene");
    for (size_t i = 0; i < s->textures.size(); ++i) {
        if (!tex_bitmap) break;
        const DevTexture &t = s->textures[i];
        const bool bmp = t.kind == RTK_TEX_BITMAP;
        tex_bitmap[i * 3] = bmp ? t.bmp[2] : 0; tex_bitmap[i * 3 + 1] = bmp ? t.bmp[0] : 0; tex_bitmap[i * 3 + 2] = bmp ? t.bmp[1] : 0;
    }
    if (tex_pixels && !s->tex_pixels.empty()) std::memcpy(tex_pixels, s->tex_pixels.data(), s->tex_pixels.size());
    return RTK_OK;
}

int rtk_decode_jpeg(const uint8_t *data, size_t size, int32_t *width, int32_t *height, int32_t *channels, uint8_t *pixels, size_t cap) {
    if (!data || !width || !height || !channels) return fail(RTK_ERR_INVALID, "null argument");
    try {
        std::vector<uint8_t> px;
        std::string err;
        int w = 0, h = 0, ch = 0;
        const int rc = decode_jpeg(data, size, w, h, ch, px, err);
        if (rc != RTK_OK) return fail(rc, err);
        *width = w; *height = h; *channels = ch;
        if (pixels) std::memcpy(pixels, px.data(), px.size() < cap ? px.size() : cap);
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_scene_vertex_normals(const rtk_scene *s, int32_t mesh, float *out) {
    if (!s || !out) return fail(RTK_ERR_INVALID, "null scene or out");
    if (mesh < 0 || size_t(mesh) >= s->meshes.size()) return fail(RTK_ERR_INVALID, "mesh index out of range");
    const HostMesh &hm = s->meshes[size_t(mesh)];
    for (size_t i = 0; i < hm.vertex_normals.size(); ++i) {
        out[i * 3] = hm.vertex_normals[i].x; out[i * 3 + 1] = hm.vertex_normals[i].y; out[i * 3 + 2] = hm.vertex_normals[i].z;
    }
    return RTK_OK;
}

void rtk_scene_destroy(rtk_scene *s) { delete s; }

// ---------------------------------------------------------------- accel

int rtk_accel_build(const rtk_scene *scene, const rtk_accel_params *params, rtk_accel **out) {
    if (!scene || !out) return fail(RTK_ERR_INVALID, "null scene or out");
    *out = nullptr;
    try {
        rtk_accel *a = new rtk_accel();
        a->scene = *scene;
        if (params) a->params = *params;
        else { a->params.max_depth = 8; a->params.max_leaf_size = 64; a->params.eps = 1e-6f; a->params.normalize_hit_normal = 1; a->params.device = -1; a->params.traversal = RTK_TRAVERSAL_REFERENCE; }
        if (a->params.traversal != RTK_TRAVERSAL_REFERENCE && a->params.traversal != RTK_TRAVERSAL_FAST) { delete a; return fail(RTK_ERR_INVALID, "unknown traversal"); }
        std::string err;
        const int rc = build_tree(a->scene, a->params.max_depth, a->params.max_leaf_size, a->tree, err);
        if (rc != RTK_OK) { delete a; return fail(rc, err); }
        if (a->tree.dev_nodes.size() > 0x7FFFFFFFull || a->tree.dev_tris.size() > 0x7FFFFFFFull) {
            delete a; return fail(RTK_ERR_INVALID, "tree too large for 32-bit node/triangle indices");
        }
        // eps: the reciprocal-estimate prefilter of tri_step and the bundle culling both assume that a determinant which
        // passes `eps <= |det|` is a normal float with a finite reciprocal
        if (!(a->params.eps >= 1.17549435e-38f && a->params.eps < 1.0f)) {
            delete a; return fail(RTK_ERR_INVALID, "eps must be in [FLT_MIN, 1)");
        }
        for (const DevMaterial &m : a->scene.materials) if (m.kind == RTK_MAT_REFRACTIVE) a->has_refractive = true;
        a->knobs = rtk_knobs::from_env();
        {
            double refs = 0.0, sq = 0.0;                        // size of the leaf a random triangle reference lives in
            for (const DevNode &l : a->tree.dev_leaves) { refs += double(l.b); sq += double(l.b) * double(l.b); }
            a->stream_slices_auto = (refs > 0.0 && sq / refs < 150.0) ? 1 : 4;
        }
        a->fast_traversal = a->params.traversal == RTK_TRAVERSAL_FAST || a->knobs.traversal_fast;
        if (a->fast_traversal) build_fast_leaf_orders(a->tree);
        a->coords_small = true;
        for (const DevTri &t : a->tree.dev_tris)
            for (int k = 0; k < 3; ++k)
                if (!(std::fabs(t.v0[k]) <= dev::kBundleLimit && std::fabs(t.e1[k]) <= dev::kBundleLimit && std::fabs(t.e2[k]) <= dev::kBundleLimit))
                    a->coords_small = false;
        *out = a;
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_accel_tree_info(const rtk_accel *a, rtk_tree_info *info) {
    if (!a || !info) return fail(RTK_ERR_INVALID, "null accel or info");
    std::memset(info, 0, sizeof(*info));
    info->n_nodes = int32_t(a->tree.nodes.size());
    for (const HostNode &n : a->tree.nodes) {
        if (n.leaf_start >= 0) { info->n_leaves += 1; if (n.leaf_count > info->max_leaf_refs) info->max_leaf_refs = n.leaf_count; }
        else info->n_inner += 1;
    }
    info->n_leaf_refs = int32_t(a->tree.leaf_refs.size());
    info->n_triangles = int32_t(a->tree.triangles.size());
    info->tree_depth = a->tree.depth;
    return RTK_OK;
}

int rtk_accel_tree_dump(const rtk_accel *a, float *nodes_box, int32_t *nodes_link, int32_t *leaf_refs) {
    if (!a) return fail(RTK_ERR_INVALID, "null accel");
    for (size_t i = 0; i < a->tree.nodes.size(); ++i) {
        const HostNode &n = a->tree.nodes[i];
        if (nodes_box) {
            float *b = nodes_box + i * 6;
            b[0] = n.box.mn.x; b[1] = n.box.mn.y; b[2] = n.box.mn.z; b[3] = n.box.mx.x; b[4] = n.box.mx.y; b[5] = n.box.mx.z;
        }
        if (nodes_link) {
            int32_t *l = nodes_link + i * 4;
            l[0] = n.child0; l[1] = n.child1; l[2] = n.leaf_start; l[3] = n.leaf_count;
        }
    }
    if (leaf_refs && !a->tree.leaf_refs.empty())
        std::memcpy(leaf_refs, a->tree.leaf_refs.data(), a->tree.leaf_refs.size() * sizeof(int32_t));
    return RTK_OK;
}

void rtk_accel_destroy(rtk_accel *a) {
    if (!a) return;
    if (a->on_device) {
        (void)hipSetDevice(a->device);
        (void)hipFree(a->d_nodes); (void)hipFree(a->d_leaves); (void)hipFree(a->d_tris); (void)hipFree(a->d_tri_ids); (void)hipFree(a->d_shade);
        (void)hipFree(a->d_materials); (void)hipFree(a->d_lights); (void)hipFree(a->d_counters);
        (void)hipFree(a->d_textures); (void)hipFree(a->d_tri_uv); (void)hipFree(a->d_tex_pixels); (void)hipFree(a->d_leaves_fast);
        (void)hipFree(a->d_occl_nodes); (void)hipFree(a->d_occl_leaves); (void)hipFree(a->d_occl_tris); (void)hipFree(a->d_occl_ids);
        free_stream_ws(a);
        // (the lane and side streams are the process's: LaneStreams)
        for (auto &e : a->lane_done) if (e) (void)hipEventDestroy(e);
        if (a->lane_fork) (void)hipEventDestroy(a->lane_fork);
        for (auto &sd : a->lane_side)
            for (int par = 0; par < 2; ++par) {
                if (sd.ready[par]) (void)hipEventDestroy(sd.ready[par]);
                if (sd.done[par]) (void)hipEventDestroy(sd.done[par]);
            }
        (void)hipFree(a->tp_prim); (void)hipFree(a->tp_bins); (void)hipFree(a->tp_bin_list); (void)hipFree(a->tp_order);
        (void)hipFree(a->rp_bounds); (void)hipFree(a->rp_keys); (void)hipFree(a->rp_idx); (void)hipFree(a->rp_temp);
        (void)hipFree(a->fb_cost); (void)hipFree(a->fb_order); (void)hipFree(a->fb_bins);
        for (auto &e : a->trial_ev) if (e) (void)hipEventDestroy(e);
        if (a->rp_done) (void)hipEventDestroy(a->rp_done);
        if (a->rp_probe_ev) (void)hipEventDestroy(a->rp_probe_ev);
        if (a->rp_host) (void)hipHostFree(a->rp_host);
        if (a->fb_nwgs_ev) (void)hipEventDestroy(a->fb_nwgs_ev);
        if (a->fb_nwgs_host) (void)hipHostFree(a->fb_nwgs_host);
    }
    delete a;
}

// ---------------------------------------------------------------- batched intersect

// workspace of the ray repacking: keys and indices (double-buffered for the sort), rocPRIM's temporary storage; grows, never shrinks
static int ensure_repack_ws(rtk_accel *a, size_t n) {
    if (!a->rp_bounds) RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->rp_bounds), kRepackBoundsAlloc * sizeof(uint32_t)));
    if (a->rp_cap >= n) return RTK_OK;
    (void)hipFree(a->rp_keys); (void)hipFree(a->rp_idx); (void)hipFree(a->rp_temp);
    a->rp_keys = a->rp_idx = nullptr; a->rp_temp = nullptr; a->rp_cap = 0;
    size_t tb = 0;
    RTK_HIP(repack_temp_bytes(n, &tb));
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->rp_keys), 2 * n * sizeof(uint32_t)));
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->rp_idx), 2 * n * sizeof(uint32_t)));
    RTK_HIP(hipMalloc(&a->rp_temp, tb > 0 ? tb : 16));
    a->rp_temp_bytes = tb; a->rp_cap = n;
    return RTK_OK;
}

static int intersect_device_impl(rtk_accel *a, const rtk_ray *d_rays, size_t n, int cull, int mode, rtk_hit *d_out,
                                 hipStream_t s, bool stats) {
    if (!valid_batch_mode(mode)) return fail(RTK_ERR_INVALID, "unknown trace_mode");
    if (n > (size_t(1) << 38)) return fail(RTK_ERR_INVALID, "too many rays for one launch");
    if (n > 0 && (!d_rays || !d_out)) return fail(RTK_ERR_INVALID, "null ray or hit buffer");
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) return fail(RTK_ERR_INVALID, "hit buffer must be 16-byte aligned");
    dev::IntersectArgs A;
    A.tree = tree_view(a);
    A.rays = d_rays; A.out = d_out; A.n = n; A.cull = cull ? 1 : 0; A.counters = a->d_counters; A.perm = nullptr; A.raster_w = 0u; A.verdict = nullptr;
    A.tree.scalar_surv = a->knobs.batch_scalar_surv ? 1 : 0;
    // Ray repacking (repack.hip).  Large batches are probed first (every 16th wave; one stream synchronisation): waves that are
    // coherent as they come are walked wave-cooperatively; a batch in no useful order is sorted by origin / direction cell and
    // then walked wave-cooperatively when the sort makes tight waves (three varying dimensions: 10 bits each), with the per-lane
    // fallback when it cannot (six: 5 bits each).  4 M rays on scene5: shuffled camera rays 4.5 -> 0.47 ms, uniform secondary
    // rays 11.3 -> 4.1 ms, camera rays in pixel order 0.8 (RTK_TRACE_AUTO before) -> 0.3 ms.
    const bool big = n >= (size_t(1) << 18) && n < (size_t(1) << 32);
    const bool forced = mode == RTK_TRACE_REPACK;
    if (forced) mode = RTK_TRACE_AUTO;
    const bool sortable = !stats && n >= 2 && n < (size_t(1) << 32);
    const bool probe = !stats && !forced && mode == RTK_TRACE_AUTO && a->knobs.repack && big;
    if ((forced && sortable) || probe) {
        int rc = ensure_repack_ws(a, n);
        if (rc != RTK_OK) return rc;
        // The workspace (bounds, keys, permutation) belongs to the accel: a batch on another stream may still be walking the
        // permutation of the previous call.  Its k_intersect recorded rp_done; this stream waits for it before it rewrites anything.
        if (a->rp_done == nullptr) RTK_HIP(hipEventCreateWithFlags(&a->rp_done, hipEventDisableTiming));
        if (a->rp_in_use) RTK_HIP(hipStreamWaitEvent(s, a->rp_done, 0));
        hipError_t eb = hipSuccess;
        bool sort = forced;
        unsigned sort_from_bit = 0u;
        bool keys_made = false;
        int sorted_mode = RTK_TRACE_AUTO;                                    // any order: wave-cooperative with the per-lane fallback
        if (probe) {
            // AUTO: the probe's verdict is needed on the host (one stream synchronisation; RTK_TRACE_REPACK and RTK_REPACK=0 never block)
            // The verdict is made on the device (k_raster_probe) and needed on the host; while it travels, the launch a coherent
            // batch needs is already under way -- it reads the same verdict and does nothing if the batch is to be sorted.
            const uint32_t probe_stride = uint32_t(((n + 63) / 64 + 4095) / 4096 > 16 ? ((n + 63) / 64 + 4095) / 4096 : 16);   // ~4,096 waves looked at
            unsigned fold = 0;
            eb = launch_ray_bounds(d_rays, n, a->rp_bounds, probe_stride, true, s, &fold);
            if (eb == hipSuccess) eb = launch_raster_probe(d_rays, n, a->rp_bounds, a->knobs.raster_tiles, s, fold);
            if (eb != hipSuccess) return hip_fail(eb, "launch k_ray_bounds (probe)");
            if (!a->rp_host) {
                RTK_HIP(hipHostMalloc(reinterpret_cast<void **>(&a->rp_host), kRepackBoundsWords * sizeof(uint32_t), hipHostMallocDefault));
                RTK_HIP(hipEventCreateWithFlags(&a->rp_probe_ev, hipEventDisableTiming));
            }
            uint32_t *h = a->rp_host;
            RTK_HIP(hipMemcpyAsync(h, a->rp_bounds, kRepackBoundsWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            RTK_HIP(hipEventRecord(a->rp_probe_ev, s));
            {
                dev::IntersectArgs Spec = A;
                Spec.verdict = a->rp_bounds;
                const hipError_t es = launch_intersect(Spec, RTK_TRACE_WAVE, false, s);
                if (es != hipSuccess) return hip_fail(es, "launch k_intersect");
                a->rp_in_use = true;
            }
            // ... and so are the keys a batch to be sorted needs (k_ray_keys returns at once if it is not): the verdict's trip to the
            // host and the launches that follow it no longer leave the stream idle
            if (!a->knobs.repack_full_bounds) {
                eb = launch_ray_keys(d_rays, n, a->rp_bounds, a->rp_keys, a->rp_idx, s, a->knobs.repack_dirs3, true);
                if (eb != hipSuccess) return hip_fail(eb, "launch k_ray_keys");
                keys_made = true;
            }
            RTK_HIP(hipEventRecord(a->rp_done, s));                         // (both read the workspace's verdict words)
            RTK_HIP(hipEventSynchronize(a->rp_probe_ev));                   // the verdict, not the trace
            sort = h[16] != 0u;
            if (!sort) return RTK_OK;                                        // coherent as it comes: that launch was the batch
            if (h[17] <= 3u) {
                sorted_mode = RTK_TRACE_WAVE;                                // ten or fifteen bits per dimension: the sort makes tight waves
                // ... also when the lowest ones stay unsorted: one or two radix passes less (two dimensions: 8 of the 15 bits each)
                sort_from_bit = h[17] <= 2u ? uint32_t(a->knobs.repack_skip_bits2) : uint32_t(a->knobs.repack_skip_bits);
            }
        }
        if (sort) {
            // The cells of the sort keys lie in the bounds of a SAMPLE of the batch (the probe's, where there was one; ~4,096 waves
            // otherwise): a ray outside them lands in a border cell -- an order a little worse for it, never another result -- and a
            // pass over all rays (0.13 ms of a 2^24-ray batch's 1.8) is saved.
            if (!probe) {
                const size_t waves = (n + 63) / 64;
                eb = launch_ray_bounds(d_rays, n, a->rp_bounds, a->knobs.repack_full_bounds ? 1u : uint32_t((waves + 4095) / 4096), false, s);
            } else if (a->knobs.repack_full_bounds) eb = launch_ray_bounds(d_rays, n, a->rp_bounds, 1u, false, s);
            if (eb == hipSuccess && !keys_made) eb = launch_ray_keys(d_rays, n, a->rp_bounds, a->rp_keys, a->rp_idx, s, a->knobs.repack_dirs3, false);
            if (eb == hipSuccess) eb = launch_key_sort(n, a->rp_keys, a->rp_idx, a->rp_temp, a->rp_temp_bytes, s, sort_from_bit);
            if (eb != hipSuccess) return hip_fail(eb, "ray repacking");
            A.perm = a->rp_idx + n;
            mode = a->knobs.repack_trace >= 0 ? a->knobs.repack_trace : sorted_mode;
        }
    }
    const hipError_t e = launch_intersect(A, mode, stats, s);
    if (e != hipSuccess) return hip_fail(e, "launch k_intersect");
    if (A.perm != nullptr) {
        RTK_HIP(hipEventRecord(a->rp_done, s));
        a->rp_in_use = true;
    }
    return RTK_OK;
}

int rtk_accel_intersect_device(rtk_accel *a, const rtk_ray *d_rays, size_t n, int cull, int mode, rtk_hit *d_out, void *stream) {
    if (!a) return fail(RTK_ERR_INVALID, "null accel");
    std::lock_guard<std::mutex> lock(a->mu);
    const int rc = ensure_device(a);
    if (rc != RTK_OK) return rc;
    return intersect_device_impl(a, d_rays, n, cull, mode, d_out, static_cast<hipStream_t>(stream), false);
}

int rtk_accel_intersect_stats(rtk_accel *a, const rtk_ray *d_rays, size_t n, int cull, int mode, rtk_hit *d_out,
                              rtk_counters *counters) {
    if (!a || !counters) return fail(RTK_ERR_INVALID, "null accel or counters");
    std::lock_guard<std::mutex> lock(a->mu);
    int rc = ensure_device(a);
    if (rc != RTK_OK) return rc;
    RTK_HIP(hipMemsetAsync(a->d_counters, 0, kCounterWords * sizeof(unsigned long long), nullptr));
    rc = intersect_device_impl(a, d_rays, n, cull, mode, d_out, nullptr, true);
    if (rc != RTK_OK) return rc;
    unsigned long long h[8];
    RTK_HIP(hipMemcpy(h, a->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    counters->rays = h[0]; counters->primary = 0; counters->hits = h[2]; counters->nodes = h[3]; counters->boxpass = h[4];
    counters->leaves = h[5]; counters->tris = h[6]; counters->packets16 = h[7];
    return RTK_OK;
}

int rtk_accel_intersect(rtk_accel *a, const rtk_ray *rays, size_t n, int cull, int mode, rtk_hit *out) {
    if (!a) return fail(RTK_ERR_INVALID, "null accel");
    if (n > 0 && (!rays || !out)) return fail(RTK_ERR_INVALID, "null ray or hit buffer");
    std::lock_guard<std::mutex> lock(a->mu);
    int rc = ensure_device(a);
    if (rc != RTK_OK) return rc;
    if (n == 0) return RTK_OK;
    rtk_ray *d_rays = nullptr;
    rtk_hit *d_out = nullptr;
    RTK_HIP(hipMalloc(reinterpret_cast<void **>(&d_rays), n * sizeof(rtk_ray)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), n * sizeof(rtk_hit));
    if (e != hipSuccess) { (void)hipFree(d_rays); return hip_fail(e, "hipMalloc hits"); }
    e = hipMemcpy(d_rays, rays, n * sizeof(rtk_ray), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = intersect_device_impl(a, d_rays, n, cull, mode, d_out, nullptr, false);
        if (rc == RTK_OK) e = hipMemcpy(out, d_out, n * sizeof(rtk_hit), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_rays); (void)hipFree(d_out);
    if (rc != RTK_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "intersect copy");
    return RTK_OK;
}

// ---------------------------------------------------------------- frame

int rtk_render_output_floats(const rtk_accel *a, const rtk_render_params *p, size_t *n_floats) {
    if (!n_floats) return fail(RTK_ERR_INVALID, "null n_floats");
    FrameGeom g;
    const int rc = frame_geom(a, p, g);
    if (rc != RTK_OK) return rc;
    *n_floats = (g.world > 1) ? size_t(g.buckets_per_rank) * g.bucket * g.bucket * 3 : size_t(g.width) * g.height * 3;
    return RTK_OK;
}

static int render_device_impl(rtk_accel *a, const rtk_render_params *p, float *d_out, hipStream_t s) {
    FrameGeom g;
    int rc = frame_geom(a, p, g);
    if (rc != RTK_OK) return rc;
    if (!d_out) return fail(RTK_ERR_INVALID, "null output buffer");
    dev::RenderArgs A;
    std::memset(&A, 0, sizeof(A));
    A.tree = tree_view(a);
    A.materials = a->d_materials; A.lights = a->d_lights;
    A.textures = a->d_textures; A.tri_uv = a->d_tri_uv; A.tex_pixels = a->d_tex_pixels;
    A.n_lights = int(a->scene.lights.size());
    A.has_refractive = a->has_refractive ? 1 : 0;
    std::memcpy(A.cam_pos, a->scene.cam_pos, sizeof(A.cam_pos));
    std::memcpy(A.cam_mat, a->scene.cam_mat, sizeof(A.cam_mat));
    std::memcpy(A.background, a->scene.background, sizeof(A.background));
    A.width = g.width; A.height = g.height;
    A.aspect = static_cast<float>(g.width) / static_cast<float>(g.height);                    // render.hpp:26
    // render.hpp:55-57: `const F fov_radians = degrees_to_radians(fov_degrees)` is evaluated in double (fov_degrees is a
    // double constant, utils/convert.hpp:4-6) and ROUNDED TO FLOAT by the declaration; `std::tan(fov_radians / F(2))` is
    // then the float overload (tanf), and `screen_x *=` a float multiply (common.hip.hpp camera_ray).
    const float fov_radians = static_cast<float>(p->fov_degrees * (3.14159265358979323846 / 180.0));
    A.tan_half_fov = std::tan(fov_radians / 2.0f);
    A.spp = p->spp; A.max_depth = p->max_ray_depth; A.diffuse_rays = p->diffuse_rays; A.seed = p->seed;
    A.width_f = static_cast<float>(g.width); A.height_f = static_cast<float>(g.height);
    A.spp_f = static_cast<float>(p->spp); A.gi_div_f = static_cast<float>(p->diffuse_rays + 1);
    A.sample_begin = g.sample_begin; A.sample_end = g.sample_end;
    A.shadow_bias = p->shadow_bias; A.reflection_bias = p->reflection_bias; A.refraction_bias = p->refraction_bias;
    A.bucket = g.bucket; A.tiles_x = g.tiles_x; A.tiles_y = g.tiles_y; A.n_buckets = g.n_buckets;
    A.blocks_per_bucket_side = g.blocks_side; A.buckets_per_rank = g.buckets_per_rank;
    A.rank = g.rank; A.world = g.world; A.compact = g.world > 1 ? 1 : 0; A.skew_q = g.skew_q;
    A.out = d_out; A.counters = a->d_counters;
    A.slice_min_tris = a->knobs.slice_min_tris;
    const bool forks = a->has_refractive || p->diffuse_rays > 0;
    // the megakernel comes in two builds: the lean one (diffuse / reflective / constant materials only) and the general one
    // (template FORKS: + refraction, diffuse GI, textures), so that the lean one does not carry the general one's registers
    const bool general = forks || !a->scene.textures.empty();
    // Occlusion queries (is_occluded) may stop at the first hit nearer than the light when no material is transmissive: the
    // frame is bit-identical (trace.hip.hpp, `exit_t`), only the per-ray work counters shrink.  collect_stats == 1 counts the
    // reference's work (every ray traced to the end), collect_stats == 2 the work of the production path.
    A.shadow_exit = (a->knobs.shadow_exit && !a->has_refractive && p->collect_stats != 1) ? 1 : 0;
    A.occl_on = (a->occl_on && p->collect_stats == 0) ? 1 : 0;
    A.occl = A.tree;
    if (A.occl_on) {
        A.occl.nodes = a->d_occl_nodes; A.occl.leaves = a->d_occl_leaves; A.occl.leaves_fast = nullptr; A.occl.n_leaves = a->occl_n_leaves;
        A.occl.tris = a->d_occl_tris; A.occl.tri_ids = a->d_occl_ids;
    }
    RTK_HIP(hipMemsetAsync(a->d_counters, 0, (kCounterWords + 4) * sizeof(unsigned long long), s));
    if (g.world > 1 && g.sample_begin == 0) {
        // buckets past the end of the frame (padding so that every rank has equal length) stay zero
        size_t nf = size_t(g.buckets_per_rank) * g.bucket * g.bucket * 3;
        RTK_HIP(hipMemsetAsync(d_out, 0, nf * sizeof(float), s));
    }
    // fork-free scenes (no refraction, no GI) can be rendered by the streaming pipeline (stream.hip)
    // RTK_TRACE_AUTO for frames: scenes whose ray trees fork (refraction, diffuse GI) go through the streaming pipeline
    // (all rays of a depth level in parallel); fork-free scenes through the GROUP4 megakernel (fewer launches)
    // Which of the two wins on a forking scene depends on how much of the frame forks (refractive dragon: pipeline 3x;
    // a small glass object: megakernel 2x), so RTK_TRACE_AUTO times both on the first frames of a shape and keeps the
    // faster: frame 1 pipeline, frames 2-3 megakernel (the second one with its cost-feedback order), then the verdict
    // as soon as the events have completed (hipEventQuery, never a host wait).  Both engines produce the same frame.
    bool stream = p->trace_mode == RTK_TRACE_STREAM || (p->trace_mode == RTK_TRACE_AUTO && forks);
    hipEvent_t trial_start = nullptr, trial_end = nullptr;
    if (p->trace_mode == RTK_TRACE_AUTO && forks && !p->collect_stats) {
        const bool trials = a->knobs.auto_trials;
        const uint64_t tsig[3] = {(uint64_t(uint32_t(g.width)) << 32) | uint32_t(g.height), (uint64_t(uint32_t(g.rank)) << 32) | uint32_t(g.world),
                                  (uint64_t(uint32_t(p->spp)) << 32) | (uint64_t(uint32_t(p->max_ray_depth)) << 16) | uint32_t(p->diffuse_rays)};
        if (std::memcmp(tsig, a->trial_sig, sizeof(tsig)) != 0) { std::memcpy(a->trial_sig, tsig, sizeof(tsig)); a->trial_state = 0; }
        if (trials) {
            if (!a->trial_ev[0]) for (auto &e : a->trial_ev) RTK_HIP(hipEventCreate(&e));
            if (a->trial_state == 3) {                                      // both timed: is the verdict in?
                float t_stream = 0.f, t_mega = 0.f;
                if (hipEventQuery(a->trial_ev[1]) == hipSuccess && hipEventQuery(a->trial_ev[3]) == hipSuccess &&
                    hipEventElapsedTime(&t_stream, a->trial_ev[0], a->trial_ev[1]) == hipSuccess &&
                    hipEventElapsedTime(&t_mega, a->trial_ev[2], a->trial_ev[3]) == hipSuccess)
                    a->trial_state = t_mega < t_stream ? 5 : 4;
                else (void)hipGetLastError();                               // not ready yet: clear the sticky "not ready"
            }
            switch (a->trial_state) {
                case 0: stream = true; trial_start = a->trial_ev[0]; trial_end = a->trial_ev[1]; a->trial_state = 1; break;
                case 1: stream = false; a->trial_state = 2; break;          // first megakernel frame: records the block costs
                case 2: stream = false; trial_start = a->trial_ev[2]; trial_end = a->trial_ev[3]; a->trial_state = 3; break;
                case 5: stream = false; break;
                default: stream = true; break;                              // 3 (waiting for the events), 4 (pipeline won)
            }
        }
    }
    // (the streaming pipeline's trial is started behind its workspace allocation, below: tens of GB of hipMalloc in front of the
    // first frame once made the megakernel "win" config 5's shape at 1.2 s a frame against 0.1)
    if (trial_start && !stream) RTK_HIP(hipEventRecord(trial_start, s));
    const bool twopass = p->trace_mode == RTK_TRACE_TWOPASS;
    if (twopass && p->spp != 1) return fail(RTK_ERR_UNSUPPORTED, "RTK_TRACE_TWOPASS needs spp == 1");
    if (twopass) {
        const size_t out_pixels = (g.world > 1) ? size_t(g.buckets_per_rank) * g.bucket * g.bucket : size_t(g.width) * g.height;
        const size_t tiles = size_t(g.buckets_per_rank) * g.blocks_side * g.blocks_side;
        rc = ensure_twopass_ws(a, out_pixels, tiles);
        if (rc != RTK_OK) return rc;
        A.prim = a->tp_prim; A.bin_count = a->tp_bins; A.n_listed = a->tp_bins + kCostBins; A.bin_list = a->tp_bin_list;
        A.tile_order = a->tp_order; A.tile_cap = uint32_t(a->tp_tiles);
        const hipError_t et = launch_twopass(A, p->collect_stats != 0, general, s);
        if (et != hipSuccess) return hip_fail(et, "launch two-pass frame");
    } else if (stream) {
        const size_t out_pixels = (g.world > 1) ? size_t(g.buckets_per_rank) * g.bucket * g.bucket : size_t(g.width) * g.height;
        const size_t n_root = size_t(g.buckets_per_rank) * g.blocks_side * g.blocks_side * 64;
        // Ray-tree nodes per sample: the camera rays plus room for the secondary rays.  Refractive scenes fork (two
        // children per interface), so they get more head room; an overflow is caught on the device and the frame
        // redone by the megakernel.
        const size_t factor = a->knobs.stream_node_factor > 0 ? size_t(a->knobs.stream_node_factor) : (forks ? 8 : 3);
        const int n_pass = g.sample_end - g.sample_begin;
        // Samples per launch (stream.hpp "batch") and batches in flight.  A batch's queues cost ~120 B per ray-tree node: the
        // batch is as large as the pass, the knob and the memory budget allow (288 GB of HBM is what this design spends).
        const size_t nodes_per_sample = n_root * factor + 4096;
        const size_t bytes_per_node = sizeof(dev::RayRec) + sizeof(dev::NodeRes) + sizeof(uint32_t) +
                                      (sizeof(dev::HitRec) + sizeof(uint32_t) + sizeof(float2) * (a->scene.lights.empty() ? 1 : a->scene.lights.size())) / 2 + 1;
        // Measured (gpurun_out/r03d-f, hw15/scene2 and hw11/scene8 at the BASELINE sizes): four batches in flight, each a
        // quarter of the pass, beat both more, smaller launches and fewer, larger ones (1920x1920, 16 samples, no helpers:
        // 53.1 ms one sample per launch, 47.6 four, 61.6 eight in two lanes; 960x960, 8 samples: 14.4 -> 9.5 ms).
        const int even = (n_pass + a->knobs.stream_lanes - 1) / a->knobs.stream_lanes;
        const int want = a->knobs.stream_batch > 0 ? a->knobs.stream_batch : even;
        int batch = n_pass < want ? n_pass : want;
        const size_t budget = size_t(a->knobs.stream_mem_gb) << 30;
        while (batch > 1 && (nodes_per_sample * size_t(batch) > 0xF0000000ull || nodes_per_sample * size_t(batch) * bytes_per_node > budget)) batch -= 1;
        const int n_launch = (n_pass + batch - 1) / batch;
        int lanes = n_launch < a->knobs.stream_lanes ? n_launch : a->knobs.stream_lanes;     // batches in flight at once (stream.hpp)
        while (lanes > 1 && nodes_per_sample * size_t(batch) * bytes_per_node * size_t(lanes) > budget) lanes -= 1;
        const size_t ws_nodes_before = a->ws_nodes;
        const int ws_lanes_before = a->ws_lanes;
        rc = ensure_stream_ws(a, out_pixels, nodes_per_sample * size_t(batch), a->scene.lights.size(), p->spp > 1, lanes);
        if (rc != RTK_OK) return rc;
        if (trial_start && (a->ws_nodes != ws_nodes_before || a->ws_lanes != ws_lanes_before)) {
            // fresh queues: their first use is not what a frame costs -- time the pipeline on the next frame instead
            trial_start = trial_end = nullptr;
            a->trial_state = 0;
        }
        if (trial_start) RTK_HIP(hipEventRecord(trial_start, s));
        dev::StreamArgs S;
        S.r = A; S.r.tree.scalar_surv = a->knobs.stream_scalar_surv ? 1 : 0; S.ws = a->ws;
        S.key_dirs = p->diffuse_rays > 0 ? 1u : 0u; S.level = 0; S.sample = 0; S.n_batch = 1; S.n_root = uint32_t(n_root); S.n_level0 = uint32_t(n_root); S.auto_min_lanes = a->knobs.auto_min_lanes;
        S.n_lanes = uint32_t(lanes);
        for (int j = 0; j < dev::kStreamLanes; ++j) S.lane_overflow[j] = a->ws_lane[j < lanes ? j : 0].ctrl + dev::kCtrlOverflow;
        // measured on MI355X: the workgroup-cooperative wave walk beats the per-lane walk at every depth, even for the
        // incoherent rays behind refractive surfaces, so no level switches strategy by default
        const int deep_level = a->knobs.stream_deep_level, deep_mode = a->knobs.stream_deep_mode;
        // fork-free trees stay coherent; sorting them would only add launches
        const int sort_from = a->knobs.stream_sort_from >= 0 ? a->knobs.stream_sort_from : (forks ? 1 : 99);
        {
            const DevNode &root = a->tree.dev_nodes[0];
            for (int k = 0; k < 3; ++k) {
                const float ext = root.hi[k] - root.lo[k];
                S.grid_lo[k] = root.lo[k];
                S.grid_scale[k] = (ext > 0.f && ext < 3.0e38f) ? 16.0f / ext : 0.f;
            }
        }
        S.nodes_sorted = S.hits_sorted = S.bin_children = S.bin_hits = 0;
        for (int j = 0; j < lanes; ++j) RTK_HIP(hipMemsetAsync(a->ws_lane[j].ctrl, 0, dev::kCtrlWords * sizeof(uint32_t), s));
        // fork: lane 0 is the caller's stream, the other lanes wait for everything enqueued on it so far
        if (lanes > 1) {
            RTK_HIP(hipEventRecord(a->lane_fork, s));
            for (int j = 1; j < lanes; ++j) RTK_HIP(hipStreamWaitEvent(a->lane_stream[j], a->lane_fork, 0));
        }
        for (int i = 0; i < n_launch; ++i) {
            const int j = i % lanes;
            S.sample = g.sample_begin + i * batch;
            S.n_batch = uint32_t(g.sample_end - S.sample < batch ? g.sample_end - S.sample : batch);
            S.n_level0 = uint32_t(n_root) * S.n_batch;
            S.ws = a->ws_lane[j];
            const hipStream_t ls = j == 0 ? s : a->lane_stream[j];
            const hipEvent_t wait = (lanes > 1 && i > 0) ? a->lane_done[(i - 1) % lanes] : nullptr;
            const hipEvent_t done = lanes > 1 ? a->lane_done[j] : nullptr;
            const hipError_t es = launch_stream_sample(S, p->collect_stats != 0, deep_level, deep_mode, sort_from, ls, wait, done,
                                                       // (side streams help while few rays are in flight: spp 1 16.7 -> 9.1 ms on config 3;
                                                       // with four lanes the GPU is full already and they cost 20 %)
                                                       (a->knobs.stream_side && lanes <= 2 && batch * lanes <= a->knobs.stream_side_below) ? &a->lane_side[j] : nullptr,
                                                       a->knobs.stream_slices > 0 ? a->knobs.stream_slices : a->stream_slices_auto);
            if (es != hipSuccess) return hip_fail(es, "launch streaming pipeline");
        }
        // join: the caller's stream continues behind the last sample of every lane
        for (int j = 1; j < lanes; ++j) RTK_HIP(hipStreamWaitEvent(s, a->lane_done[j], 0));
        S.ws = a->ws;
        // safety net: if any queue overflowed, the megakernel renders the frame again (a no-op otherwise)
        hipError_t ef = launch_stream_overflow_reset(S, s);
        if (ef != hipSuccess) return hip_fail(ef, "launch overflow reset");
        dev::RenderArgs F = A;
        F.only_if = a->ws.ctrl + dev::kCtrlOverflow;
        ef = launch_render(F, RTK_TRACE_GROUP4, p->collect_stats != 0, general, s);
        if (ef != hipSuccess) return hip_fail(ef, "launch fallback k_render");
        if (a->knobs.stream_debug) {
            uint32_t h[dev::kCtrlWords];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, a->ws.ctrl, sizeof(h), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[rtk stream] node_cap %u hit_cap %u overflow %u; nodes per level:", a->ws.node_cap, a->ws.hit_cap, h[dev::kCtrlOverflow]);
            for (int l = 0; l <= p->max_ray_depth + 1; ++l) std::fprintf(stderr, " %u", l == 0 ? unsigned(n_root) : h[dev::kCtrlNodeCount + l]);
            std::fprintf(stderr, "; hits:");
            for (int l = 0; l <= p->max_ray_depth; ++l) std::fprintf(stderr, " %u", h[dev::kCtrlHitCount + l]);
            std::fprintf(stderr, "\n");
        }
    } else {
        // Cost feedback: the frame time is set by the few pixel blocks whose rays graze the mesh (hundreds of microseconds
        // each, against ~3 for a background block).  Started late they are the tail of the frame, so every block reports
        // its cycle count and the next frame of the same shape starts them most-expensive-first.  Only the launch order
        // changes: every block is rendered in full, every frame.  RTK_COST_FEEDBACK=0 turns it off.
        const bool feedback = a->knobs.cost_feedback;
        const size_t units = size_t(g.buckets_per_rank) * g.blocks_side * g.blocks_side;
        A.n_units = uint32_t(units);
        // RTK_TRACE_AUTO for frames: workgroup-cooperative leaves.  Four waves per pixel block when there are enough blocks
        // to fill the chip several times over (the frame is then bound by how many blocks run side by side); eight when
        // there are few (a rank of a sharded frame, a small image: the frame is then as long as its most expensive block,
        // and eight waves get through its big leaves faster).  Measured on config 2 (tools/rank_times.py): 32,400 blocks
        // 0.44 ms (GROUP4) vs 0.87 (GROUP8); 4,050 blocks (one rank of eight) 0.43 vs 0.32.
        const size_t group8_below = a->knobs.group8_below;
        const int frame_mode = p->trace_mode != RTK_TRACE_AUTO ? p->trace_mode : (units < group8_below ? RTK_TRACE_GROUP8 : RTK_TRACE_GROUP4);
        if (feedback && units > 0 && units <= 0x7FFFFFFFull) {
            const uint64_t sig[4] = {(uint64_t(uint32_t(g.width)) << 32) | uint32_t(g.height),
                                     (uint64_t(uint32_t(g.rank)) << 32) | uint32_t(g.world),
                                     (uint64_t(uint32_t(p->spp)) << 32) | (uint64_t(uint32_t(p->max_ray_depth)) << 16) | uint32_t(p->diffuse_rays),
                                     (uint64_t(uint32_t(g.bucket)) << 32) | (uint64_t(uint32_t(g.sample_end - g.sample_begin) & 0xFFFFu) << 16) | uint32_t(p->trace_mode)};
            if (a->fb_units < units) {
                // (capacity, never shrunk; the first allocation also covers the scene's own frame size, so that a small frame
                // rendered first -- a warm-up -- does not leave three hipMallocs, ~0.1 ms, in front of the first full-size frame)
                size_t cap = units;
                {
                    const uint32_t bk = g.bucket, bs = g.blocks_side;
                    const uint64_t tx = (uint64_t(a->scene.width > 0 ? a->scene.width : 0) + bk - 1) / bk, ty = (uint64_t(a->scene.height > 0 ? a->scene.height : 0) + bk - 1) / bk;
                    const uint64_t native = tx * ty * bs * bs;
                    if (a->fb_units == 0 && native > cap && native <= (1ull << 24)) cap = size_t(native);
                }
                (void)hipFree(a->fb_cost); (void)hipFree(a->fb_order); (void)hipFree(a->fb_bins);
                a->fb_cost = a->fb_order = nullptr; a->fb_bins = nullptr; a->fb_units = 0; a->fb_valid = false; a->fb_order_valid = false;
                RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->fb_cost), cap * sizeof(uint32_t)));
                RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->fb_order), (2 * cap + 4 + 8) * sizeof(uint32_t)));   // order, header, workgroup list, prior's counters
                RTK_HIP(hipMalloc(reinterpret_cast<void **>(&a->fb_bins), cap));
                a->fb_units = cap;
            }
            const bool same_shape = a->fb_valid && std::memcmp(sig, a->fb_sig, sizeof(sig)) == 0;
            if (!same_shape) a->fb_order_valid = false;
            // The first frame of a shape has no costs to go by: a prior from the camera rays alone stands in for them
            // (k_block_prior: background blocks packed four to a workgroup, the others by what their centre ray looks at).
            // A one-shot render is exactly this frame (the reference CLI renders one, src/main.cpp:13-25).
            const bool prior = !same_shape && a->knobs.first_frame_prior && frame_mode == RTK_TRACE_GROUP4 && p->collect_stats == 0;
            if (prior) {
                const hipError_t ep = launch_block_prior(A, a->fb_bins, a->fb_order, a->fb_order + units + 4, a->fb_order + units,
                                                         reinterpret_cast<uint32_t *>(a->d_counters + kCounterWords), 4u, s);
                if (ep != hipSuccess) return hip_fail(ep, "launch k_block_prior");
                A.order_in = a->fb_order; A.order_hdr = a->fb_order + units; A.wg_list = a->fb_order + units + 4;
            }
            if (same_shape) {
                // The order is refreshed from the newest costs every few frames only: the sort is one small workgroup whose
                // ~28 us sit in front of the frame, and an order that is a few frames old is as good (costs move slowly).
                const unsigned every = a->knobs.resort_every;
                if (!a->fb_order_valid || a->fb_age >= every) {
                    // blocks that cost less than this many cycles (background, a handful of nodes) are packed four to a workgroup
                    const uint32_t light_cycles = a->knobs.light_cycles;
                    const bool group_mode = frame_mode == RTK_TRACE_GROUP4;                 // light packing: GROUP4 only
                    const hipError_t eo = launch_order_by_cost(a->fb_cost, a->fb_bins, a->fb_order, a->fb_order + units + 4, a->fb_order + units,
                                                               uint32_t(units), group_mode ? light_cycles >> 4 : 0u,
                                                               a->knobs.order_floor_cycles >> 4, 4u, s);
                    if (eo != hipSuccess) return hip_fail(eo, "launch k_order_by_cost");
                    a->fb_order_valid = true;
                    a->fb_age = 0;
                    // how many workgroups the list has: known on the host a frame or two later; until then the launch covers every block
                    if (!a->fb_nwgs_host) {
                        RTK_HIP(hipHostMalloc(reinterpret_cast<void **>(&a->fb_nwgs_host), sizeof(uint32_t), hipHostMallocDefault));
                        RTK_HIP(hipEventCreateWithFlags(&a->fb_nwgs_ev, hipEventDisableTiming));
                    }
                    a->fb_nwgs_known = false;
                    RTK_HIP(hipMemcpyAsync(a->fb_nwgs_host, a->fb_order + units, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    RTK_HIP(hipEventRecord(a->fb_nwgs_ev, s));
                    a->fb_nwgs_pending = true;
                }
                a->fb_age += 1;
                A.order_in = a->fb_order; A.order_hdr = a->fb_order + units; A.wg_list = a->fb_order + units + 4;
            }
            A.cost_out = a->fb_cost;
            std::memcpy(a->fb_sig, sig, sizeof(sig));
            a->fb_valid = true;
        }
        unsigned n_wgs = 0;
        if (A.wg_list != nullptr && A.order_in == a->fb_order && a->fb_order_valid) {
            if (a->fb_nwgs_pending && hipEventQuery(a->fb_nwgs_ev) == hipSuccess) { a->fb_nwgs_pending = false; a->fb_nwgs_known = true; }
            else if (a->fb_nwgs_pending) (void)hipGetLastError();            // not ready: clear the sticky status
            if (a->fb_nwgs_known && !a->fb_nwgs_pending) n_wgs = *a->fb_nwgs_host;
        }
        const hipError_t e = launch_render(A, frame_mode, p->collect_stats != 0, general, s, n_wgs);
        if (e != hipSuccess) return hip_fail(e, "launch k_render");
    }
    if (trial_end) RTK_HIP(hipEventRecord(trial_end, s));
    a->last_stream = s;
    a->last_stats = p->collect_stats != 0;
    // primary rays of this rank: pixels of its buckets x spp
    uint64_t pixels = 0;
    for (uint32_t j = 0; j < g.buckets_per_rank; ++j) {
        const uint32_t b = dev::rank_bucket(uint32_t(g.rank), j, uint32_t(g.world), g.skew_q);
        if (b >= g.n_buckets) continue;
        const uint32_t bx = (b % g.tiles_x) * g.bucket, by = (b / g.tiles_x) * g.bucket;
        const uint32_t w = (bx + g.bucket <= g.width) ? g.bucket : g.width - bx;
        const uint32_t h = (by + g.bucket <= g.height) ? g.bucket : g.height - by;
        pixels += uint64_t(w) * h;
    }
    a->last_primary = pixels * uint64_t(g.sample_end - g.sample_begin);
    return RTK_OK;
}

int rtk_render_frame_device(rtk_accel *a, const rtk_render_params *p, float *d_out, void *stream) {
    if (!a || !p) return fail(RTK_ERR_INVALID, "null accel or params");
    std::lock_guard<std::mutex> lock(a->mu);
    const int rc = ensure_device(a);
    if (rc != RTK_OK) return rc;
    return render_device_impl(a, p, d_out, static_cast<hipStream_t>(stream));
}

int rtk_render_last_counters(rtk_accel *a, rtk_counters *c) {
    if (!a || !c) return fail(RTK_ERR_INVALID, "null accel or counters");
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->on_device) return fail(RTK_ERR_INVALID, "no frame has been rendered on this accel");
    RTK_HIP(hipSetDevice(a->device));
    RTK_HIP(hipStreamSynchronize(a->last_stream));
    unsigned long long h[kCounterWords];
    RTK_HIP(hipMemcpy(h, a->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    std::memset(c, 0, sizeof(*c));
    for (int i = 0; i < kRayCounterShards; ++i) h[0] += h[8 + i];      // the frame kernel shards its ray counter
    c->rays = h[0]; c->primary = a->last_primary;
    if (a->last_stats) { c->hits = h[2]; c->nodes = h[3]; c->boxpass = h[4]; c->leaves = h[5]; c->tris = h[6]; c->packets16 = h[7]; }
    return RTK_OK;
}

int rtk_render_last_critical_path(rtk_accel *a, double *ms) {
    if (!a || !ms) return fail(RTK_ERR_INVALID, "null accel or ms");
    std::lock_guard<std::mutex> lock(a->mu);
    if (!a->on_device) return fail(RTK_ERR_INVALID, "no frame has been rendered on this accel");
    RTK_HIP(hipSetDevice(a->device));
    RTK_HIP(hipStreamSynchronize(a->last_stream));
    unsigned long long shard[kRayCounterShards], ticks = 0;
    RTK_HIP(hipMemcpy(shard, a->d_counters + kCriticalWord, sizeof(shard), hipMemcpyDeviceToHost));
    for (unsigned long long t : shard) ticks = t > ticks ? t : ticks;
    *ms = double(ticks) * 1.0e-5;                                        // s_memrealtime counts at 100 MHz; 0 = no block took 10 us
    return RTK_OK;
}

int rtk_render_frame(rtk_accel *a, const rtk_render_params *p, float *rgb, rtk_counters *counters) {
    if (!a || !p || !rgb) return fail(RTK_ERR_INVALID, "null accel, params or rgb");
    if (p->world_size > 1) return fail(RTK_ERR_INVALID, "rtk_render_frame renders whole frames; use rtk_render_frame_device for sharded output");
    size_t nf = 0;
    int rc = rtk_render_output_floats(a, p, &nf);
    if (rc != RTK_OK) return rc;
    {
        std::lock_guard<std::mutex> lock(a->mu);
        rc = ensure_device(a);
        if (rc != RTK_OK) return rc;
        float *d_out = nullptr;
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&d_out), nf * sizeof(float)));
        hipError_t e = hipSuccess;
        // a later pass of a progressive frame continues the running per-pixel sums the previous pass left in `rgb`
        if (p->sample_begin > 0) e = hipMemcpy(d_out, rgb, nf * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d_out); return hip_fail(e, "upload of the running sums"); }
        rc = render_device_impl(a, p, d_out, nullptr);
        if (rc == RTK_OK) e = hipMemcpy(rgb, d_out, nf * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(d_out);
        if (rc != RTK_OK) return rc;
        if (e != hipSuccess) return hip_fail(e, "frame copy");
    }
    if (counters) return rtk_render_last_counters(a, counters);
    return RTK_OK;
}

int rtk_tiles_assemble_device(const rtk_accel *a, const rtk_render_params *p, const float *d_gathered, float *d_rgb, void *stream) {
    FrameGeom g;
    const int rc = frame_geom(a, p, g);
    if (rc != RTK_OK) return rc;
    if (!d_gathered || !d_rgb) return fail(RTK_ERR_INVALID, "null buffer");
    dev::AssembleArgs A;
    A.gathered = d_gathered; A.rgb = d_rgb; A.width = g.width; A.height = g.height; A.bucket = g.bucket;
    A.tiles_x = g.tiles_x; A.world = uint32_t(g.world); A.buckets_per_rank = g.buckets_per_rank; A.skew_q = g.skew_q;
    const hipError_t e = launch_assemble(A, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch k_assemble");
    return RTK_OK;
}

// ---------------------------------------------------------------- camera rays

static int camera_args(rtk_accel *a, const rtk_render_params *p, int32_t sample, dev::RenderArgs &A) {
    FrameGeom g;
    const int rc = frame_geom(a, p, g);
    if (rc != RTK_OK) return rc;
    if (sample < 0 || sample >= p->spp) return fail(RTK_ERR_INVALID, "sample must be in [0, spp)");
    std::memset(&A, 0, sizeof(A));
    std::memcpy(A.cam_pos, a->scene.cam_pos, sizeof(A.cam_pos));
    std::memcpy(A.cam_mat, a->scene.cam_mat, sizeof(A.cam_mat));
    A.width = g.width; A.height = g.height;
    A.aspect = static_cast<float>(g.width) / static_cast<float>(g.height);
    A.tan_half_fov = std::tan(static_cast<float>(p->fov_degrees * (3.14159265358979323846 / 180.0)) / 2.0f);   // as render_device_impl
    A.spp = p->spp; A.seed = p->seed;
    A.width_f = static_cast<float>(g.width); A.height_f = static_cast<float>(g.height); A.spp_f = static_cast<float>(p->spp);
    return RTK_OK;
}

int rtk_camera_rays_device(rtk_accel *a, const rtk_render_params *p, int32_t sample, rtk_ray *d_rays, void *stream) {
    if (!a || !p || !d_rays) return fail(RTK_ERR_INVALID, "null accel, params or ray buffer");
    std::lock_guard<std::mutex> lock(a->mu);
    int rc = ensure_device(a);
    if (rc != RTK_OK) return rc;
    dev::RenderArgs A;
    if ((rc = camera_args(a, p, sample, A)) != RTK_OK) return rc;
    const hipError_t e = launch_camera_rays(A, sample, d_rays, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch k_camera_rays");
    return RTK_OK;
}

int rtk_camera_rays(rtk_accel *a, const rtk_render_params *p, int32_t sample, rtk_ray *rays) {
    if (!a || !p || !rays) return fail(RTK_ERR_INVALID, "null accel, params or ray buffer");
    FrameGeom g;
    int rc = frame_geom(a, p, g);
    if (rc != RTK_OK) return rc;
    const size_t n = size_t(g.width) * g.height;
    rtk_ray *d = nullptr;
    {
        std::lock_guard<std::mutex> lock(a->mu);
        if ((rc = ensure_device(a)) != RTK_OK) return rc;
        RTK_HIP(hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(rtk_ray)));
    }
    rc = rtk_camera_rays_device(a, p, sample, d, nullptr);
    hipError_t e = hipSuccess;
    if (rc == RTK_OK) e = hipMemcpy(rays, d, n * sizeof(rtk_ray), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (rc != RTK_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "camera ray copy");
    return RTK_OK;
}

// ---------------------------------------------------------------- image out

int rtk_format_ppm(const float *rgb, int32_t width, int32_t height, char *buf, size_t cap, size_t *n) {
    if (!rgb || !n || width <= 0 || height <= 0) return fail(RTK_ERR_INVALID, "bad image");
    try {
        const std::string s = format_ppm(rgb, width, height);
        *n = s.size();
        if (buf) std::memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_frame_to_rgb8_device(const float *d_rgb, size_t n, uint8_t *d_out, void *stream) {
    if (n > 0 && (!d_rgb || !d_out)) return fail(RTK_ERR_INVALID, "null buffer");
    if (n > (size_t(1) << 38)) return fail(RTK_ERR_INVALID, "too many values for one launch");
    const hipError_t e = launch_to_rgb8(d_rgb, n, d_out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch k_to_rgb8");
    return RTK_OK;
}

int rtk_format_ppm_rgb8(const uint8_t *rgb8, int32_t width, int32_t height, char *buf, size_t cap, size_t *n) {
    if (!rgb8 || !n || width <= 0 || height <= 0) return fail(RTK_ERR_INVALID, "bad image");
    try {
        const std::string s = format_ppm_rgb8(rgb8, width, height);
        *n = s.size();
        if (buf) std::memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

int rtk_write_ppm(const float *rgb, int32_t width, int32_t height, const char *path) {
    if (!rgb || !path || width <= 0 || height <= 0) return fail(RTK_ERR_INVALID, "bad image or path");
    try {
        const std::string s = format_ppm(rgb, width, height);
        std::FILE *f = std::fopen(path, "wb");
        if (!f) return fail(RTK_ERR_IO, std::string("cannot open ") + path);
        const size_t w = std::fwrite(s.data(), 1, s.size(), f);
        const int c = std::fclose(f);
        if (w != s.size() || c != 0) return fail(RTK_ERR_IO, std::string("short write to ") + path);
        return RTK_OK;
    } catch (const std::exception &e) { return fail(RTK_ERR_INVALID, e.what()); }
}

}  // extern "C"
