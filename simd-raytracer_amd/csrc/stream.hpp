// Argument blocks of the streaming frame pipeline (stream.hip).
#pragma once

#include "kernels.hpp"

namespace rtk {
namespace dev {

struct PathRay {            // 32 B: a reflection ray waiting for depth level d
    float o[3];
    uint32_t pixel;         // output index (in pixels) this path ends in
    float d[3];
    uint32_t pad;
};
struct HitRec {             // 32 B: a diffuse shading point waiting for its light loop
    float P[3];
    uint32_t pixel;
    float ncos[3];          // the normal the cosine law uses (hit_normal if smooth_shading, else face_normal)
    uint32_t mat;
};
static_assert(sizeof(PathRay) == 32 && sizeof(HitRec) == 32, "queue records are two 16-byte stores");

// control block (u32 words): queue fill counts per depth level
constexpr uint32_t kCtrlPathCount = 0;                      // [kMaxRayDepth + 2]
constexpr uint32_t kCtrlHitCount = kMaxRayDepth + 2;        // [kMaxRayDepth + 2]
constexpr uint32_t kCtrlTicket = 2 * (kMaxRayDepth + 2);     // [2][kMaxRayDepth + 2] dynamic work-unit tickets (path, shadow)
constexpr uint32_t kCtrlDebug = 4 * (kMaxRayDepth + 2);      // -DRTK_DEBUG_WAVE_TIME: [stage 0..2][level 0..3]{sum, max, n, block max}
constexpr uint32_t kCtrlWords = kCtrlDebug + 3 * 4 * 4;

struct StreamWs {
    PathRay *path[2];       // ping-pong by level parity, capacity = pixels of this rank
    HitRec *hits;           // capacity = pixels of this rank (reused by every level)
    float2 *contrib;        // [capacity * n_lights] {contribution, unoccluded}
    float *sumbuf;          // [capacity * 3] running sample sum (spp > 1 only)
    uint32_t *ctrl;
};

struct StreamArgs {
    RenderArgs r;
    StreamWs ws;
    uint32_t level;
    int sample;
    uint32_t auto_min_lanes;   // RTK_TRACE_AUTO: leave the wave-cooperative walk when fewer rays than this share a node
};

}  // namespace dev

hipError_t launch_stream_sample(const dev::StreamArgs &base, bool stats, int slices, hipStream_t s);

}  // namespace rtk
