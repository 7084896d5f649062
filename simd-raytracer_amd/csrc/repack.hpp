// Ray repacking for batched intersect (repack.hip): bounds + coherence probe, keys, radix sort of (key, index).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/rtk.h"

namespace rtk {

constexpr int kRepackBoundsWords = 18;   // 6 minima, 6 maxima (monotone uint images of floats), wide waves, sampled waves, sum of origin extents,
                                         // [15] raster width, [16] verdict: 1 = sort the batch, [17] coordinates that vary (k_raster_probe)

struct RepackProbe {
    int active_dims;            // of origin xyz / direction xyz: how many vary at all over the probed rays
    uint32_t waves;             // probed
    float wide_dir_fraction;    // share of the probed waves whose own directions span more than 0.25
    float origin_spread;        // mean extent of a wave's origins / extent of all origins
};
RepackProbe decode_probe(const uint32_t *host_bounds);

// d_bounds[kRepackBoundsWords]; every wave_stride-th wave of 64 rays is looked at
hipError_t launch_ray_bounds(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, uint32_t wave_stride, hipStream_t s);
// behind launch_ray_bounds' probe: d_bounds[16], [17] = the verdict (see above); d_bounds[15] = width of the raster the batch is
// (rows of camera rays), 0 if it is none or want_raster is false
hipError_t launch_raster_probe(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, bool want_raster, hipStream_t s);
hipError_t repack_temp_bytes(size_t n, size_t *bytes);
// keys from the bounds, then sort on key bits [begin_bit, 30); the sorted ray indices end up in d_idx[n .. 2n)
hipError_t launch_ray_sort(const rtk_ray *d_rays, size_t n, const uint32_t *d_bounds, uint32_t *d_keys, uint32_t *d_idx, void *d_temp,
                           size_t temp_bytes, hipStream_t s, unsigned begin_bit = 0);

}  // namespace rtk
