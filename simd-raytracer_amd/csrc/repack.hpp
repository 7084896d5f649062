// Ray repacking for batched intersect (repack.hip): bounds + coherence probe, keys, radix sort of (key, index).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/rtk.h"

namespace rtk {

constexpr int kRepackBoundsWords = 22;   // 6 minima, 6 maxima (monotone uint images of floats), wide waves, sampled waves, sum of origin extents,
                                         // [15] raster width, [16] verdict: 1 = sort the batch, [17] coordinates that vary (k_raster_probe),
                                         // [18..19] minima of the directions' octahedral coordinates, [20..21] maxima
constexpr int kRepackRowWords = 20;      // a workgroup's row of partial results (k_ray_bounds)

constexpr int kRepackMaxBlocks = 2048;   // workgroups of k_ray_bounds; each leaves kRepackRowWords words behind the bounds for the fold
constexpr size_t kRepackBoundsAlloc = size_t(kRepackBoundsWords) + size_t(kRepackRowWords) * size_t(kRepackMaxBlocks);   // words to allocate for d_bounds

// d_bounds[kRepackBoundsAlloc]; every wave_stride-th wave of 64 rays is looked at; probe: the coherence counts as well.
// unfolded_blocks != nullptr: the workgroups' rows are left for launch_raster_probe to fold (a launch less)
hipError_t launch_ray_bounds(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, uint32_t wave_stride, bool probe, hipStream_t s,
                             unsigned *unfolded_blocks = nullptr);
// behind launch_ray_bounds' probe: d_bounds[16], [17] = the verdict (see above); d_bounds[15] = width of the raster the batch is
// (rows of camera rays), 0 if it is none or want_raster is false
hipError_t launch_raster_probe(const rtk_ray *d_rays, size_t n, uint32_t *d_bounds, bool want_raster, hipStream_t s, unsigned fold_blocks = 0);
hipError_t repack_temp_bytes(size_t n, size_t *bytes);
// keys (and the identity permutation) from the bounds; only_if_sort: the kernel does nothing unless the probe's verdict (d_bounds[16]) says sort
hipError_t launch_ray_keys(const rtk_ray *d_rays, size_t n, const uint32_t *d_bounds, uint32_t *d_keys, uint32_t *d_idx, hipStream_t s,
                           bool dirs3 = false, bool only_if_sort = false);
// sort on key bits [begin_bit, 30); the sorted ray indices end up in d_idx[n .. 2n)
hipError_t launch_key_sort(size_t n, uint32_t *d_keys, uint32_t *d_idx, void *d_temp, size_t temp_bytes, hipStream_t s, unsigned begin_bit = 0);

}  // namespace rtk
