// Multi-GPU host path of rtk_render (rtk_multi.cpp): one process per GPU, RCCL all-gather of the rank-local bucket buffers.
#pragma once

#include <vector>

#include "rtk.h"

// how the RCCL id travels from rank 0 to the others: pipes opened by the launcher before the fork
struct rtk_multi_link {
    int id_read_fd = -1;               // ranks > 0: read end of this rank's pipe
    std::vector<int> id_write_fds;     // rank 0: write ends, one per other rank
};

// forks `world` rank processes before any GPU call; returns the rank in a child, -1 - status in the parent after all children
// exited (the first child that fails terminates the others)
int rtk_multi_fork(int world, rtk_multi_link &link);
// renders this rank's buckets `frames` times, gathers and assembles; rank 0 gets the last frame in rgb_out; best_seconds is
// the slowest rank's best frame
int rtk_multi_rank(rtk_accel *accel, rtk_render_params p, int rank, int world, const rtk_multi_link &link, int frames,
                   std::vector<float> &rgb_out, double &best_seconds, unsigned long long &rays_total);
