// Multi-GPU host path of rtk_render (rtk_multi.cpp): one process per GPU, RCCL all-gather of the rank-local bucket buffers.
#pragma once

#include <string>
#include <vector>

#include "rtk.h"

// forks `world` rank processes before any GPU call; returns the rank in a child, -1 - status in the parent after all children exited
int rtk_multi_fork(int world, std::string &id_path);
// renders this rank's buckets `frames` times, gathers and assembles; rank 0 gets the last frame in rgb_out
int rtk_multi_rank(rtk_accel *accel, rtk_render_params p, int rank, int world, const char *id_path, int frames,
                   std::vector<float> &rgb_out, double &best_seconds, unsigned long long &rays_total);
