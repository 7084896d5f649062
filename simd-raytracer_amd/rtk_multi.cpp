// Multi-GPU frames from C++: one process per GPU, buckets dealt round-robin to the ranks (bucket i -> rank i % world; diagonally when tiles_x % world == 0, see rtk.h; the
// reference's bucket_schedule, render/tile/bucket.hpp:7-21, spread over GPUs instead of threads), ONE collective -- an RCCL
// all-gather of the rank-local bucket buffers over xGMI -- and rtk_tiles_assemble_device on every rank.  This is the C++
// twin of simd-raytracer_amd/parallel.py; the engine library itself (librtk_hip.so) stays free of RCCL.
//
// Process model: `rtk_render --world N` forks N rank processes BEFORE anything touches the GPU (the launcher process never
// initialises HIP), rank r takes device r.  Rank 0 creates the ncclUniqueId and hands it to every other rank through a pipe
// the launcher opened before the fork: no path in the file system is shared.  The launcher waits for ANY child; the first one
// that fails takes the others down with it (SIGTERM), so a rank that dies never leaves its peers blocked in a collective.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rtk.h"
#include "rtk_multi.hpp"

namespace {

#define MG_HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "rtk_render[%d]: %s: %s\n", rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define MG_NCCL(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "rtk_render[%d]: %s: %s\n", rank, #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define MG_RTK(call) do { if ((call) != RTK_OK) { std::fprintf(stderr, "rtk_render[%d]: %s: %s\n", rank, #call, rtk_last_error()); return 1; } } while (0)

bool read_all(int fd, void *buf, size_t n) {
    char *p = static_cast<char *>(buf);
    while (n > 0) {
        const ssize_t got = read(fd, p, n);
        if (got <= 0) return false;                    // 0: rank 0 went away without sending the id
        p += got; n -= static_cast<size_t>(got);
    }
    return true;
}
bool write_all(int fd, const void *buf, size_t n) {
    const char *p = static_cast<const char *>(buf);
    while (n > 0) {
        const ssize_t put = write(fd, p, n);
        if (put <= 0) return false;
        p += put; n -= static_cast<size_t>(put);
    }
    return true;
}

}  // namespace

// The launcher: forks `world` children BEFORE anything touches the GPU (the parent never initialises HIP and only waits).
// Returns the rank (0..world-1) in a child, and -1 - status in the parent once every child has exited.
int rtk_multi_fork(int world, rtk_multi_link &link) {
    if (world < 1 || world > 64) { std::fprintf(stderr, "rtk_render: --world must be 1..64\n"); return -2; }
    std::vector<int> rd(static_cast<size_t>(world), -1), wr(static_cast<size_t>(world), -1);
    for (int r = 1; r < world; ++r) {                  // one pipe per rank > 0: rank 0 writes the id, rank r reads it
        int fds[2];
        if (pipe(fds) != 0) { std::perror("rtk_render: pipe"); return -2; }
        rd[static_cast<size_t>(r)] = fds[0]; wr[static_cast<size_t>(r)] = fds[1];
    }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);      // dmabuf IPC between the ranks' processes
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            std::perror("rtk_render: fork");
            for (const pid_t k : kids) kill(k, SIGTERM);
            return -2;
        }
        if (pid == 0) {
            signal(SIGPIPE, SIG_IGN);                  // a peer that died shows up as a failed write, not as a signal
            link.id_read_fd = -1; link.id_write_fds.clear();
            for (int q = 1; q < world; ++q) {
                if (r == 0) { close(rd[static_cast<size_t>(q)]); link.id_write_fds.push_back(wr[static_cast<size_t>(q)]); }
                else {
                    close(wr[static_cast<size_t>(q)]);
                    if (q == r) link.id_read_fd = rd[static_cast<size_t>(q)]; else close(rd[static_cast<size_t>(q)]);
                }
            }
            return r;
        }
        kids.push_back(pid);
    }
    for (int r = 1; r < world; ++r) { close(rd[static_cast<size_t>(r)]); close(wr[static_cast<size_t>(r)]); }
    int rc = 0;
    size_t left = kids.size();
    while (left > 0) {                                  // whichever child ends first is seen first
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) { rc = 1; break; }
        bool mine = false;
        for (pid_t &k : kids) if (k == pid) { k = -1; mine = true; }
        if (!mine) continue;
        left -= 1;
        if ((!WIFEXITED(st) || WEXITSTATUS(st) != 0) && rc == 0) {
            rc = 1;
            for (const pid_t k : kids) if (k > 0) kill(k, SIGTERM);   // its peers would wait for it in the next collective forever
        }
    }
    return -1 - rc;
}

// One rank: renders `frames` frames of its buckets, gathers, assembles; rank 0 returns the last frame in `rgb_out`.
int rtk_multi_rank(rtk_accel *accel, rtk_render_params p, int rank, int world, const rtk_multi_link &link, int frames,
                   std::vector<float> &rgb_out, double &best_seconds, unsigned long long &rays_total) {
    int n_dev = 0;
    MG_HIP(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { std::fprintf(stderr, "rtk_render[%d]: no HIP device\n", rank); return 1; }
    if (world > n_dev) {                               // RCCL refuses (or hangs on) two ranks of one communicator on one device
        if (rank == 0) std::fprintf(stderr, "rtk_render: --world %d but only %d HIP device(s)\n", world, n_dev);
        return 1;
    }
    MG_HIP(hipSetDevice(rank));
    ncclUniqueId id;
    if (rank == 0) {
        MG_NCCL(ncclGetUniqueId(&id));
        for (const int fd : link.id_write_fds) {
            if (!write_all(fd, &id, sizeof(id))) { std::fprintf(stderr, "rtk_render[0]: cannot send the RCCL id to a rank\n"); return 1; }
            close(fd);
        }
    } else {
        if (!read_all(link.id_read_fd, &id, sizeof(id))) { std::fprintf(stderr, "rtk_render[%d]: rank 0 ended before it sent the RCCL id\n", rank); return 1; }
        close(link.id_read_fd);
    }
    ncclComm_t comm;
    MG_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    MG_HIP(hipStreamCreate(&stream));

    p.rank = rank; p.world_size = world;
    size_t n_local = 0;
    MG_RTK(rtk_render_output_floats(accel, &p, &n_local));
    float *d_local = nullptr, *d_gathered = nullptr, *d_frame = nullptr;
    MG_HIP(hipMalloc(reinterpret_cast<void **>(&d_local), n_local * sizeof(float)));
    MG_HIP(hipMalloc(reinterpret_cast<void **>(&d_gathered), n_local * size_t(world) * sizeof(float)));
    // frame size: every rank knows width/height through the params or the scene; ask the library with a world-1 copy
    rtk_render_params whole = p; whole.rank = 0; whole.world_size = 1;
    size_t n_frame = 0;
    MG_RTK(rtk_render_output_floats(accel, &whole, &n_frame));
    MG_HIP(hipMalloc(reinterpret_cast<void **>(&d_frame), n_frame * sizeof(float)));

    unsigned long long *d_rays = nullptr;
    MG_HIP(hipMalloc(reinterpret_cast<void **>(&d_rays), sizeof(unsigned long long)));
    MG_HIP(hipMemset(d_rays, 0, sizeof(unsigned long long)));
    best_seconds = 1e30;
    for (int f = 0; f < (frames > 0 ? frames : 1); ++f) {
        MG_NCCL(ncclAllReduce(d_rays, d_rays, 1, ncclUint64, ncclSum, comm, stream));       // a barrier: every rank starts the frame together
        MG_HIP(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::high_resolution_clock::now();
        MG_RTK(rtk_render_frame_device(accel, &p, d_local, stream));
        if (world > 1) {
            MG_NCCL(ncclAllGather(d_local, d_gathered, n_local, ncclFloat, comm, stream));
            MG_RTK(rtk_tiles_assemble_device(accel, &p, d_gathered, d_frame, stream));
        } else {
            MG_HIP(hipMemcpyAsync(d_frame, d_local, n_frame * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
        MG_HIP(hipStreamSynchronize(stream));
        const double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        if (s < best_seconds) best_seconds = s;
    }
    rtk_counters c{};
    MG_RTK(rtk_render_last_counters(accel, &c));
    // total rays over the ranks (one 8-byte all-reduce)
    const unsigned long long mine = c.rays;
    MG_HIP(hipMemcpyAsync(d_rays, &mine, sizeof(mine), hipMemcpyHostToDevice, stream));
    MG_NCCL(ncclAllReduce(d_rays, d_rays, 1, ncclUint64, ncclSum, comm, stream));
    MG_HIP(hipMemcpyAsync(&rays_total, d_rays, sizeof(rays_total), hipMemcpyDeviceToHost, stream));
    MG_HIP(hipStreamSynchronize(stream));
    // the frame takes as long as its slowest rank: max over the ranks of each rank's best time
    static_assert(sizeof(double) == sizeof(unsigned long long), "the 8-byte scratch word carries the time as well");
    MG_HIP(hipMemcpyAsync(d_rays, &best_seconds, sizeof(double), hipMemcpyHostToDevice, stream));
    MG_NCCL(ncclAllReduce(d_rays, d_rays, 1, ncclDouble, ncclMax, comm, stream));
    MG_HIP(hipMemcpyAsync(&best_seconds, d_rays, sizeof(double), hipMemcpyDeviceToHost, stream));
    MG_HIP(hipStreamSynchronize(stream));
    if (rank == 0) {
        rgb_out.resize(n_frame);
        MG_HIP(hipMemcpy(rgb_out.data(), d_frame, n_frame * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_rays); (void)hipFree(d_local); (void)hipFree(d_gathered); (void)hipFree(d_frame);
    MG_NCCL(ncclCommDestroy(comm));
    (void)hipStreamDestroy(stream);
    return 0;
}
