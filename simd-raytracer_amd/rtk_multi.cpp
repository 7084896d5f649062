// Multi-GPU frames from C++: one process per GPU, buckets dealt round-robin to the ranks (bucket i -> rank i % world, the
// reference's bucket_schedule, render/tile/bucket.hpp:7-21, spread over GPUs instead of threads), ONE collective -- an RCCL
// all-gather of the rank-local bucket buffers over xGMI -- and rtk_tiles_assemble_device on every rank.  This is the C++
// twin of simd-raytracer_amd/parallel.py; the engine library itself (librtk_hip.so) stays free of RCCL.
//
// Process model: `rtk_render --world N` forks N rank processes BEFORE anything touches the GPU (the launcher process never
// initialises HIP), rank r takes device r.  Rank 0 creates the ncclUniqueId and publishes it through a
// file (write to a temporary name, then rename: readers never see a partial id).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
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

bool read_id(const std::string &path, ncclUniqueId &id) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t n = std::fread(&id, 1, sizeof(id), f);
    std::fclose(f);
    return n == sizeof(id);
}

}  // namespace

// The launcher: forks `world` children BEFORE anything touches the GPU (the parent never initialises HIP and only waits).
// Returns the rank (0..world-1) in a child, and -1 - status in the parent once every child has exited.
int rtk_multi_fork(int world, std::string &id_path) {
    char tmpl[] = "/tmp/rtk_nccl_id_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) { std::perror("rtk_render: mkstemp"); return -2; }
    close(fd);
    unlink(tmpl);                                      // rank 0 creates it (atomically, by rename) once the id exists
    id_path = tmpl;
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);      // dmabuf IPC between the ranks' processes
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("rtk_render: fork"); return -2; }
        if (pid == 0) return r;
        kids.push_back(pid);
    }
    int rc = 0;
    for (const pid_t pid : kids) {
        int st = 0;
        if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    unlink(id_path.c_str());
    return -1 - rc;
}

// One rank: renders `frames` frames of its buckets, gathers, assembles; rank 0 returns the last frame in `rgb_out`.
int rtk_multi_rank(rtk_accel *accel, rtk_render_params p, int rank, int world, const char *id_path, int frames,
                   std::vector<float> &rgb_out, double &best_seconds, unsigned long long &rays_total) {
    int n_dev = 0;
    MG_HIP(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { std::fprintf(stderr, "rtk_render[%d]: no HIP device\n", rank); return 1; }
    MG_HIP(hipSetDevice(rank % n_dev));
    ncclUniqueId id;
    if (rank == 0) {
        MG_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = std::string(id_path) + ".tmp";
        std::FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0 || std::rename(tmp.c_str(), id_path) != 0) {
            std::fprintf(stderr, "rtk_render[0]: cannot publish the RCCL id at %s\n", id_path);
            return 1;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!read_id(id_path, id)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                std::fprintf(stderr, "rtk_render[%d]: timed out waiting for the RCCL id\n", rank);
                return 1;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
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
    if (rank == 0) {
        rgb_out.resize(n_frame);
        MG_HIP(hipMemcpy(rgb_out.data(), d_frame, n_frame * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_rays); (void)hipFree(d_local); (void)hipFree(d_gathered); (void)hipFree(d_frame);
    MG_NCCL(ncclCommDestroy(comm));
    (void)hipStreamDestroy(stream);
    return 0;
}
