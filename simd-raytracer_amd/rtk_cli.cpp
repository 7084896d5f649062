// rtk_render — the reference's CLI (src/main.cpp:27-46: `./raytracer FILE` -> image.ppm, prints the render time)
// on top of the rtk C-ABI, with the reference's compile-time constants (config.hpp:6-17) as runtime flags.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtk.h"
#include "rtk_multi.hpp"

static int die(const char *what) {
    std::fprintf(stderr, "rtk_render: %s: %s\n", what, rtk_last_error());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::puts("Usage: ./rtk_render FILE [--width W] [--height H] [--spp N] [--depth D] [--diffuse K] [--seed S] [--fov DEG]\n"
                  "                         [--trace auto|lane|wave] [--no-normalize] [--frames N] [--out image.ppm]\n"
                  "                         [--world N]   one process per GPU, buckets dealt round-robin, RCCL all-gather of the frame\n"
                  "                         [--fast-traversal]   front-to-back leaf order (same distances, ties may pick another triangle: not the parity mode)");
        return 1;
    }
    rtk_render_params p{};
    p.spp = 1; p.max_ray_depth = 5; p.diffuse_rays = 0; p.seed = 42; p.fov_degrees = 90.0;
    p.shadow_bias = p.reflection_bias = p.refraction_bias = 1e-4f;
    p.trace_mode = RTK_TRACE_AUTO; p.world_size = 1;
    rtk_accel_params ap{8, 64, 1e-6f, 1, -1, RTK_TRAVERSAL_REFERENCE};
    std::string out = "image.ppm";
    int frames = 1, world = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "rtk_render: %s needs a value\n", name); std::exit(1); }
            return argv[++i];
        };
        if (a == "--width") p.width = std::atoi(val("--width"));
        else if (a == "--height") p.height = std::atoi(val("--height"));
        else if (a == "--spp") p.spp = std::atoi(val("--spp"));
        else if (a == "--depth") p.max_ray_depth = std::atoi(val("--depth"));
        else if (a == "--diffuse") p.diffuse_rays = std::atoi(val("--diffuse"));
        else if (a == "--seed") p.seed = static_cast<uint32_t>(std::strtoul(val("--seed"), nullptr, 10));
        else if (a == "--frames") frames = std::atoi(val("--frames"));
        else if (a == "--fast-traversal") ap.traversal = RTK_TRAVERSAL_FAST;      // front-to-back leaf order: NOT the parity mode (rtk.h)
        else if (a == "--world") world = std::atoi(val("--world"));
        else if (a == "--fov") p.fov_degrees = std::atof(val("--fov"));
        else if (a == "--out") out = val("--out");
        else if (a == "--no-normalize") ap.normalize_hit_normal = 0;
        else if (a == "--trace") {
            const std::string m = val("--trace");
            p.trace_mode = m == "lane" ? RTK_TRACE_LANE : (m == "wave" ? RTK_TRACE_WAVE : RTK_TRACE_AUTO);
        } else { std::fprintf(stderr, "rtk_render: unknown option %s\n", a.c_str()); return 1; }
    }
    // --world N: the rank processes are forked here, before this process has made any GPU call
    int rank = 0;
    rtk_multi_link link;
    if (world >= 1) {
        rank = rtk_multi_fork(world, link);
        if (rank < 0) return -1 - rank;                 // the launcher: every rank has exited
        ap.device = -1;
    }
    rtk_scene *scene = nullptr;
    if (rtk_scene_load_crtscene(argv[1], &scene) != RTK_OK) return die("scene");
    rtk_scene_info info{};
    rtk_scene_get_info(scene, &info);
    const int w = p.width > 0 ? p.width : info.width, h = p.height > 0 ? p.height : info.height;
    if (world >= 1) {
        ap.device = rank;                               // rank r renders on device r (rtk_multi_rank refuses world > devices)
    }
    rtk_accel *accel = nullptr;
    if (rtk_accel_build(scene, &ap, &accel) != RTK_OK) return die("accel");
    rtk_scene_destroy(scene);
    if (world >= 1) {
        std::vector<float> rgb;
        double best = 0.0;
        unsigned long long rays = 0;
        if (rtk_multi_rank(accel, p, rank, world, link, frames, rgb, best, rays) != 0) return 1;
        if (rank == 0) {
            std::printf("Rendering took %g seconds.\n", best);
            std::printf("%llu rays on %d GPUs, %.1f Mrays/s (render + RCCL all-gather + assemble, frame left on the device)\n", rays, world, double(rays) / best / 1e6);
            if (rtk_write_ppm(rgb.data(), w, h, out.c_str()) != RTK_OK) return die("write_ppm");
        }
        rtk_accel_destroy(accel);
        return 0;
    }
    size_t n = 0;
    if (rtk_render_output_floats(accel, &p, &n) != RTK_OK) return die("params");
    std::vector<float> rgb(n);
    rtk_counters c{};
    double best = 1e30;
    for (int f = 0; f < (frames > 0 ? frames : 1); ++f) {
        const auto t0 = std::chrono::high_resolution_clock::now();
        if (rtk_render_frame(accel, &p, rgb.data(), &c) != RTK_OK) return die("render");
        const double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        if (s < best) best = s;
    }
    std::printf("Rendering took %g seconds.\n", best);                                   // src/main.cpp:21
    std::printf("%llu rays (%llu primary), %.1f Mrays/s including the device->host copy\n",
                (unsigned long long)c.rays, (unsigned long long)c.primary, double(c.rays) / best / 1e6);
    if (rtk_write_ppm(rgb.data(), w, h, out.c_str()) != RTK_OK) return die("write_ppm");
    rtk_accel_destroy(accel);
    return 0;
}
