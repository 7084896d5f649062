// hip_accel.hpp — header-only adapter that models the reference's `accelerator` concept
// (render/accel/accel.hpp:8-12) on top of the rtk C-ABI (include/rtk.h).
//
// Drop it next to the reference's headers and change ONE line of src/main.cpp:
//     using A = kd_tree_simd_accel<F, static_cast<F>(epsilon)>;      // src/main.cpp:37
// to  using A = hip_accel<F, static_cast<F>(epsilon)>;
// Everything the reference's callers need is here: a constructor from std::shared_ptr<const scene<F>>
// (src/main.cpp:41, kd_tree_simd.hpp:100), the public `scene_ptr` member that render_frame / color_hit /
// is_occluded dereference (render/render.hpp:21,113,136), and
//     template <bool cull> std::optional<hit<F>> intersect(const ray3<F>&) const noexcept;
// `intersect` is one synchronous one-ray launch (correct, slow — it exists so the reference's own CPU
// render_frame can drive the GPU tree ray by ray).  The fast paths are `intersect_batch` (one ray per lane)
// and `render_frame` (the whole render loop device-side, replacing render/render.hpp:18-108).
#pragma once

#include <cstdio>
#include <exception>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <variant>
#include <vector>

#include <raytracer/core/math/ray3.hpp>
#include <raytracer/render/hit.hpp>
#include <raytracer/scene/scene.hpp>

#include "rtk.h"

template <typename F, F eps, std::size_t max_depth = 8, std::size_t max_leaf_size = 64>
struct hip_accel {
    static_assert(std::is_same_v<F, float>, "the rtk engine computes in float (the reference instantiates F = float, src/main.cpp:36)");

    std::shared_ptr<const scene<F>> scene_ptr;

    explicit hip_accel(std::shared_ptr<const scene<F>> scene_ptr_, bool normalize_hit_normal = true, int device = -1)
        : scene_ptr(std::move(scene_ptr_)) {
        const scene<F> &sc = *scene_ptr;
        // flatten scene<F> exactly as parse_scene_file laid it out (io/json/loader.hpp:235-265)
        std::vector<int32_t> mesh_material, mesh_nverts, mesh_ntris, mat_kind, mat_smooth, mat_texture, mesh_has_uvs, tex_kind;
        std::vector<float> vertices, mat_albedo, mat_ior, light_pos, light_intensity, uvs, tex_a, tex_b, tex_param;
        std::vector<uint32_t> indices;
        std::vector<uint8_t> tex_pixels;                  // bitmap_texture::texture back as the bytes stbi_load gave (bitmap.hpp:26-28)
        std::vector<int32_t> tex_bitmap;                  // [n_textures][3] byte offset, width, height
        // scene.textures is keyed by name (scene/scene.hpp:18); the C-ABI refers to textures by index
        std::vector<std::string> texture_names;
        for (const auto &[name, tv] : sc.textures) {
            float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f}, prm = 0.f;
            int kind = -1;
            int32_t bmp[3] = {0, 0, 0};
            std::visit([&](const auto &t) {
                using T = std::decay_t<decltype(t)>;
                if constexpr (std::is_same_v<T, albedo_texture<F>>) { kind = RTK_TEX_ALBEDO; a[0] = t.albedo.red; a[1] = t.albedo.green; a[2] = t.albedo.blue; }
                else if constexpr (std::is_same_v<T, edge_texture<F>>) {
                    kind = RTK_TEX_EDGES; prm = t.edge_width;
                    a[0] = t.edge_color.red; a[1] = t.edge_color.green; a[2] = t.edge_color.blue;
                    b[0] = t.inner_color.red; b[1] = t.inner_color.green; b[2] = t.inner_color.blue;
                } else if constexpr (std::is_same_v<T, checker_texture<F>>) {
                    kind = RTK_TEX_CHECKER; prm = t.square_size;
                    a[0] = t.color_a.red; a[1] = t.color_a.green; a[2] = t.color_a.blue;
                    b[0] = t.color_b.red; b[1] = t.color_b.green; b[2] = t.color_b.blue;
                } else {                                  // bitmap_texture (scene/texture/bitmap.hpp:40-44)
                    kind = RTK_TEX_BITMAP;
                    const std::size_t h = t.texture.get_height(), w = t.texture.get_width();
                    bmp[0] = static_cast<int32_t>(tex_pixels.size()); bmp[1] = static_cast<int32_t>(w); bmp[2] = static_cast<int32_t>(h);
                    for (std::size_t r = 0; r < h; ++r) for (std::size_t c = 0; c < w; ++c) {
                        const auto &px = t.texture.get_pixel(r, c);     // F(byte) * F(1.0 / 255.0), bitmap.hpp:19-28: invert exactly
                        const F ch[3] = {px.red, px.green, px.blue};
                        for (F v : ch) tex_pixels.push_back(static_cast<uint8_t>(v * F(255) + F(0.5)));
                    }
                }
            }, tv);
            tex_bitmap.insert(tex_bitmap.end(), bmp, bmp + 3);
            texture_names.push_back(name);
            tex_kind.push_back(kind); tex_param.push_back(prm);
            tex_a.insert(tex_a.end(), a, a + 3); tex_b.insert(tex_b.end(), b, b + 3);
        }
        for (const auto &mesh : sc.meshes) {
            mesh_material.push_back(static_cast<int32_t>(mesh.material_idx));
            mesh_nverts.push_back(static_cast<int32_t>(mesh.vertices.size()));
            mesh_ntris.push_back(static_cast<int32_t>(mesh.triangles.size()));
            for (const auto &v : mesh.vertices) { vertices.push_back(v.x); vertices.push_back(v.y); vertices.push_back(v.z); }
            const bool has_uvs = mesh.uvs.size() >= mesh.vertices.size() && !mesh.uvs.empty();
            mesh_has_uvs.push_back(has_uvs ? 1 : 0);
            if (has_uvs) for (std::size_t i = 0; i < mesh.vertices.size(); ++i) { uvs.push_back(mesh.uvs[i].x); uvs.push_back(mesh.uvs[i].y); }
            for (const auto &t : mesh.triangles)
                for (int k = 0; k < 3; ++k) indices.push_back(static_cast<uint32_t>(t.vertex_indices[k]));
            first_triangle_.push_back(n_triangles_);
            n_triangles_ += mesh.triangles.size();
        }
        for (const auto &mv : sc.materials) {
            float albedo[3] = {0.f, 0.f, 0.f};
            float ior = 1.f;
            int kind = RTK_MAT_DIFFUSE, smooth = 0, texture = -1;
            std::visit([&](const auto &m) {
                using M = std::decay_t<decltype(m)>;
                smooth = m.smooth_shading ? 1 : 0;
                if constexpr (std::is_same_v<M, diffuse_material<F>>) kind = RTK_MAT_DIFFUSE;
                else if constexpr (std::is_same_v<M, reflective_material<F>>) kind = RTK_MAT_REFLECTIVE;
                else if constexpr (std::is_same_v<M, refractive_material<F>>) { kind = RTK_MAT_REFRACTIVE; ior = m.ior; }
                else if constexpr (std::is_same_v<M, constant_material<F>>) kind = RTK_MAT_CONSTANT;
                else {                                    // texture_material (scene/material/texture.hpp)
                    kind = RTK_MAT_TEXTURE;
                    for (std::size_t ti = 0; ti < texture_names.size(); ++ti) if (texture_names[ti] == m.texture) texture = static_cast<int>(ti);
                    if (texture < 0) throw std::invalid_argument("hip_accel: material uses unknown texture '" + m.texture + "'");
                }
                if constexpr (requires { m.albedo; }) { albedo[0] = m.albedo.red; albedo[1] = m.albedo.green; albedo[2] = m.albedo.blue; }
            }, mv);
            mat_kind.push_back(kind); mat_smooth.push_back(smooth); mat_ior.push_back(ior); mat_texture.push_back(texture);
            mat_albedo.insert(mat_albedo.end(), albedo, albedo + 3);
        }
        for (const auto &l : sc.lights) {
            light_pos.push_back(l.position.x); light_pos.push_back(l.position.y); light_pos.push_back(l.position.z);
            light_intensity.push_back(l.intensity);
        }
        rtk_scene_desc d{};
        d.n_meshes = static_cast<int32_t>(mesh_material.size());
        d.mesh_material = mesh_material.data(); d.mesh_nverts = mesh_nverts.data(); d.mesh_ntris = mesh_ntris.data();
        d.vertices = vertices.data(); d.indices = indices.data();
        d.n_materials = static_cast<int32_t>(mat_kind.size());
        d.mat_kind = mat_kind.data(); d.mat_albedo = mat_albedo.data(); d.mat_ior = mat_ior.data(); d.mat_smooth = mat_smooth.data();
        d.mat_texture = mat_texture.data(); d.uvs = uvs.data(); d.mesh_has_uvs = mesh_has_uvs.data();
        d.n_textures = static_cast<int32_t>(tex_kind.size());
        d.tex_kind = tex_kind.data(); d.tex_color_a = tex_a.data(); d.tex_color_b = tex_b.data(); d.tex_param = tex_param.data();
        d.tex_pixels = tex_pixels.data(); d.tex_bitmap = tex_bitmap.data();
        d.n_lights = static_cast<int32_t>(light_intensity.size());
        d.light_pos = light_pos.data(); d.light_intensity = light_intensity.data();
        d.cam_pos[0] = sc.viewpoint.position.x; d.cam_pos[1] = sc.viewpoint.position.y; d.cam_pos[2] = sc.viewpoint.position.z;
        for (int i = 0; i < 9; ++i) d.cam_mat[i] = sc.viewpoint.matrix.m[static_cast<std::size_t>(i)];
        d.background[0] = sc.config.background_color.red; d.background[1] = sc.config.background_color.green;
        d.background[2] = sc.config.background_color.blue;
        d.width = static_cast<int32_t>(sc.config.image_width); d.height = static_cast<int32_t>(sc.config.image_height);
        d.bucket_size = static_cast<int32_t>(sc.config.bucket_size);

        rtk_scene *rs = nullptr;
        check(rtk_scene_create(&d, &rs));
        rtk_accel_params ap{};
        ap.max_depth = static_cast<int32_t>(max_depth); ap.max_leaf_size = static_cast<int32_t>(max_leaf_size);
        ap.eps = eps; ap.normalize_hit_normal = normalize_hit_normal ? 1 : 0; ap.device = device;
        rtk_accel *ra = nullptr;
        const int rc = rtk_accel_build(rs, &ap, &ra);
        rtk_scene_destroy(rs);                       // the accel keeps its own copy (as kd_tree_simd.hpp:106-107 does)
        check(rc);
        accel_ = std::shared_ptr<rtk_accel>(ra, rtk_accel_destroy);
    }

    // accel.template intersect<cull>(ray) — render/accel/accel.hpp:9-10
    template <bool backface_culling>
    [[nodiscard]] std::optional<hit<F>> intersect(const ray3<F> &ray) const noexcept {
        rtk_ray r{{ray.origin.x, ray.origin.y, ray.origin.z}, {ray.direction.x, ray.direction.y, ray.direction.z}};
        rtk_hit h{};
        // intersect is noexcept in the reference (kd_tree_simd.hpp:188) and has no error channel: a device failure must not be
        // read as "the ray missed" (a silently black frame), so it ends the program with the library's message.
        if (rtk_accel_intersect(accel_.get(), &r, 1, backface_culling ? 1 : 0, RTK_TRACE_AUTO, &h) != RTK_OK) {
            std::fprintf(stderr, "hip_accel::intersect: %s\n", rtk_last_error());
            std::terminate();
        }
        return to_hit(ray, h);
    }

    // one ray per lane; out[i] corresponds to rays[i]
    template <bool backface_culling>
    [[nodiscard]] std::vector<std::optional<hit<F>>> intersect_batch(const std::vector<ray3<F>> &rays) const {
        std::vector<rtk_ray> in(rays.size());
        for (std::size_t i = 0; i < rays.size(); ++i)
            in[i] = rtk_ray{{rays[i].origin.x, rays[i].origin.y, rays[i].origin.z},
                            {rays[i].direction.x, rays[i].direction.y, rays[i].direction.z}};
        std::vector<rtk_hit> out(rays.size());
        check(rtk_accel_intersect(accel_.get(), in.data(), in.size(), backface_culling ? 1 : 0, RTK_TRACE_AUTO, out.data()));
        std::vector<std::optional<hit<F>>> res(rays.size());
        for (std::size_t i = 0; i < rays.size(); ++i) res[i] = to_hit(rays[i], out[i]);
        return res;
    }

    // render_frame<A,F>(accel, BUCKET_TILES) with the whole loop device-side; pixels [h][w] as in image<F>
    [[nodiscard]] std::vector<std::vector<color<F>>> render_frame(const rtk_render_params &params, rtk_counters *counters = nullptr) const {
        // this returns a finished image: a partial pass of a progressive frame (sample_begin / sample_count) needs the running
        // sums of the passes before it, which live in the caller's buffer -- use rtk_render_frame / _device for those
        if (params.sample_begin != 0 || (params.sample_count != 0 && params.sample_count != params.spp))
            throw std::invalid_argument("hip_accel::render_frame renders whole frames (sample_begin = 0, all spp samples)");
        std::size_t n = 0;
        check(rtk_render_output_floats(accel_.get(), &params, &n));
        std::vector<float> rgb(n);
        check(rtk_render_frame(accel_.get(), &params, rgb.data(), counters));
        const std::size_t w = params.width > 0 ? static_cast<std::size_t>(params.width) : scene_ptr->config.image_width;
        const std::size_t h = n / 3 / w;
        std::vector<std::vector<color<F>>> px(h, std::vector<color<F>>(w));
        for (std::size_t y = 0; y < h; ++y)
            for (std::size_t x = 0; x < w; ++x)
                px[y][x] = color<F>{rgb[(y * w + x) * 3], rgb[(y * w + x) * 3 + 1], rgb[(y * w + x) * 3 + 2]};
        return px;
    }

    // config.hpp:6-17 defaults
    [[nodiscard]] static rtk_render_params default_params() noexcept {
        rtk_render_params p{};
        p.spp = 1; p.max_ray_depth = 5; p.diffuse_rays = 0; p.seed = 42; p.fov_degrees = 90.0;
        p.shadow_bias = 1e-4f; p.reflection_bias = 1e-4f; p.refraction_bias = 1e-4f;
        p.trace_mode = RTK_TRACE_AUTO; p.world_size = 1;
        return p;
    }

    [[nodiscard]] rtk_accel *handle() const noexcept { return accel_.get(); }

private:
    std::shared_ptr<rtk_accel> accel_;
    std::vector<std::size_t> first_triangle_;     // global triangle index of each mesh's first triangle
    std::size_t n_triangles_ = 0;

    static void check(int rc) {
        if (rc != RTK_OK) throw std::runtime_error(std::string("rtk: ") + rtk_last_error());
    }

    // rebuilds hit<F> (render/hit.hpp:9-21) from the compact record, as kd_tree_simd.hpp:252-263 fills it
    [[nodiscard]] std::optional<hit<F>> to_hit(const ray3<F> &ray, const rtk_hit &h) const noexcept {
        if (h.tri == 0xFFFFFFFFu) return std::nullopt;
        const auto &mesh = scene_ptr->meshes[h.mesh];
        const auto &tri = mesh.triangles[h.tri - first_triangle_[h.mesh]];
        return hit<F>{
            ray,
            ray.origin + (h.t * ray.direction),
            vec3<F>{h.normal[0], h.normal[1], h.normal[2]},
            tri.normal,
            tri.uvs,
            h.t,
            h.u,
            h.v,
            static_cast<F>(1.) - h.u - h.v,
            static_cast<std::size_t>(h.mesh)
        };
    }
};
