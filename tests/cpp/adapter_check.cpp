// Compiles simd-raytracer_amd/hip_accel.hpp against test-only stand-ins for the reference's types and drives it
// the way the reference's callers do (concept check, ctor from shared_ptr<const scene>, intersect<true/false>).
// Prints one line per query for tests/test_cpp_host.py to compare with the oracle.
#include <concepts>
#include <cstdio>
#include <memory>

#include "hip_accel.hpp"

template <typename A, typename F>
concept accelerator = requires(A accel, const ray3<F> &ray) {            // the shape of render/accel/accel.hpp:8-12
    { accel.template intersect<true>(ray) } -> std::same_as<std::optional<hit<F>>>;
    { accel.template intersect<false>(ray) } -> std::same_as<std::optional<hit<F>>>;
};

int main() {
    using F = float;
    using A = hip_accel<F, 1e-6f>;
    static_assert(accelerator<A, F>);
    scene<F> sc{};
    sc.config = {{0.f, 0.5f, 0.f}, 16, 16, 64};
    sc.viewpoint = {{0.f, 0.f, 0.f}, {{1, 0, 0, 0, 1, 0, 0, 0, 1}}};
    sc.lights.push_back({{0.f, 2.f, 0.f}, 100.f});
    sc.materials.push_back(diffuse_material<F>{{1.f, 1.f, 0.f}, false});
    // a bitmap texture as load_bitmap leaves it (scene/texture/bitmap.hpp:19-28): F(byte) * F(1.0 / 255.0); the adapter hands the bytes on
    const F cs = F(1.0 / 255.0);
    sc.textures.emplace("bm", bitmap_texture<F>{image<F>{2, 2, {{{F(255) * cs, F(0) * cs, F(1) * cs}, {F(2) * cs, F(127) * cs, F(128) * cs}},
                                                                 {{F(254) * cs, F(3) * cs, F(85) * cs}, {F(170) * cs, F(200) * cs, F(33) * cs}}}}});
    mesh_object<F> m{};
    m.material_idx = 0;
    m.vertices = {{-1.75f, -1.75f, -3.f}, {1.75f, -1.75f, -3.f}, {0.f, 1.75f, -3.f}};
    triangle<F> t{};
    t.v0 = m.vertices[0]; t.v1 = m.vertices[1]; t.v2 = m.vertices[2];
    t.normal = {0.f, 0.f, 1.f};
    t.vertex_indices = {0, 1, 2};
    t.mesh_idx = 0;
    m.triangles.push_back(t);
    sc.meshes.push_back(m);
    A accel(std::make_shared<const scene<F>>(sc));
    const ray3<F> hit_ray({0.f, 0.f, 0.f}, {0.f, 0.f, -1.f}), miss_ray({0.f, 0.f, 0.f}, {0.f, 1.f, 0.f});
    const auto h = accel.intersect<true>(hit_ray);
    const auto mss = accel.intersect<false>(miss_ray);
    std::printf("single hit=%d t=%.9g u=%.9g v=%.9g w=%.9g mesh=%zu n=(%.9g,%.9g,%.9g) pos=(%.9g,%.9g,%.9g) miss=%d\n", h.has_value(),
                h ? h->distance : -1.f, h ? h->u : 0.f, h ? h->v : 0.f, h ? h->w : 0.f, h ? h->mesh_idx : 0, h ? h->hit_normal.x : 0.f,
                h ? h->hit_normal.y : 0.f, h ? h->hit_normal.z : 0.f, h ? h->position.x : 0.f, h ? h->position.y : 0.f,
                h ? h->position.z : 0.f, !mss.has_value());
    const auto batch = accel.intersect_batch<false>({hit_ray, miss_ray, hit_ray});
    std::printf("batch %d %d %d\n", batch[0].has_value(), batch[1].has_value(), batch[2].has_value());
    auto p = A::default_params();
    rtk_counters c{};
    const auto px = accel.render_frame(p, &c);
    std::printf("frame %zux%zu rays=%llu centre=(%.9g,%.9g,%.9g) corner=(%.9g,%.9g,%.9g)\n", px[0].size(), px.size(),
                (unsigned long long)c.rays, px[8][8].red, px[8][8].green, px[8][8].blue, px[0][0].red, px[0][0].green, px[0][0].blue);
    return 0;
}
