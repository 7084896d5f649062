// TEST-ONLY stand-in for scene/scene.hpp and what it includes: only the members hip_accel.hpp reads.
#pragma once
#include <array>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>
#include <raytracer/core/math/ray3.hpp>
template <typename F> struct color { F red, green, blue; };
template <typename F> struct mat3 { std::array<F, 9> m; };
template <typename F> struct diffuse_material { color<F> albedo; bool smooth_shading; };
template <typename F> struct reflective_material { color<F> albedo; bool smooth_shading; };
template <typename F> struct refractive_material { F ior; bool smooth_shading; };
template <typename F> struct constant_material { color<F> albedo; bool smooth_shading; };
template <typename F> struct texture_material { std::string texture; bool smooth_shading; };
template <typename F> using material_variant = std::variant<diffuse_material<F>, reflective_material<F>, refractive_material<F>,
                                                            constant_material<F>, texture_material<F>>;
template <typename F> struct albedo_texture { color<F> albedo; };
template <typename F> struct edge_texture { color<F> edge_color, inner_color; F edge_width; };
template <typename F> struct checker_texture { color<F> color_a, color_b; F square_size; };
template <typename F> struct image {                       // scene/image.hpp:7-33
    std::size_t height, width;
    std::vector<std::vector<color<F>>> pixels;
    std::size_t get_height() const { return height; }
    std::size_t get_width() const { return width; }
    const color<F> &get_pixel(std::size_t row, std::size_t column) const { return pixels[row][column]; }
};
template <typename F> struct bitmap_texture { image<F> texture; };     // scene/texture/bitmap.hpp:40-44
template <typename F> using texture_variant = std::variant<albedo_texture<F>, edge_texture<F>, checker_texture<F>, bitmap_texture<F>>;
template <typename F> struct triangle {
    vec3<F> v0, v1, v2, e1, e2, normal;
    std::array<std::size_t, 3> vertex_indices;
    std::size_t mesh_idx;
    vec3<vec2<F>> uvs;
};
template <typename F> struct mesh_object {
    std::size_t material_idx;
    std::vector<vec3<F>> vertices;
    std::vector<vec2<F>> uvs;
    std::vector<triangle<F>> triangles;
};
template <typename F> struct light { vec3<F> position; F intensity; };
template <typename F> struct camera { vec3<F> position; mat3<F> matrix; };
template <typename F> struct settings { color<F> background_color; std::size_t image_height, image_width, bucket_size; };
template <typename F> struct scene {
    settings<F> config;
    camera<F> viewpoint;
    std::vector<light<F>> lights;
    std::unordered_map<std::string, texture_variant<F>> textures;
    std::vector<material_variant<F>> materials;
    std::vector<mesh_object<F>> meshes;
};
