// TEST-ONLY stand-in for render/hit.hpp: same field order as the reference's hit<F> aggregate.
#pragma once
#include <raytracer/core/math/ray3.hpp>
template <typename F> struct hit {
    ray3<F> ray;
    vec3<F> position, hit_normal, face_normal;
    vec3<vec2<F>> uvs;
    F distance, u, v, w;
    std::size_t mesh_idx;
};
