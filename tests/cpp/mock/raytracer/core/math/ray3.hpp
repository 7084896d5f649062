// TEST-ONLY stand-in with the member names hip_accel.hpp touches on the reference's vec3/ray3
// (core/math/vec3.hpp, core/math/ray3.hpp).  Not a copy of the reference: just the data members and the two
// operators the adapter uses, so the adapter can be compile-checked and run without the reference tree.
#pragma once
#include <cstddef>
template <typename F> struct vec2 { F x, y; };
template <typename F> struct vec3 { F x, y, z; };
template <typename F> vec3<F> operator+(const vec3<F> &a, const vec3<F> &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename F> vec3<F> operator*(F s, const vec3<F> &a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename F> struct ray3 {
    vec3<F> origin, direction, inv_direction;
    ray3(const vec3<F> &o, const vec3<F> &d) : origin(o), direction(d), inv_direction{F(1) / d.x, F(1) / d.y, F(1) / d.z} {}
};
