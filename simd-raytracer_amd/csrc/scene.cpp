// Host-side scene model: flattened description -> meshes with smooth vertex normals.
// Follows scene/object/mesh.hpp:23-44 and scene/primitive/triangle.hpp:20-30 operation by operation
// (this file is compiled with -ffp-contract=off; the normals feed hit_normal bit for bit).
#include <cfloat>
#include <cmath>
#include <cstring>

#include "rtk_internal.hpp"

namespace rtk {

namespace {

inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {                      // core/math/vec3.hpp:124-131
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 unit(const Vec3 &v) {                                      // core/math/vec3.hpp:104-108
    const float inv_length = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x * inv_length, v.y * inv_length, v.z * inv_length};
}

}  // namespace

void box_reset(Box &b) {                                               // core/math/aabb3.hpp:20-22
    b.mn = {FLT_MAX, FLT_MAX, FLT_MAX};
    b.mx = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
}

void box_grow(Box &b, const Vec3 &p) {                                 // core/math/aabb3.hpp:24-31
    b.mn.x = p.x < b.mn.x ? p.x : b.mn.x;
    b.mn.y = p.y < b.mn.y ? p.y : b.mn.y;
    b.mn.z = p.z < b.mn.z ? p.z : b.mn.z;
    b.mx.x = b.mx.x < p.x ? p.x : b.mx.x;
    b.mx.y = b.mx.y < p.y ? p.y : b.mx.y;
    b.mx.z = b.mx.z < p.z ? p.z : b.mx.z;
}

// Finishes a mesh whose vertices/indices are filled in: bounds + area-unweighted vertex normals.
int finish_mesh(HostMesh &m, std::string &err) {
    const size_t nt = m.indices.size() / 3;
    const size_t nv = m.vertices.size();
    for (uint32_t ix : m.indices) {
        if (ix >= nv) { err = "triangle references a vertex index out of range"; return RTK_ERR_INVALID; }
    }
    m.vertex_normals.assign(nv, Vec3{0.f, 0.f, 0.f});
    box_reset(m.box);
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t a = m.indices[t * 3], b = m.indices[t * 3 + 1], c = m.indices[t * 3 + 2];
        const Vec3 &v0 = m.vertices[a], &v1 = m.vertices[b], &v2 = m.vertices[c];
        box_grow(m.box, v0); box_grow(m.box, v1); box_grow(m.box, v2);
        const Vec3 fn = unit(cross(v1 - v0, v2 - v0));                 // mesh.hpp:34
        m.vertex_normals[a] = m.vertex_normals[a] + fn;                // mesh.hpp:36-38
        m.vertex_normals[b] = m.vertex_normals[b] + fn;
        m.vertex_normals[c] = m.vertex_normals[c] + fn;
    }
    for (Vec3 &n : m.vertex_normals) n = unit(n);                      // mesh.hpp:41-43 (unused vertices become NaN, as there)
    return RTK_OK;
}

int scene_from_desc(const rtk_scene_desc &d, rtk_scene &out, std::string &err) {
    if (d.n_meshes < 0 || d.n_materials < 0 || d.n_lights < 0) { err = "negative count in rtk_scene_desc"; return RTK_ERR_INVALID; }
    if (d.n_meshes > 0 && (!d.mesh_material || !d.mesh_nverts || !d.mesh_ntris)) { err = "null mesh arrays"; return RTK_ERR_INVALID; }
    out.meshes.clear();
    size_t voff = 0, toff = 0, uvoff = 0;
    for (int32_t mi = 0; mi < d.n_meshes; ++mi) {
        const int32_t nv = d.mesh_nverts[mi], nt = d.mesh_ntris[mi];
        if (nv < 0 || nt < 0) { err = "negative mesh size"; return RTK_ERR_INVALID; }
        if ((nv > 0 && !d.vertices) || (nt > 0 && !d.indices)) { err = "null vertex/index array"; return RTK_ERR_INVALID; }
        if (d.mesh_material[mi] < 0 || d.mesh_material[mi] >= d.n_materials) { err = "material_index out of range"; return RTK_ERR_INVALID; }
        HostMesh m;
        m.material = d.mesh_material[mi];
        m.vertices.resize(static_cast<size_t>(nv));
        for (int32_t i = 0; i < nv; ++i) {
            const float *p = d.vertices + (voff + static_cast<size_t>(i)) * 3;
            m.vertices[static_cast<size_t>(i)] = {p[0], p[1], p[2]};
        }
        m.indices.assign(d.indices + toff * 3, d.indices + (toff + static_cast<size_t>(nt)) * 3);
        if (d.mesh_has_uvs && d.mesh_has_uvs[mi] != 0) {
            if (!d.uvs) { err = "mesh_has_uvs set but uvs is null"; return RTK_ERR_INVALID; }
            m.uvs.assign(d.uvs + uvoff * 2, d.uvs + (uvoff + static_cast<size_t>(nv)) * 2);
            uvoff += static_cast<size_t>(nv);
        }
        const int rc = finish_mesh(m, err);
        if (rc != RTK_OK) return rc;
        out.meshes.push_back(std::move(m));
        voff += static_cast<size_t>(nv);
        toff += static_cast<size_t>(nt);
    }
    out.n_vertices = static_cast<int32_t>(voff);
    out.n_triangles = static_cast<int32_t>(toff);
    out.materials.resize(static_cast<size_t>(d.n_materials));
    for (int32_t i = 0; i < d.n_materials; ++i) {
        DevMaterial &m = out.materials[static_cast<size_t>(i)];
        std::memset(&m, 0, sizeof(m));
        m.kind = d.mat_kind[i];
        if (m.kind < RTK_MAT_DIFFUSE || m.kind > RTK_MAT_TEXTURE) { err = "material type unknown"; return RTK_ERR_INVALID; }
        m.texture = -1;
        if (m.kind == RTK_MAT_TEXTURE) {
            m.texture = d.mat_texture ? d.mat_texture[i] : -1;
            if (m.texture < 0 || m.texture >= d.n_textures) { err = "texture material refers to a texture that does not exist"; return RTK_ERR_INVALID; }
        }
        m.smooth = d.mat_smooth ? d.mat_smooth[i] : 0;
        if (d.mat_albedo) std::memcpy(m.albedo, d.mat_albedo + i * 3, sizeof(float) * 3);
        m.ior = d.mat_ior ? d.mat_ior[i] : 1.0f;
    }
    if (d.n_textures < 0) { err = "negative texture count"; return RTK_ERR_INVALID; }
    out.textures.resize(static_cast<size_t>(d.n_textures));
    out.tex_pixels.clear();
    for (int32_t i = 0; i < d.n_textures; ++i) {
        DevTexture &t = out.textures[static_cast<size_t>(i)];
        std::memset(&t, 0, sizeof(t));
        t.kind = d.tex_kind[i];
        if (t.kind < RTK_TEX_ALBEDO || t.kind > RTK_TEX_BITMAP) { err = "texture type unknown"; return RTK_ERR_INVALID; }
        if (t.kind == RTK_TEX_BITMAP) {                                  // bitmap_texture, texture/bitmap.hpp:40-44
            if (!d.tex_bitmap || !d.tex_pixels) { err = "bitmap texture without tex_bitmap / tex_pixels"; return RTK_ERR_INVALID; }
            const int32_t *b = d.tex_bitmap + i * 3;
            if (b[0] < 0 || b[1] <= 0 || b[2] <= 0 || static_cast<int64_t>(b[1]) * b[2] > (int64_t{1} << 28)) { err = "bad bitmap texture size"; return RTK_ERR_INVALID; }
            const size_t n = static_cast<size_t>(b[1]) * static_cast<size_t>(b[2]) * 3;
            t.bmp[0] = b[1]; t.bmp[1] = b[2]; t.bmp[2] = static_cast<int32_t>(out.tex_pixels.size());
            out.tex_pixels.insert(out.tex_pixels.end(), d.tex_pixels + b[0], d.tex_pixels + b[0] + n);
            continue;
        }
        if (d.tex_color_a) std::memcpy(t.a, d.tex_color_a + i * 3, sizeof(float) * 3);
        if (d.tex_color_b) std::memcpy(t.b, d.tex_color_b + i * 3, sizeof(float) * 3);
        t.param = d.tex_param ? d.tex_param[i] : 0.0f;
    }
    out.lights.resize(static_cast<size_t>(d.n_lights));
    for (int32_t i = 0; i < d.n_lights; ++i) {
        DevLight &l = out.lights[static_cast<size_t>(i)];
        std::memcpy(l.pos, d.light_pos + i * 3, sizeof(float) * 3);
        l.intensity = d.light_intensity[i];
    }
    std::memcpy(out.cam_pos, d.cam_pos, sizeof(out.cam_pos));
    std::memcpy(out.cam_mat, d.cam_mat, sizeof(out.cam_mat));
    std::memcpy(out.background, d.background, sizeof(out.background));
    out.width = d.width; out.height = d.height;
    out.bucket_size = d.bucket_size > 0 ? d.bucket_size : 64;          // loader.hpp:48
    return RTK_OK;
}

}  // namespace rtk
