// Internal host/device data model of the rtk engine (not part of the C-ABI).
// Reference citations are relative to /root/reference/include/raytracer/.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtk.h"

namespace rtk {

// ---------------------------------------------------------------- device layout (HBM)
// Everything the kernels read is uploaded once per accel and stays resident.

// 32-byte kd-tree node, stored in TRAVERSAL order: the reference pops child1 before child0
// (kd_tree_simd.hpp:207-214) and has no near/far ordering, so the visiting order is the same for
// every ray.  Nodes are laid out in that order (node, child1 subtree, child0 subtree), which turns
// the LIFO stack into two indices: "descend" = n+1, "skip subtree" = skip.
struct DevNode {
    float lo[3];
    float hi[3];
    uint32_t a;   // inner: index of the next node once this subtree is done (skip); leaf: first leaf-ref
    uint32_t b;   // inner: 0xFFFFFFFF; leaf: number of leaf-refs
};
static_assert(sizeof(DevNode) == 32, "DevNode must be 32 bytes");
constexpr uint32_t DEV_INNER = 0xFFFFFFFFu;

// 36-byte leaf reference: the packet payload v0,e1,e2 of one triangle (kd_tree_simd.hpp:15-23),
// stored once per (leaf, triangle) in leaf order so a leaf is one contiguous run.
struct DevTri {
    float v0[3];
    float e1[3];
    float e2[3];
};
static_assert(sizeof(DevTri) == 36, "DevTri must be 36 bytes");

// 64-byte per-triangle record used only when a hit is reconstructed (kd_tree_simd.hpp:234-263).
struct DevShade {
    float n0[3];      // mesh.vertex_normals[vertex_indices[0]]
    float n1[3];
    float n2[3];
    float fn[3];      // triangle.normal
    uint32_t mesh;
    uint32_t material;
    uint32_t pad[2];
};
static_assert(sizeof(DevShade) == 64, "DevShade must be 64 bytes");

struct DevMaterial {  // scene/material/*.hpp flattened
    int32_t kind;
    int32_t smooth;
    float albedo[3];
    float ior;
    int32_t texture;      // RTK_MAT_TEXTURE: index into the texture table
    uint32_t pad;
};

struct DevTexture {   // scene/texture/{albedo,edge,checker,bitmap}.hpp flattened
    int32_t kind;
    union {
        float a[3];       // albedo / edge_color / color_a
        int32_t bmp[3];   // RTK_TEX_BITMAP: width, height, byte offset of the first texel in the texel buffer
    };
    float b[3];           // - / inner_color / color_b
    float param;          // - / edge_width / square_size
};
static_assert(sizeof(DevTexture) == 32, "DevTexture must be 32 bytes");

struct DevTriUv {     // triangle::uvs (scene/primitive/triangle.hpp:18): uv of vertex_indices[0], [1], [2]
    float uv[6];
};
static_assert(sizeof(DevMaterial) == 32, "DevMaterial must be 32 bytes");

struct DevLight {     // scene/light.hpp
    float pos[3];
    float intensity;
};

// ---------------------------------------------------------------- host model

struct Vec3 { float x, y, z; };
struct Box { Vec3 mn, mx; };

struct HostTriangle {   // scene/primitive/triangle.hpp:11-30
    Vec3 v0, v1, v2, e1, e2, normal;
    uint32_t vi[3];
    uint32_t mesh;
    Box box;
};

struct HostMesh {       // scene/object/mesh.hpp:15-44
    int32_t material = 0;
    std::vector<Vec3> vertices;
    std::vector<Vec3> vertex_normals;
    std::vector<float> uvs;          // [nverts*2] or empty (then every triangle's uvs are zero, loader.hpp:199-207)
    std::vector<uint32_t> indices;   // [ntris*3]
    Box box;
};

struct HostNode {       // kd_tree_simd.hpp:75-84, reference (creation) order
    Box box;
    int32_t child0 = -1, child1 = -1;
    int32_t leaf_start = -1, leaf_count = 0;   // into leaf_refs (unpadded)
    int32_t depth = 0;
};

}  // namespace rtk

struct rtk_scene {
    std::vector<rtk::HostMesh> meshes;
    std::vector<rtk::DevMaterial> materials;
    std::vector<rtk::DevTexture> textures;
    std::vector<uint8_t> tex_pixels;           // RGB bytes of all bitmap textures (bitmap.hpp:11-37), DevTexture::bmp points into it
    std::vector<rtk::DevLight> lights;
    float cam_pos[3];
    float cam_mat[9];
    float background[3];
    int32_t width = 0, height = 0, bucket_size = 64;
    int32_t n_vertices = 0, n_triangles = 0;
};

namespace rtk {

struct HostTree {
    std::vector<HostTriangle> triangles;       // concatenated in mesh order (kd_tree_simd.hpp:103-111)
    std::vector<HostNode> nodes;               // reference order
    std::vector<int32_t> leaf_refs;            // unpadded leaf triangle lists, reference leaf order
    // device-order flattening
    std::vector<DevNode> dev_nodes;
    std::vector<DevNode> dev_leaves;           // the leaves of dev_nodes alone, same order (leaf-list traversal, trace.hip.hpp)
    std::vector<DevNode> dev_leaves_fast;      // RTK_TRAVERSAL_FAST: [8][n_leaves] the same leaves front to back for each direction octant
                                               // (bit a of the octant set = rays travel towards smaller coordinates on axis a)
    std::vector<DevTri> dev_tris;
    std::vector<uint32_t> dev_tri_ids;         // leaf-ref -> global triangle index
    std::vector<DevShade> dev_shade;           // per global triangle
    std::vector<DevTriUv> dev_tri_uv;          // per global triangle (only filled when the scene has textures)
    int32_t depth = 0;
};

// scene.cpp
int scene_from_desc(const rtk_scene_desc &d, rtk_scene &out, std::string &err);
// crtscene.cpp
int scene_from_crtscene(const char *path, rtk_scene &out, std::string &err);
// jpeg.cpp: stbi_load's result for a baseline JPEG (bitmap.hpp:15)
int decode_jpeg(const uint8_t *data, size_t size, int &width, int &height, int &channels, std::vector<uint8_t> &pixels, std::string &err);
int load_bitmap_file(const std::string &path, int &width, int &height, int &channels, std::vector<uint8_t> &pixels, std::string &err);
// kdtree.cpp
int build_tree(const rtk_scene &scene, int max_depth, int max_leaf, HostTree &out, std::string &err);
void build_fast_leaf_orders(HostTree &tree);          // fills dev_leaves_fast (RTK_TRAVERSAL_FAST)
// ppm.cpp
std::string format_ppm(const float *rgb, int width, int height);
std::string format_ppm_rgb8(const uint8_t *rgb8, int width, int height);

void set_error(const std::string &msg);

}  // namespace rtk
