// Host kd-tree build (kd_tree_simd.hpp:100-185) and flattening into the device layout.
// Serial and cheap (thousands of triangles, depth <= max_depth); stays on the host as in the reference.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "rtk_internal.hpp"

namespace rtk {

void box_reset(Box &b);                       // scene.cpp
void box_grow(Box &b, const Vec3 &p);

namespace {

inline float comp(const Vec3 &v, unsigned axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
inline void set_comp(Vec3 &v, unsigned axis, float val) { if (axis == 0) v.x = val; else if (axis == 1) v.y = val; else v.z = val; }

inline Vec3 sub(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 crossp(const Vec3 &a, const Vec3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 unitv(const Vec3 &v) {
    const float inv_length = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return {v.x * inv_length, v.y * inv_length, v.z * inv_length};
}

// core/math/aabb3.hpp:68-72 (inclusive on both sides => triangles on the split plane go to both children)
inline bool overlaps(const Box &a, const Box &o) {
    return (o.mn.x <= a.mx.x && a.mn.x <= o.mx.x) && (o.mn.y <= a.mx.y && a.mn.y <= o.mx.y) &&
           (o.mn.z <= a.mx.z && a.mn.z <= o.mx.z);
}

// core/math/aabb3.hpp:43-60: midpoint split; an axis of zero extent hands over to the next axis
bool split_box(const Box &b, unsigned axis, Box &lo, Box &hi) {
    for (int tries = 0; tries < 3 && comp(b.mn, axis) == comp(b.mx, axis); ++tries) axis = (axis + 1u) % 3u;
    if (comp(b.mn, axis) == comp(b.mx, axis)) return false;            // a point box: the reference would recurse forever
    const float mid = comp(b.mn, axis) + ((comp(b.mx, axis) - comp(b.mn, axis)) / 2.0f);
    lo = b; hi = b;
    set_comp(lo.mx, axis, mid);
    set_comp(hi.mn, axis, mid);
    return true;
}

struct Builder {
    HostTree &t;
    int max_depth, max_leaf;

    void make_leaf(int32_t node, const std::vector<int32_t> &ids) {     // kd_tree_simd.hpp:117-144 without the W padding
        t.nodes[size_t(node)].leaf_start = int32_t(t.leaf_refs.size());
        t.nodes[size_t(node)].leaf_count = int32_t(ids.size());
        t.leaf_refs.insert(t.leaf_refs.end(), ids.begin(), ids.end());
    }

    void build(int32_t node, int depth, const std::vector<int32_t> &ids) {   // kd_tree_simd.hpp:146-185
        t.nodes[size_t(node)].depth = depth;
        if (depth > t.depth) t.depth = depth;
        Box b0, b1;
        if (depth == max_depth || int64_t(ids.size()) <= max_leaf ||
            !split_box(t.nodes[size_t(node)].box, unsigned(depth % 3), b0, b1)) {
            make_leaf(node, ids);
            return;
        }
        std::vector<int32_t> c0, c1;
        c0.reserve(ids.size());
        c1.reserve(ids.size());
        for (int32_t id : ids) {
            const Box &tb = t.triangles[size_t(id)].box;
            if (overlaps(b0, tb)) c0.push_back(id);
            if (overlaps(b1, tb)) c1.push_back(id);
        }
        if (!c0.empty()) {
            const int32_t c = int32_t(t.nodes.size());
            HostNode n; n.box = b0;
            t.nodes.push_back(n);
            t.nodes[size_t(node)].child0 = c;
            build(c, depth + 1, c0);
        }
        if (!c1.empty()) {
            const int32_t c = int32_t(t.nodes.size());
            HostNode n; n.box = b1;
            t.nodes.push_back(n);
            t.nodes[size_t(node)].child1 = c;
            build(c, depth + 1, c1);
        }
    }
};

// Emits `node` and its subtree in traversal order (child1 before child0, kd_tree_simd.hpp:207-214).
void flatten(HostTree &t, int32_t node) {
    const HostNode &hn = t.nodes[size_t(node)];
    const size_t at = t.dev_nodes.size();
    DevNode dn;
    dn.lo[0] = hn.box.mn.x; dn.lo[1] = hn.box.mn.y; dn.lo[2] = hn.box.mn.z;
    dn.hi[0] = hn.box.mx.x; dn.hi[1] = hn.box.mx.y; dn.hi[2] = hn.box.mx.z;
    dn.a = 0; dn.b = DEV_INNER;
    t.dev_nodes.push_back(dn);
    if (hn.leaf_start >= 0) {
        t.dev_nodes[at].a = uint32_t(t.dev_tris.size());
        t.dev_nodes[at].b = uint32_t(hn.leaf_count);
        for (int32_t k = 0; k < hn.leaf_count; ++k) {
            const int32_t id = t.leaf_refs[size_t(hn.leaf_start + k)];
            const HostTriangle &tr = t.triangles[size_t(id)];
            DevTri d;
            d.v0[0] = tr.v0.x; d.v0[1] = tr.v0.y; d.v0[2] = tr.v0.z;
            d.e1[0] = tr.e1.x; d.e1[1] = tr.e1.y; d.e1[2] = tr.e1.z;
            d.e2[0] = tr.e2.x; d.e2[1] = tr.e2.y; d.e2[2] = tr.e2.z;
            t.dev_tris.push_back(d);
            t.dev_tri_ids.push_back(uint32_t(id));
        }
        t.dev_leaves.push_back(t.dev_nodes[at]);
        return;
    }
    if (hn.child1 >= 0) flatten(t, hn.child1);
    if (hn.child0 >= 0) flatten(t, hn.child0);
    t.dev_nodes[at].a = uint32_t(t.dev_nodes.size());                  // skip = first node after this subtree
}

// RTK_TRAVERSAL_FAST: the leaves of the subtree of `node`, front to back for rays of direction octant `oct`.  The children
// of a node are the halves of its box on either side of one axis-aligned plane (aabb3::split, aabb3.hpp:43-60: child0 below,
// child1 above), so which half a ray passes first depends on the sign of its direction on that axis alone.
void leaves_front_to_back(const HostTree &t, int32_t node, unsigned oct, std::vector<DevNode> &out, const std::vector<uint32_t> &leaf_slot) {
    const HostNode &hn = t.nodes[size_t(node)];
    if (hn.leaf_start >= 0) { out.push_back(t.dev_leaves[leaf_slot[size_t(node)]]); return; }
    int axis = 0;
    if (hn.child0 >= 0) {
        const Box &c = t.nodes[size_t(hn.child0)].box;
        axis = c.mx.x != hn.box.mx.x ? 0 : (c.mx.y != hn.box.mx.y ? 1 : 2);
    } else if (hn.child1 >= 0) {
        const Box &c = t.nodes[size_t(hn.child1)].box;
        axis = c.mn.x != hn.box.mn.x ? 0 : (c.mn.y != hn.box.mn.y ? 1 : 2);
    }
    const bool downwards = ((oct >> axis) & 1u) != 0u;
    const int32_t first = downwards ? hn.child1 : hn.child0, second = downwards ? hn.child0 : hn.child1;
    if (first >= 0) leaves_front_to_back(t, first, oct, out, leaf_slot);
    if (second >= 0) leaves_front_to_back(t, second, oct, out, leaf_slot);
}

}  // namespace

void build_fast_leaf_orders(HostTree &t) {
    // where each leaf node went in dev_leaves: replay flatten's order (node, child1 subtree, child0 subtree)
    std::vector<uint32_t> leaf_slot(t.nodes.size(), 0u);
    {
        uint32_t next = 0;
        std::vector<int32_t> stack{0};
        while (!stack.empty()) {
            const int32_t n = stack.back();
            stack.pop_back();
            const HostNode &hn = t.nodes[size_t(n)];
            if (hn.leaf_start >= 0) { leaf_slot[size_t(n)] = next++; continue; }
            if (hn.child0 >= 0) stack.push_back(hn.child0);          // popped second
            if (hn.child1 >= 0) stack.push_back(hn.child1);          // popped first
        }
    }
    t.dev_leaves_fast.clear();
    t.dev_leaves_fast.reserve(t.dev_leaves.size() * 8);
    for (unsigned oct = 0; oct < 8u; ++oct) leaves_front_to_back(t, 0, oct, t.dev_leaves_fast, leaf_slot);
}

int build_tree(const rtk_scene &scene, int max_depth, int max_leaf, HostTree &out, std::string &err) {
    if (max_depth < 0 || max_depth > 24) { err = "max_depth must be in [0, 24]"; return RTK_ERR_INVALID; }
    if (max_leaf < 1) { err = "max_leaf_size must be >= 1"; return RTK_ERR_INVALID; }
    out = HostTree{};
    Box root;
    box_reset(root);
    std::vector<int32_t> all;
    for (size_t mi = 0; mi < scene.meshes.size(); ++mi) {               // kd_tree_simd.hpp:101-111
        const HostMesh &m = scene.meshes[mi];
        // aabb3::unite (aabb3.hpp:33-40)
        root.mn.x = m.box.mn.x < root.mn.x ? m.box.mn.x : root.mn.x;
        root.mn.y = m.box.mn.y < root.mn.y ? m.box.mn.y : root.mn.y;
        root.mn.z = m.box.mn.z < root.mn.z ? m.box.mn.z : root.mn.z;
        root.mx.x = root.mx.x < m.box.mx.x ? m.box.mx.x : root.mx.x;
        root.mx.y = root.mx.y < m.box.mx.y ? m.box.mx.y : root.mx.y;
        root.mx.z = root.mx.z < m.box.mx.z ? m.box.mx.z : root.mx.z;
        for (size_t ti = 0; ti < m.indices.size() / 3; ++ti) {          // triangle ctor, triangle.hpp:20-30
            HostTriangle tr;
            for (int k = 0; k < 3; ++k) tr.vi[k] = m.indices[ti * 3 + size_t(k)];
            tr.v0 = m.vertices[tr.vi[0]]; tr.v1 = m.vertices[tr.vi[1]]; tr.v2 = m.vertices[tr.vi[2]];
            tr.normal = unitv(crossp(sub(tr.v1, tr.v0), sub(tr.v2, tr.v0)));
            tr.e1 = sub(tr.v1, tr.v0);
            tr.e2 = sub(tr.v2, tr.v0);
            tr.mesh = uint32_t(mi);
            box_reset(tr.box);
            box_grow(tr.box, tr.v0); box_grow(tr.box, tr.v1); box_grow(tr.box, tr.v2);
            all.push_back(int32_t(out.triangles.size()));
            out.triangles.push_back(tr);
        }
    }
    HostNode rn; rn.box = root;
    out.nodes.push_back(rn);
    Builder b{out, max_depth, max_leaf};
    b.build(0, 0, all);

    out.dev_nodes.reserve(out.nodes.size());
    out.dev_tris.reserve(out.leaf_refs.size());
    out.dev_tri_ids.reserve(out.leaf_refs.size());
    flatten(out, 0);

    out.dev_shade.resize(out.triangles.size());
    for (size_t i = 0; i < out.triangles.size(); ++i) {
        const HostTriangle &tr = out.triangles[i];
        const HostMesh &m = scene.meshes[tr.mesh];
        DevShade &s = out.dev_shade[i];
        std::memset(&s, 0, sizeof(s));
        const Vec3 &a = m.vertex_normals[tr.vi[0]], &bb = m.vertex_normals[tr.vi[1]], &c = m.vertex_normals[tr.vi[2]];
        s.n0[0] = a.x; s.n0[1] = a.y; s.n0[2] = a.z;
        s.n1[0] = bb.x; s.n1[1] = bb.y; s.n1[2] = bb.z;
        s.n2[0] = c.x; s.n2[1] = c.y; s.n2[2] = c.z;
        s.fn[0] = tr.normal.x; s.fn[1] = tr.normal.y; s.fn[2] = tr.normal.z;
        s.mesh = tr.mesh;
        s.material = uint32_t(m.material);
    }
    if (!scene.textures.empty()) {                                       // triangle::uvs, loader.hpp:199-207
        out.dev_tri_uv.resize(out.triangles.size());
        for (size_t i = 0; i < out.triangles.size(); ++i) {
            const HostTriangle &tr = out.triangles[i];
            const HostMesh &m = scene.meshes[tr.mesh];
            DevTriUv &u = out.dev_tri_uv[i];
            for (int k = 0; k < 3; ++k) {
                u.uv[k * 2] = m.uvs.empty() ? 0.0f : m.uvs[size_t(tr.vi[k]) * 2];
                u.uv[k * 2 + 1] = m.uvs.empty() ? 0.0f : m.uvs[size_t(tr.vi[k]) * 2 + 1];
            }
        }
    }
    if (out.dev_nodes.size() != out.nodes.size() || out.dev_tris.size() != out.leaf_refs.size()) {
        err = "internal: flattening lost nodes"; return RTK_ERR_INVALID;
    }
    return RTK_OK;
}

}  // namespace rtk
