/*
 * rt_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see rt_oracle.h for the rules and the parity-pin status).
 *
 * Build with -ffp-contract=off: every expression below keeps the reference's
 * operation order so that IEEE add/mul/div/sqrt give bit-identical results to a
 * reference build with the same flag.
 *
 * File:line citations are relative to /root/reference/include/raytracer/.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#ifndef ORA_MAXW
#define ORA_MAXW 16
#endif

/* ------------------------------------------------------------------ math (core/math) */

typedef struct { float x, y, z; } v3;

static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vec3.hpp:77-79 */
static inline v3 sub3(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }   /* vec3.hpp:81-83 */
static inline v3 scl3(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }      /* vec3.hpp:94-97 */
static inline v3 neg3(v3 a) { return mk(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); } /* vec3.hpp:119-122 */
static inline v3 cross3(v3 a, v3 b) {                                               /* vec3.hpp:124-131 */
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float len3(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); } /* vec3.hpp:85-91 */
static inline v3 norm3(v3 a) {                                                      /* vec3.hpp:104-108 */
    const float inv = 1.0f / len3(a);
    return mk(a.x * inv, a.y * inv, a.z * inv);
}
static inline float get3(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

typedef struct { v3 origin, direction, inv_direction; } ray3;
static inline ray3 mkray(v3 o, v3 d) {                                              /* ray3.hpp:11-14 */
    ray3 r; r.origin = o; r.direction = d;
    r.inv_direction = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                       /* vec3.hpp:99-102 */
    return r;
}

typedef struct { v3 min, max; } aabb3;
static inline aabb3 aabb_empty(void) {                                              /* aabb3.hpp:20-22 */
    aabb3 b; b.min = mk(FLT_MAX, FLT_MAX, FLT_MAX); b.max = mk(-FLT_MAX, -FLT_MAX, -FLT_MAX); return b;
}
static inline float fminstd(float a, float b) { return (b < a) ? b : a; }           /* std::min */
static inline float fmaxstd(float a, float b) { return (a < b) ? b : a; }           /* std::max */
static inline void aabb_expand(aabb3 *b, v3 p) {                                    /* aabb3.hpp:24-31 */
    b->min.x = fminstd(b->min.x, p.x); b->min.y = fminstd(b->min.y, p.y); b->min.z = fminstd(b->min.z, p.z);
    b->max.x = fmaxstd(b->max.x, p.x); b->max.y = fmaxstd(b->max.y, p.y); b->max.z = fmaxstd(b->max.z, p.z);
}
static inline void aabb_unite(aabb3 *b, const aabb3 *o) {                           /* aabb3.hpp:33-40 */
    b->min.x = fminstd(b->min.x, o->min.x); b->min.y = fminstd(b->min.y, o->min.y); b->min.z = fminstd(b->min.z, o->min.z);
    b->max.x = fmaxstd(b->max.x, o->max.x); b->max.y = fmaxstd(b->max.y, o->max.y); b->max.z = fmaxstd(b->max.z, o->max.z);
}
static inline int aabb_overlap(const aabb3 *a, const aabb3 *o) {                    /* aabb3.hpp:68-72 */
    return (o->min.x <= a->max.x && a->min.x <= o->max.x) &&
           (o->min.y <= a->max.y && a->min.y <= o->max.y) &&
           (o->min.z <= a->max.z && a->min.z <= o->max.z);
}
static void aabb_split(const aabb3 *b, unsigned axis, aabb3 *b0, aabb3 *b1) {       /* aabb3.hpp:43-60 */
    float mn = get3(b->min, (int)axis), mx = get3(b->max, (int)axis);
    int guard = 0;
    while (mn == mx && guard < 3) {  /* the reference recurses; all-degenerate boxes never reach here */
        axis = (axis + 1u) % 3u; mn = get3(b->min, (int)axis); mx = get3(b->max, (int)axis); ++guard;
    }
    const float mid = mn + ((mx - mn) / 2.0f);
    *b0 = *b; *b1 = *b;
    if (axis == 0) { b0->max.x = mid; b1->min.x = mid; }
    else if (axis == 1) { b0->max.y = mid; b1->min.y = mid; }
    else { b0->max.z = mid; b1->min.z = mid; }
}
/* aabb3.hpp:74-90 — slab test; NaN handling follows std::minmax/max/min exactly. */
#define SLAB_AXIS(lo_, hi_, o_, inv_)                                        \
    do {                                                                     \
        const float a_ = ((lo_) - (o_)) * (inv_), c_ = ((hi_) - (o_)) * (inv_); \
        const float t1_ = (c_ < a_) ? c_ : a_;   /* std::minmax(a,c).first  */ \
        const float t2_ = (c_ < a_) ? a_ : c_;   /* std::minmax(a,c).second */ \
        t_min = (t_min < t1_) ? t1_ : t_min;     /* std::max(t_min,t1) */    \
        t_max = (t2_ < t_max) ? t2_ : t_max;     /* std::min(t_max,t2) */    \
        if (t_max < t_min) return 0;                                         \
    } while (0)
static inline int aabb_ray(const aabb3 *b, const ray3 *r, float *out_tmin) {
    float t_min = 0.0f, t_max = FLT_MAX;
    SLAB_AXIS(b->min.x, b->max.x, r->origin.x, r->inv_direction.x);
    SLAB_AXIS(b->min.y, b->max.y, r->origin.y, r->inv_direction.y);
    SLAB_AXIS(b->min.z, b->max.z, r->origin.z, r->inv_direction.z);
    *out_tmin = t_min;
    return 1;
}

/* ------------------------------------------------------------------ scene (scene/) */

typedef struct {                                /* scene/primitive/triangle.hpp:11-30 */
    v3 v0, v1, v2, e1, e2, normal;
    uint32_t vi[3];
    uint32_t mesh_idx;
    aabb3 box;
    float uvs[6];                               /* vec3<vec2<F>> uvs, triangle.hpp:18 */
} triangle;

typedef struct {                                /* scene/object/mesh.hpp:15-44 */
    int32_t material_idx;
    int32_t nverts, ntris;
    v3 *vertices;
    v3 *vertex_normals;
    triangle *triangles;
    aabb3 box;
} mesh_object;

typedef struct { int32_t kind; float albedo[3]; float ior; int32_t smooth; int32_t texture; } material;
typedef struct { int32_t kind; float a[3], b[3]; float param;             /* scene/texture/{albedo,edge,checker}.hpp */
                 int32_t bw, bh; uint8_t *pixels; } texture;              /* bitmap.hpp:40-44: image<F> as its RGB bytes */
typedef struct { v3 position; float intensity; } light;

struct ora_scene {
    int32_t n_meshes; mesh_object *meshes;
    int32_t n_materials; material *materials;
    int32_t n_textures; texture *textures;
    int32_t n_lights; light *lights;
    v3 cam_pos; float cam_mat[9]; float background[3];
    int32_t width, height, bucket_size;
};

static triangle make_triangle(v3 v0, v3 v1, v3 v2, const uint32_t vi[3], uint32_t mesh_idx) { /* triangle.hpp:20-30 */
    triangle t;
    t.v0 = v0; t.v1 = v1; t.v2 = v2;
    t.vi[0] = vi[0]; t.vi[1] = vi[1]; t.vi[2] = vi[2];
    t.mesh_idx = mesh_idx;
    t.normal = norm3(cross3(sub3(v1, v0), sub3(v2, v0)));
    t.e1 = sub3(v1, v0);
    t.e2 = sub3(v2, v0);
    t.box = aabb_empty();
    aabb_expand(&t.box, v0); aabb_expand(&t.box, v1); aabb_expand(&t.box, v2);
    return t;
}

ora_scene *ora_scene_create(const ora_scene_desc *d) {
    ora_scene *s = (ora_scene *)calloc(1, sizeof(*s));
    s->n_meshes = d->n_meshes;
    s->meshes = (mesh_object *)calloc((size_t)d->n_meshes, sizeof(mesh_object));
    size_t voff = 0, toff = 0, uvoff = 0;
    for (int m = 0; m < d->n_meshes; ++m) {                          /* loader.hpp:149-233 + mesh.hpp:23-44 */
        mesh_object *mo = &s->meshes[m];
        mo->material_idx = d->mesh_material[m];
        mo->nverts = d->mesh_nverts[m];
        mo->ntris = d->mesh_ntris[m];
        mo->vertices = (v3 *)malloc(sizeof(v3) * (size_t)(mo->nverts ? mo->nverts : 1));
        mo->vertex_normals = (v3 *)calloc((size_t)(mo->nverts ? mo->nverts : 1), sizeof(v3));
        mo->triangles = (triangle *)malloc(sizeof(triangle) * (size_t)(mo->ntris ? mo->ntris : 1));
        for (int i = 0; i < mo->nverts; ++i)
            mo->vertices[i] = mk(d->vertices[(voff + (size_t)i) * 3 + 0], d->vertices[(voff + (size_t)i) * 3 + 1],
                                 d->vertices[(voff + (size_t)i) * 3 + 2]);
        for (int i = 0; i < mo->ntris; ++i) {
            const uint32_t *ix = &d->indices[(toff + (size_t)i) * 3];
            mo->triangles[i] = make_triangle(mo->vertices[ix[0]], mo->vertices[ix[1]], mo->vertices[ix[2]], ix, (uint32_t)m);
            memset(mo->triangles[i].uvs, 0, sizeof(mo->triangles[i].uvs));
            if (d->mesh_has_uvs && d->mesh_has_uvs[m])                       /* loader.hpp:199-207 */
                for (int k = 0; k < 3; ++k) {
                    mo->triangles[i].uvs[k * 2] = d->uvs[(uvoff + ix[k]) * 2];
                    mo->triangles[i].uvs[k * 2 + 1] = d->uvs[(uvoff + ix[k]) * 2 + 1];
                }
        }
        if (d->mesh_has_uvs && d->mesh_has_uvs[m]) uvoff += (size_t)mo->nverts;
        mo->box = aabb_empty();
        for (int i = 0; i < mo->ntris; ++i) {                         /* mesh.hpp:27-38 */
            const triangle *t = &mo->triangles[i];
            aabb_expand(&mo->box, t->v0); aabb_expand(&mo->box, t->v1); aabb_expand(&mo->box, t->v2);
            const v3 tn = norm3(cross3(sub3(t->v1, t->v0), sub3(t->v2, t->v0)));
            mo->vertex_normals[t->vi[0]] = add3(mo->vertex_normals[t->vi[0]], tn);
            mo->vertex_normals[t->vi[1]] = add3(mo->vertex_normals[t->vi[1]], tn);
            mo->vertex_normals[t->vi[2]] = add3(mo->vertex_normals[t->vi[2]], tn);
        }
        for (int i = 0; i < mo->nverts; ++i) mo->vertex_normals[i] = norm3(mo->vertex_normals[i]); /* mesh.hpp:41-43 */
        voff += (size_t)mo->nverts; toff += (size_t)mo->ntris;
    }
    s->n_materials = d->n_materials;
    s->materials = (material *)calloc((size_t)(d->n_materials ? d->n_materials : 1), sizeof(material));
    for (int i = 0; i < d->n_materials; ++i) {
        s->materials[i].kind = d->mat_kind[i];
        memcpy(s->materials[i].albedo, &d->mat_albedo[i * 3], sizeof(float) * 3);
        s->materials[i].ior = d->mat_ior[i];
        s->materials[i].smooth = d->mat_smooth[i];
        s->materials[i].texture = d->mat_texture ? d->mat_texture[i] : -1;
    }
    s->n_textures = d->n_textures;
    s->textures = (texture *)calloc((size_t)(d->n_textures ? d->n_textures : 1), sizeof(texture));
    for (int i = 0; i < d->n_textures; ++i) {
        s->textures[i].kind = d->tex_kind[i];
        memcpy(s->textures[i].a, &d->tex_color_a[i * 3], sizeof(float) * 3);
        memcpy(s->textures[i].b, &d->tex_color_b[i * 3], sizeof(float) * 3);
        s->textures[i].param = d->tex_param[i];
        if (d->tex_kind[i] == ORA_TEX_BITMAP) {                             /* load_bitmap, bitmap.hpp:11-37 (decoded by the caller) */
            const int32_t *b = &d->tex_bitmap[i * 3];
            const size_t n = (size_t)b[1] * (size_t)b[2] * 3;
            s->textures[i].bw = b[1]; s->textures[i].bh = b[2];
            s->textures[i].pixels = (uint8_t *)malloc(n ? n : 1);
            memcpy(s->textures[i].pixels, d->tex_pixels + b[0], n);
        }
    }
    s->n_lights = d->n_lights;
    s->lights = (light *)calloc((size_t)(d->n_lights ? d->n_lights : 1), sizeof(light));
    for (int i = 0; i < d->n_lights; ++i) {
        s->lights[i].position = mk(d->light_pos[i * 3], d->light_pos[i * 3 + 1], d->light_pos[i * 3 + 2]);
        s->lights[i].intensity = d->light_intensity[i];
    }
    s->cam_pos = mk(d->cam_pos[0], d->cam_pos[1], d->cam_pos[2]);
    memcpy(s->cam_mat, d->cam_mat, sizeof(s->cam_mat));
    memcpy(s->background, d->background, sizeof(s->background));
    s->width = d->width; s->height = d->height; s->bucket_size = d->bucket_size;
    return s;
}

void ora_scene_destroy(ora_scene *s) {
    if (!s) return;
    for (int m = 0; m < s->n_meshes; ++m) {
        free(s->meshes[m].vertices); free(s->meshes[m].vertex_normals); free(s->meshes[m].triangles);
    }
    for (int i = 0; i < s->n_textures; ++i) free(s->textures[i].pixels);
    free(s->meshes); free(s->materials); free(s->textures); free(s->lights); free(s);
}

void ora_scene_vertex_normals(const ora_scene *s, int mesh, float *out) {
    const mesh_object *mo = &s->meshes[mesh];
    for (int i = 0; i < mo->nverts; ++i) {
        out[i * 3 + 0] = mo->vertex_normals[i].x; out[i * 3 + 1] = mo->vertex_normals[i].y; out[i * 3 + 2] = mo->vertex_normals[i].z;
    }
}

/* ------------------------------------------------------------------ accel (render/accel) */

#define EMPTY (-1)

typedef struct {                                 /* kd_tree_simd.hpp:75-84 / kd_tree.hpp:14-23 */
    aabb3 box;
    int32_t child0, child1;
    int32_t start_idx;    /* packets (kd_simd) or leaf_indices (kd_scalar); EMPTY for inner */
    int32_t count;        /* pack_count or count */
    int32_t ref_start;    /* start into the unpadded leaf-ref array */
    int32_t ref_count;
} kd_node;

typedef struct {                                 /* kd_tree_simd.hpp:15-23, SoA of W lanes */
    float v0x[ORA_MAXW], v0y[ORA_MAXW], v0z[ORA_MAXW];
    float e1x[ORA_MAXW], e1y[ORA_MAXW], e1z[ORA_MAXW];
    float e2x[ORA_MAXW], e2y[ORA_MAXW], e2z[ORA_MAXW];
    uint32_t tri[ORA_MAXW];
} tri_packet __attribute__((aligned(64)));

struct ora_accel {
    const ora_scene *scene;
    int kind; float eps; int max_depth, max_leaf, W;
    triangle *triangles; int64_t n_triangles;
    kd_node *tree; int64_t n_nodes, cap_nodes;
    tri_packet *packs; int64_t n_packs, cap_packs;
    int32_t *leaf_refs; int64_t n_refs, cap_refs;
};

static int32_t push_node(ora_accel *a, const aabb3 *box) {
    if (a->n_nodes == a->cap_nodes) {
        a->cap_nodes = a->cap_nodes ? a->cap_nodes * 2 : 256;
        a->tree = (kd_node *)realloc(a->tree, sizeof(kd_node) * (size_t)a->cap_nodes);
    }
    kd_node *n = &a->tree[a->n_nodes];
    n->box = *box; n->child0 = EMPTY; n->child1 = EMPTY; n->start_idx = EMPTY; n->count = 0; n->ref_start = EMPTY; n->ref_count = 0;
    return (int32_t)a->n_nodes++;
}

static void push_refs(ora_accel *a, const int32_t *idx, int64_t n) {
    while (a->n_refs + n > a->cap_refs) {
        a->cap_refs = a->cap_refs ? a->cap_refs * 2 : 1024;
        a->leaf_refs = (int32_t *)realloc(a->leaf_refs, sizeof(int32_t) * (size_t)a->cap_refs);
    }
    memcpy(a->leaf_refs + a->n_refs, idx, sizeof(int32_t) * (size_t)n);
    a->n_refs += n;
}

/* kd_tree_simd.hpp:117-144 (packets, tail padded with the last triangle) /
 * kd_tree.hpp:41-46 (plain index list). */
static void build_leaf(ora_accel *a, int32_t node, const int32_t *idx, int64_t n) {
    a->tree[node].ref_start = (int32_t)a->n_refs;
    a->tree[node].ref_count = (int32_t)n;
    push_refs(a, idx, n);
    if (a->kind == ORA_ACCEL_KD_SCALAR) {
        a->tree[node].start_idx = a->tree[node].ref_start;
        a->tree[node].count = (int32_t)n;
        return;
    }
    const int W = a->W;
    const int64_t first = a->n_packs;
    for (int64_t i = 0; i < n; i += W) {
        if (a->n_packs == a->cap_packs) {
            a->cap_packs = a->cap_packs ? a->cap_packs * 2 : 256;
            tri_packet *np = NULL;
            if (posix_memalign((void **)&np, 64, sizeof(tri_packet) * (size_t)a->cap_packs)) abort();
            if (a->packs) { memcpy(np, a->packs, sizeof(tri_packet) * (size_t)a->n_packs); free(a->packs); }
            a->packs = np;
        }
        tri_packet *p = &a->packs[a->n_packs++];
        memset(p, 0, sizeof(*p));
        for (int lane = 0; lane < W; ++lane) {
            const int64_t j = (i + lane < n - 1) ? i + lane : n - 1;      /* std::min(i+lane, size-1) :123 */
            const triangle *t = &a->triangles[idx[j]];
            p->v0x[lane] = t->v0.x; p->v0y[lane] = t->v0.y; p->v0z[lane] = t->v0.z;
            p->e1x[lane] = t->e1.x; p->e1y[lane] = t->e1.y; p->e1z[lane] = t->e1.z;
            p->e2x[lane] = t->e2.x; p->e2y[lane] = t->e2.y; p->e2z[lane] = t->e2.z;
            p->tri[lane] = (uint32_t)idx[j];
        }
    }
    a->tree[node].start_idx = (int32_t)first;
    a->tree[node].count = (int32_t)(a->n_packs - first);
}

/* kd_tree_simd.hpp:146-185 == kd_tree.hpp:40-80 */
static void build_tree(ora_accel *a, int32_t parent, int depth, const int32_t *idx, int64_t n) {
    if (depth == a->max_depth || n <= a->max_leaf) { build_leaf(a, parent, idx, n); return; }
    aabb3 b0, b1;
    aabb_split(&a->tree[parent].box, (unsigned)(depth % 3), &b0, &b1);
    int32_t *c0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *c1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int64_t n0 = 0, n1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        const triangle *t = &a->triangles[idx[i]];
        if (aabb_overlap(&b0, &t->box)) c0[n0++] = idx[i];
        if (aabb_overlap(&b1, &t->box)) c1[n1++] = idx[i];
    }
    if (n0) { const int32_t c = push_node(a, &b0); a->tree[parent].child0 = c; build_tree(a, c, depth + 1, c0, n0); }
    if (n1) { const int32_t c = push_node(a, &b1); a->tree[parent].child1 = c; build_tree(a, c, depth + 1, c1, n1); }
    free(c0); free(c1);
}

ora_accel *ora_accel_build(const ora_scene *s, int kind, float eps, int max_depth, int max_leaf, int W) {
    if (W < 1 || W > ORA_MAXW) return NULL;
    ora_accel *a = (ora_accel *)calloc(1, sizeof(*a));
    a->scene = s; a->kind = kind; a->eps = eps; a->max_depth = max_depth; a->max_leaf = max_leaf; a->W = W;
    int64_t nt = 0;
    for (int m = 0; m < s->n_meshes; ++m) nt += s->meshes[m].ntris;
    a->n_triangles = nt;
    a->triangles = (triangle *)malloc(sizeof(triangle) * (size_t)(nt ? nt : 1));
    int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
    aabb3 root = aabb_empty();
    int64_t k = 0;
    for (int m = 0; m < s->n_meshes; ++m) {                                 /* kd_tree_simd.hpp:101-111 */
        aabb_unite(&root, &s->meshes[m].box);
        for (int i = 0; i < s->meshes[m].ntris; ++i) { a->triangles[k] = s->meshes[m].triangles[i]; all[k] = (int32_t)k; ++k; }
    }
    push_node(a, &root);
    build_tree(a, 0, 0, all, nt);
    free(all);
    return a;
}

void ora_accel_destroy(ora_accel *a) {
    if (!a) return;
    free(a->triangles); free(a->tree); free(a->packs); free(a->leaf_refs); free(a);
}

int64_t ora_accel_num_nodes(const ora_accel *a) { return a->n_nodes; }
int64_t ora_accel_num_packets(const ora_accel *a) { return a->n_packs; }
int64_t ora_accel_num_leaf_refs(const ora_accel *a) { return a->n_refs; }
int64_t ora_accel_num_triangles(const ora_accel *a) { return a->n_triangles; }

void ora_accel_dump(const ora_accel *a, float *nodes_box, int32_t *nodes_link, int32_t *leaf_refs) {
    for (int64_t i = 0; i < a->n_nodes; ++i) {
        const kd_node *n = &a->tree[i];
        float *b = &nodes_box[i * 6];
        b[0] = n->box.min.x; b[1] = n->box.min.y; b[2] = n->box.min.z; b[3] = n->box.max.x; b[4] = n->box.max.y; b[5] = n->box.max.z;
        int32_t *l = &nodes_link[i * 4];
        l[0] = n->child0; l[1] = n->child1; l[2] = (n->start_idx == EMPTY) ? -1 : n->ref_start; l[3] = n->ref_count;
    }
    if (leaf_refs) memcpy(leaf_refs, a->leaf_refs, sizeof(int32_t) * (size_t)a->n_refs);
}

/* ------------------------------------------------------------------ intersection */

typedef struct { float t, u, v; int32_t tri; int found; } candidate;

/* kd_tree_simd.hpp:25-60 (W-wide Möller–Trumbore) + :266-302 (leaf loop, hmin, first-set lane).
 * Written with GCC vector extensions so that every arithmetic line is one W-wide instruction, like the
 * std::experimental::simd code of the reference (W = 16: one zmm register, W = 8: one ymm, W = 4: one xmm). */
#define DEFINE_LEAF_SIMD(W_)                                                                                   \
typedef float vf##W_ __attribute__((vector_size(4 * W_), aligned(4 * W_)));                                    \
typedef int32_t vi##W_ __attribute__((vector_size(4 * W_), aligned(4 * W_)));                                  \
static candidate leaf_simd_##W_(const ora_accel *a, const ray3 *r, const kd_node *leaf, int cull, uint64_t *cn) { \
    candidate best; best.found = 0; best.t = FLT_MAX; best.u = best.v = 0.f; best.tri = -1;                    \
    const float eps = a->eps;                                                                                  \
    const float dx = r->direction.x, dy = r->direction.y, dz = r->direction.z;                                 \
    const float ox = r->origin.x, oy = r->origin.y, oz = r->origin.z;                                          \
    const vi##W_ absmask = (vi##W_){0} + 0x7FFFFFFF;                                                           \
    for (int32_t pi = leaf->start_idx; pi < leaf->start_idx + leaf->count; ++pi) {                             \
        const tri_packet *p = &a->packs[pi];                                                                   \
        const vf##W_ v0x = *(const vf##W_ *)p->v0x, v0y = *(const vf##W_ *)p->v0y, v0z = *(const vf##W_ *)p->v0z; \
        const vf##W_ e1x = *(const vf##W_ *)p->e1x, e1y = *(const vf##W_ *)p->e1y, e1z = *(const vf##W_ *)p->e1z; \
        const vf##W_ e2x = *(const vf##W_ *)p->e2x, e2y = *(const vf##W_ *)p->e2y, e2z = *(const vf##W_ *)p->e2z; \
        const vf##W_ pvx = dy * e2z - dz * e2y;                                                                \
        const vf##W_ pvy = dz * e2x - dx * e2z;                                                                \
        const vf##W_ pvz = dx * e2y - dy * e2x;                                                                \
        const vf##W_ det = e1x * pvx + e1y * pvy + e1z * pvz;                                                  \
        const vf##W_ adet = (vf##W_)((vi##W_)det & absmask);                                                   \
        vi##W_ m = cull ? (eps <= det) : (eps <= adet);                                                        \
        const vf##W_ inv_det = 1.0f / det;                                                                     \
        const vf##W_ tvx = ox - v0x, tvy = oy - v0y, tvz = oz - v0z;                                           \
        const vf##W_ u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv_det;                                        \
        m &= (0.0f <= u) & (u <= 1.0f);                                                                        \
        const vf##W_ qx = tvy * e1z - tvz * e1y;                                                               \
        const vf##W_ qy = tvz * e1x - tvx * e1z;                                                               \
        const vf##W_ qz = tvx * e1y - tvy * e1x;                                                               \
        const vf##W_ v = (dx * qx + dy * qy + dz * qz) * inv_det;                                              \
        m &= (0.0f <= v) & (u + v <= 1.0f);                                                                    \
        vf##W_ t = (e2x * qx + e2y * qy + e2z * qz) * inv_det;                                                 \
        m &= (eps < t);                                                                                        \
        if (cn) cn[ORA_C_PACKETS] += 1;                                                                        \
        int any = 0;                                                                                           \
        for (int l = 0; l < W_; ++l) any |= m[l];                                                              \
        if (!any) continue;                                           /* none_of(mask) :276 */                 \
        const float best_t = best.found ? best.t : FLT_MAX;           /* :280 */                               \
        const vf##W_ bt = (vf##W_){0} + best_t;                                                                \
        t = (vf##W_)(((vi##W_)t & m) | ((vi##W_)bt & ~m));            /* where(!mask,t)=best_t :281 */         \
        float t_min = t[0];                                                                                    \
        for (int l = 1; l < W_; ++l) t_min = (t[l] < t_min) ? t[l] : t_min;   /* hmin :283 */                  \
        if (best_t <= t_min) continue;                                /* :284 */                               \
        int lane = 0;                                                                                          \
        while (lane < W_ && !(t[lane] == t_min)) ++lane;              /* find_first_set :288-290 */            \
        best.found = 1; best.t = t[lane]; best.u = u[lane]; best.v = v[lane]; best.tri = (int32_t)p->tri[lane]; \
    }                                                                                                          \
    return best;                                                                                               \
}

DEFINE_LEAF_SIMD(4)
DEFINE_LEAF_SIMD(8)
DEFINE_LEAF_SIMD(16)

/* generic-width fallback (any W <= ORA_MAXW), same semantics */
static candidate leaf_simd_any(const ora_accel *a, const ray3 *r, const kd_node *leaf, int cull, uint64_t *cn) {
    candidate best; best.found = 0; best.t = FLT_MAX; best.u = best.v = 0.f; best.tri = -1;
    const int W = a->W; const float eps = a->eps;
    const v3 d = r->direction, o = r->origin;
    for (int32_t pi = leaf->start_idx; pi < leaf->start_idx + leaf->count; ++pi) {
        const tri_packet *p = &a->packs[pi];
        float t[ORA_MAXW], u[ORA_MAXW], v[ORA_MAXW]; int mask[ORA_MAXW]; int any = 0;
        for (int l = 0; l < W; ++l) {
            const float pvx = d.y * p->e2z[l] - d.z * p->e2y[l];
            const float pvy = d.z * p->e2x[l] - d.x * p->e2z[l];
            const float pvz = d.x * p->e2y[l] - d.y * p->e2x[l];
            const float det = p->e1x[l] * pvx + p->e1y[l] * pvy + p->e1z[l] * pvz;
            int m = cull ? (eps <= det) : (eps <= fabsf(det));
            const float inv_det = 1.0f / det;
            const float tvx = o.x - p->v0x[l], tvy = o.y - p->v0y[l], tvz = o.z - p->v0z[l];
            const float uu = (tvx * pvx + tvy * pvy + tvz * pvz) * inv_det;
            m &= (0.0f <= uu) & (uu <= 1.0f);
            const float qx = tvy * p->e1z[l] - tvz * p->e1y[l];
            const float qy = tvz * p->e1x[l] - tvx * p->e1z[l];
            const float qz = tvx * p->e1y[l] - tvy * p->e1x[l];
            const float vv = (d.x * qx + d.y * qy + d.z * qz) * inv_det;
            m &= (0.0f <= vv) & (uu + vv <= 1.0f);
            const float tt = (p->e2x[l] * qx + p->e2y[l] * qy + p->e2z[l] * qz) * inv_det;
            m &= (eps < tt);
            t[l] = tt; u[l] = uu; v[l] = vv; mask[l] = m; any |= m;
        }
        if (cn) cn[ORA_C_PACKETS] += 1;
        if (!any) continue;
        const float best_t = best.found ? best.t : FLT_MAX;
        float t_min = 0.f;
        for (int l = 0; l < W; ++l) { if (!mask[l]) t[l] = best_t; t_min = (l == 0) ? t[0] : ((t[l] < t_min) ? t[l] : t_min); }
        if (best_t <= t_min) continue;
        int lane = 0; while (lane < W && !(t[lane] == t_min)) ++lane;
        best.found = 1; best.t = t[lane]; best.u = u[lane]; best.v = v[lane]; best.tri = (int32_t)p->tri[lane];
    }
    return best;
}

/* scene/primitive/triangle.hpp:32-67 — scalar Möller–Trumbore with its own (different) strictness. */
static inline int tri_scalar(const triangle *tr, const ray3 *r, int cull, float eps, float *ot, float *ou, float *ov) {
    const v3 pvec = cross3(r->direction, tr->e2);
    const float det = dot3(tr->e1, pvec);
    if (cull) { if (det <= eps) return 0; } else { if (fabsf(det) <= eps) return 0; }
    const float inv_det = 1.0f / det;
    const v3 tvec = sub3(r->origin, tr->v0);
    const float u = dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || 1.0f < u) return 0;
    const v3 qvec = cross3(tvec, tr->e1);
    const float v = dot3(r->direction, qvec) * inv_det;
    if (v < 0.0f || 1.0f < u + v) return 0;
    const float dist = dot3(tr->e2, qvec) * inv_det;
    if (dist < eps) return 0;
    *ot = dist; *ou = u; *ov = v;
    return 1;
}

typedef struct {                                 /* render/hit.hpp:9-21 (uvs omitted: textures out of scope) */
    ray3 ray; v3 position, hit_normal, face_normal;
    float distance, u, v, w;
    uint32_t mesh_idx, tri_idx;
    const float *uvs;                            /* triangle.uvs (6 floats) */
} hit_rec;

#define ORA_STACK_CAP 256

/* kd_tree_simd.hpp:187-264 and kd_tree.hpp:82-162. */
/* cn: ORA_C_RAYS / ORA_C_HITS are always counted; the per-node / per-triangle tallies only when cn[ORA_C_COUNT] != 0
 * (a detail flag stored one past the counters) so that a timed baseline run does not pay for them. */
static int accel_intersect(const ora_accel *a, const ray3 *ray, int cull, hit_rec *out, uint64_t *cn_all) {
    uint64_t *cn = (cn_all && cn_all[ORA_C_COUNT]) ? cn_all : NULL;
    candidate closest; closest.found = 0; closest.t = FLT_MAX; closest.u = closest.v = 0.f; closest.tri = -1;
    int32_t stack[ORA_STACK_CAP]; int sp = 0;
    stack[sp++] = 0;
    if (cn_all) cn_all[ORA_C_RAYS] += 1;
    while (sp > 0) {
        const kd_node *node = &a->tree[stack[--sp]];
        if (cn) cn[ORA_C_NODES] += 1;
        const float best_t = closest.found ? closest.t : FLT_MAX;
        float box_tmin;
        if (!aabb_ray(&node->box, ray, &box_tmin) || best_t < box_tmin) continue;
        if (cn) cn[ORA_C_BOXPASS] += 1;
        if (node->start_idx == EMPTY) {
            if (node->child0 != EMPTY) stack[sp++] = node->child0;
            if (node->child1 != EMPTY) stack[sp++] = node->child1;
            if (sp > ORA_STACK_CAP - 2) abort();
        } else if (a->kind == ORA_ACCEL_KD_SIMD) {
            if (cn) { cn[ORA_C_LEAVES] += 1; cn[ORA_C_TRIS] += (uint64_t)node->ref_count; }
            candidate c;
            switch (a->W) {
                case 4: c = leaf_simd_4(a, ray, node, cull, cn); break;
                case 8: c = leaf_simd_8(a, ray, node, cull, cn); break;
                case 16: c = leaf_simd_16(a, ray, node, cull, cn); break;
                default: c = leaf_simd_any(a, ray, node, cull, cn); break;
            }
            if (!c.found) continue;
            const float bt = closest.found ? closest.t : FLT_MAX;
            if (c.t < bt) closest = c;                                    /* :222-226 */
        } else {
            if (cn) { cn[ORA_C_LEAVES] += 1; cn[ORA_C_TRIS] += (uint64_t)node->count; }
            for (int32_t k = node->start_idx; k < node->start_idx + node->count; ++k) {   /* kd_tree.hpp:125-157 */
                const int32_t ti = a->leaf_refs[k];
                float t, u, v;
                if (tri_scalar(&a->triangles[ti], ray, cull, a->eps, &t, &u, &v) && (!closest.found || t < closest.t)) {
                    closest.found = 1; closest.t = t; closest.u = u; closest.v = v; closest.tri = ti;
                }
            }
        }
    }
    if (!closest.found) return 0;
    if (cn_all) cn_all[ORA_C_HITS] += 1;
    const triangle *tr = &a->triangles[closest.tri];
    const mesh_object *mesh = &a->scene->meshes[tr->mesh_idx];
    const float u = closest.u, v = closest.v;
    const float w = 1.0f - u - v;
    v3 hn = add3(add3(scl3(u, mesh->vertex_normals[tr->vi[1]]), scl3(v, mesh->vertex_normals[tr->vi[2]])),
                 scl3(w, mesh->vertex_normals[tr->vi[0]]));
    if (a->kind == ORA_ACCEL_KD_SIMD) hn = norm3(hn);                      /* kd_tree_simd.hpp:250 vs kd_tree.hpp:140 */
    out->ray = *ray;
    out->position = add3(ray->origin, scl3(closest.t, ray->direction));
    out->hit_normal = hn;
    out->face_normal = tr->normal;
    out->distance = closest.t; out->u = u; out->v = v; out->w = w;
    out->mesh_idx = tr->mesh_idx; out->tri_idx = (uint32_t)closest.tri;
    out->uvs = tr->uvs;
    return 1;
}

void ora_intersect(const ora_accel *a, const float *rays, size_t n, int cull, ora_hit *out, uint64_t *counters) {
    uint64_t local[ORA_C_COUNT + 1];
    memset(local, 0, sizeof(local));
    local[ORA_C_COUNT] = 1;
    for (size_t i = 0; i < n; ++i) {
        const float *r = &rays[i * 6];
        const ray3 ray = mkray(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]));
        hit_rec h;
        if (accel_intersect(a, &ray, cull, &h, counters ? local : NULL)) {
            out[i].t = h.distance; out[i].u = h.u; out[i].v = h.v; out[i].tri = h.tri_idx; out[i].mesh = h.mesh_idx;
            out[i].normal[0] = h.hit_normal.x; out[i].normal[1] = h.hit_normal.y; out[i].normal[2] = h.hit_normal.z;
        } else {
            out[i].t = -1.0f; out[i].u = 0.f; out[i].v = 0.f; out[i].tri = 0xFFFFFFFFu; out[i].mesh = 0xFFFFFFFFu;
            out[i].normal[0] = out[i].normal[1] = out[i].normal[2] = 0.f;
        }
    }
    if (counters) for (int k = 0; k < ORA_C_COUNT; ++k) counters[k] += local[k];
}

/* ------------------------------------------------------------------ RNG / trig shared with the HIP path */

static inline uint32_t pcg_hash(uint32_t x) {
    const uint32_t s = x * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
/* Replaces utils/rand.hpp:5-19 (thread_local minstd_rand seeded 42 on every thread — a race) with a
 * counter-based generator keyed by the POSITION OF A RAY IN ITS SAMPLE'S RAY TREE, not by the order in which a
 * particular traversal happens to draw: root key = f(seed, absolute pixel, sample); a secondary ray's key is derived
 * from its parent ray's key and its child index; draw j at a ray is a pure function of (key, j).  A recursive CPU
 * evaluation, a per-lane GPU state machine and a level-by-level GPU wavefront therefore all see the same numbers.
 * Uniform in [0,1), 24 bits. */
uint32_t ora_root_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return pcg_hash(sample + pcg_hash(pixel + pcg_hash(seed)));
}
uint32_t ora_child_key(uint32_t key, uint32_t child) { return pcg_hash(key ^ (0x632BE5ABu * (child + 1u))); }
float ora_urand_key(uint32_t key, uint32_t j) {
    const uint32_t h = pcg_hash(key + j * 0x9E3779B9u + 0x85EBCA6Bu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* Deterministic sin/cos in double (Cody–Waite reduction by pi/2 + Taylor), rounded to float once.
 * Stands in for std::sin/std::cos(float) at render.hpp:160-167 so CPU and GPU agree bit for bit. */
void ora_sincos(float angle, float *s, float *c) {
    const double x = (double)angle;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;
    const double kf = floor(x * two_over_pi + 0.5);
    const double r = (x - kf * pio2_hi) - kf * pio2_lo;
    const double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;              /* -1/15! */
    ps = ps * r2 + 1.0 / 6227020800.0;               /*  1/13! */
    ps = ps * r2 - 1.0 / 39916800.0;                 /* -1/11! */
    ps = ps * r2 + 1.0 / 362880.0;                   /*  1/9!  */
    ps = ps * r2 - 1.0 / 5040.0;                     /* -1/7!  */
    ps = ps * r2 + 1.0 / 120.0;                      /*  1/5!  */
    ps = ps * r2 - 1.0 / 6.0;                        /* -1/3!  */
    const double sr = r + r * (r2 * ps);
    double pc = 1.0 / 20922789888000.0;              /*  1/16! */
    pc = pc * r2 - 1.0 / 87178291200.0;              /* -1/14! */
    pc = pc * r2 + 1.0 / 479001600.0;                /*  1/12! */
    pc = pc * r2 - 1.0 / 3628800.0;                  /* -1/10! */
    pc = pc * r2 + 1.0 / 40320.0;                    /*  1/8!  */
    pc = pc * r2 - 1.0 / 720.0;                      /* -1/6!  */
    pc = pc * r2 + 1.0 / 24.0;                       /*  1/4!  */
    pc = pc * r2 - 0.5;                              /* -1/2!  */
    const double cr = 1.0 + r2 * pc;
    const long long k = (long long)kf;
    double sv, cv;
    switch ((int)(k & 3)) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *s = (float)sv; *c = (float)cv;
}

/* ------------------------------------------------------------------ shading (render/render.hpp) */

typedef struct {
    const ora_accel *accel;
    ora_render_params p;
    int width, height;
    float aspect;
    float tan_half_fov;
    float *rgb;
    /* tile scheduling (tile/bucket.hpp:7-21 + tile/queue.hpp:30-41, as an atomic cursor) */
    int tiles_x, tiles_y, bucket;
    volatile int next_tile;
} frame_ctx;

typedef struct {
    uint64_t cn[ORA_C_COUNT + 1];     /* [ORA_C_COUNT] = detail flag, see accel_intersect */
} thread_ctx;

typedef struct { float r, g, b; } col;
static inline col mkcol(float r, float g, float b) { col c = {r, g, b}; return c; }
static inline col cadd(col a, col b) { return mkcol(a.r + b.r, a.g + b.g, a.b + b.b); }      /* color.hpp:9-14 */
static inline col cscl(float s, col a) { return mkcol(s * a.r, s * a.g, s * a.b); }          /* color.hpp:35-41 */
static inline col cdiv(col a, float s) { return mkcol(a.r / s, a.g / s, a.b / s); }          /* color.hpp:16-21 */

/* render.hpp:110-131 */
static int is_occluded(const frame_ctx *f, thread_ctx *tc, ray3 ray, float max_t) {
    const ora_scene *sc = f->accel->scene;
    while (0.0f < max_t) {
        hit_rec h;
        if (!accel_intersect(f->accel, &ray, 0, &h, tc->cn) || max_t < h.distance) return 0;
        const material *m = &sc->materials[sc->meshes[h.mesh_idx].material_idx];
        if (m->kind != ORA_MAT_REFRACTIVE) return 1;                        /* material/queries.hpp:28-30 */
        ray.origin = add3(h.position, scl3(f->p.shadow_bias, ray.direction));
        max_t -= h.distance;
    }
    return 0;
}

/* sample(texture_variant, hit, uvs): scene/texture/albedo.hpp:9-11, edge.hpp:12-21, checker.hpp:13-28.
 * `1. - hit_u - hit_v` has a double literal there: hit_w is computed in double and rounded to F. */
static col sample_texture(const texture *t, const hit_rec *h) {
    const col a = {t->a[0], t->a[1], t->a[2]}, b = {t->b[0], t->b[1], t->b[2]};
    if (t->kind == ORA_TEX_ALBEDO) return a;
    const float hit_u = h->u, hit_v = h->v;
    const float hit_w = (float)(1. - hit_u - hit_v);
    if (t->kind == ORA_TEX_EDGES) return (hit_u < t->param || hit_v < t->param || hit_w < t->param) ? a : b;
    const float fx = (hit_w * h->uvs[0] + hit_u * h->uvs[2]) + hit_v * h->uvs[4];
    const float fy = (hit_w * h->uvs[1] + hit_u * h->uvs[3]) + hit_v * h->uvs[5];
    if (t->kind == ORA_TEX_BITMAP) {                                        /* bitmap.hpp:46-59 */
        /* `(1. - final_uv.y) * height`: double; `final_uv.x * width`: float (size_t converts to F); both truncate into a
         * size_t.  A negative value is undefined there; x86-64 gives 0 above -1 and a huge value below, modelled here. */
        const double rd = (1. - (double)fy) * (double)(size_t)t->bh;
        const float cf = fx * (float)(size_t)t->bw;
        const int64_t ri = (int64_t)rd, ci = (int64_t)cf;
        const size_t row = ri < 0 || (size_t)ri > (size_t)t->bh - 1 ? (size_t)t->bh - 1 : (size_t)ri;
        const size_t column = ci < 0 || (size_t)ci > (size_t)t->bw - 1 ? (size_t)t->bw - 1 : (size_t)ci;
        const uint8_t *px = &t->pixels[(row * (size_t)t->bw + column) * 3];
        const float color_scale = (float)(1.0 / 255.0);                     /* bitmap.hpp:19 */
        return mkcol((float)px[0] * color_scale, (float)px[1] * color_scale, (float)px[2] * color_scale);
    }
    const int32_t u2 = (int32_t)(fx / t->param), v2 = (int32_t)(fy / t->param);
    return ((u2 + v2) % 2 == 0) ? a : b;
}

/* render.hpp:133-308 */
/* `key` is the RNG key of the ray that produced this hit */
static col color_hit(const frame_ctx *f, thread_ctx *tc, const hit_rec *hr, int depth, uint32_t key) {
    const ora_scene *sc = f->accel->scene;
    const col background = mkcol(sc->background[0], sc->background[1], sc->background[2]);
    if (depth == f->p.max_depth) return background;                         /* :138-139 */
    const ray3 in = hr->ray;
    const v3 P = hr->position, hn = hr->hit_normal, fn = hr->face_normal;
    const material *m = &sc->materials[sc->meshes[hr->mesh_idx].material_idx];
    const col albedo = mkcol(m->albedo[0], m->albedo[1], m->albedo[2]);
    const float PI_F = 3.14159265358979323846f;                             /* std::numbers::pi_v<float> */

    switch (m->kind) {
    case ORA_MAT_DIFFUSE: {
        col final = mkcol(0.f, 0.f, 0.f);
        for (int i = 0; i < f->p.diffuse_rays; ++i) {                       /* :151-182 */
            const v3 right = norm3(cross3(in.direction, hn));
            const v3 up = hn;
            const v3 fwd = cross3(right, up);
            const float a_xy = PI_F * ora_urand_key(key, 2u + 2u * (uint32_t)i);
            float s1, c1; ora_sincos(a_xy, &s1, &c1);
            v3 rv = mk(c1, s1, 0.0f);
            const float a_xz = PI_F * ora_urand_key(key, 3u + 2u * (uint32_t)i) * 2.0f;
            float s2, c2; ora_sincos(a_xz, &s2, &c2);
            /* rotate_y_mat * rand_xy_vec (mat3.hpp:53-60), rows {c,0,-s},{0,1,0},{s,0,c} */
            rv = mk(c2 * rv.x + 0.0f * rv.y + (-s2) * rv.z,
                    0.0f * rv.x + 1.0f * rv.y + 0.0f * rv.z,
                    s2 * rv.x + 0.0f * rv.y + c2 * rv.z);
            const v3 org = add3(P, scl3(f->p.reflection_bias, hn));
            /* local_hit_mat(right, up, forward) * rv : rows are the three axes (mat3.hpp:13-17) */
            const v3 dir = mk(right.x * rv.x + right.y * rv.y + right.z * rv.z,
                              up.x * rv.x + up.y * rv.y + up.z * rv.z,
                              fwd.x * rv.x + fwd.y * rv.y + fwd.z * rv.z);
            const ray3 gr = mkray(org, dir);
            hit_rec gh;
            if (!accel_intersect(f->accel, &gr, 0, &gh, tc->cn)) continue;
            final = cadd(final, color_hit(f, tc, &gh, depth + 1, ora_child_key(key, (uint32_t)i)));
        }
        for (int li = 0; li < sc->n_lights; ++li) {                          /* :184-206 */
            const light *L = &sc->lights[li];
            v3 ld = sub3(L->position, P);
            const float radius = len3(ld);
            const float area = 4.0f * PI_F * radius * radius;
            ld = norm3(ld);
            const float cosine = fmaxstd(0.0f, dot3(ld, m->smooth ? hn : fn));
            const ray3 sr = mkray(add3(P, scl3(f->p.shadow_bias, ld)), ld);
            if (is_occluded(f, tc, sr, radius)) continue;
            final = cadd(final, cscl((L->intensity / area) * cosine, albedo));
        }
        final = cdiv(final, (float)(f->p.diffuse_rays + 1));                /* :208 */
        return final;
    }
    case ORA_MAT_TEXTURE: {                                                  /* :211-238: light loop, no GI, no division */
        col final = mkcol(0.f, 0.f, 0.f);
        for (int li = 0; li < sc->n_lights; ++li) {
            const light *L = &sc->lights[li];
            v3 ld = sub3(L->position, P);
            const float radius = len3(ld);
            const float area = 4.0f * PI_F * radius * radius;
            ld = norm3(ld);
            const float cosine = fmaxstd(0.0f, dot3(ld, m->smooth ? hn : fn));
            const ray3 sr = mkray(add3(P, scl3(f->p.shadow_bias, ld)), ld);
            if (is_occluded(f, tc, sr, radius)) continue;
            final = cadd(final, cscl((L->intensity / area) * cosine, sample_texture(&sc->textures[m->texture], hr)));
        }
        return final;
    }
    case ORA_MAT_REFLECTIVE: {                                               /* :239-250 */
        const v3 rd = sub3(in.direction, scl3(2.0f * dot3(in.direction, hn), hn));
        const v3 ro = add3(P, scl3(f->p.reflection_bias, rd));
        const ray3 rr = mkray(ro, rd);
        hit_rec rh;
        if (!accel_intersect(f->accel, &rr, 0, &rh, tc->cn)) return background;
        return color_hit(f, tc, &rh, depth + 1, ora_child_key(key, 0u));
    }
    case ORA_MAT_REFRACTIVE: {                                               /* :252-301 */
        v3 n = norm3(m->smooth ? hn : fn);
        const v3 i = norm3(in.direction);
        float eta_i = 1.0f, eta_r = m->ior;
        if (0.0f < dot3(i, n)) { const float tmp = eta_i; eta_i = eta_r; eta_r = tmp; n = neg3(n); }
        const float cos_i_n = -dot3(i, n);
        const float sin_i_n = sqrtf(1.0f - cos_i_n * cos_i_n);
        if (eta_r / eta_i < sin_i_n) {                                       /* total internal reflection :266-276 */
            const v3 rd = sub3(i, scl3(2.0f * dot3(i, n), n));
            const ray3 rr = mkray(add3(P, scl3(f->p.reflection_bias, rd)), rd);
            hit_rec rh;
            if (!accel_intersect(f->accel, &rr, 0, &rh, tc->cn)) return mkcol(0.f, 0.f, 0.f);
            return color_hit(f, tc, &rh, depth + 1, ora_child_key(key, 0u));
        }
        const float sin_r = ((sin_i_n * eta_i) / eta_r);
        const float cos_r = sqrtf(1.0f - sin_r * sin_r);
        const v3 r = add3(scl3(cos_r, neg3(n)), scl3(sin_r, norm3(add3(i, scl3(cos_i_n, n)))));
        const ray3 fr = mkray(add3(P, scl3(f->p.refraction_bias, r)), r);
        hit_rec fh;
        col refr = mkcol(0.f, 0.f, 0.f);
        if (accel_intersect(f->accel, &fr, 0, &fh, tc->cn)) refr = color_hit(f, tc, &fh, depth + 1, ora_child_key(key, 0u));
        const v3 rd = sub3(i, scl3(2.0f * dot3(i, n), n));
        const ray3 rr = mkray(add3(P, scl3(f->p.reflection_bias, rd)), rd);
        hit_rec rh;
        col refl = mkcol(0.f, 0.f, 0.f);
        if (accel_intersect(f->accel, &rr, 0, &rh, tc->cn)) refl = color_hit(f, tc, &rh, depth + 1, ora_child_key(key, 1u));
        /* :300 — 0.5 * std::pow(float, int) is evaluated in double; x^5 by multiplication here and on the GPU */
        const double x = (double)(1.0f + dot3(i, n));
        const float fresnel = (float)(0.5 * (x * x * x * x * x));
        return cadd(cscl(fresnel, refl), cscl(1.0f - fresnel, refr));
    }
    case ORA_MAT_CONSTANT:
        return albedo;                                                       /* :302-303 */
    default:
        return mkcol(0.f, 0.f, 0.f);
    }
}

/* The camera ray of pixel (x, y), render.hpp:35-62; `key` is the sample's root key (draws 0 and 1 jitter the sample). */
static ray3 camera_ray(const frame_ctx *f, int x, int y, uint32_t key) {
    const ora_scene *sc = f->accel->scene;
    const float *M = sc->cam_mat;
    float rx = (float)x, ry = (float)y;                                    /* :37-38 size_t -> F */
    if (f->p.spp == 1) { rx += 0.5f; ry += 0.5f; }
    else { rx += ora_urand_key(key, 0u); ry += ora_urand_key(key, 1u); }
    const float ndc_x = rx / (float)f->width;                              /* :47-48 F / size_t: the size_t converts to F */
    const float ndc_y = ry / (float)f->height;
    float sx = (2.0f * ndc_x) - 1.0f;
    float sy = 1.0f - (2.0f * ndc_y);
    sx *= f->aspect;
    sx *= f->tan_half_fov;                                                 /* :55-57 float *= tanf(float) */
    sy *= f->tan_half_fov;
    /* transpose(camera.matrix) * (sx, sy, -1)  (mat3.hpp:34-41, :53-60) */
    v3 d = mk(M[0] * sx + M[3] * sy + M[6] * -1.0f,
              M[1] * sx + M[4] * sy + M[7] * -1.0f,
              M[2] * sx + M[5] * sy + M[8] * -1.0f);
    d = norm3(d);
    return mkray(sc->cam_pos, d);
}

/* render.hpp:30-77 */
static void render_tile(frame_ctx *f, thread_ctx *tc, int x0, int y0, int x1, int y1) {
    const ora_scene *sc = f->accel->scene;
    const col background = mkcol(sc->background[0], sc->background[1], sc->background[2]);
    for (int y = y0; y < y1; ++y) {
        for (int x = x0; x < x1; ++x) {
            col final = mkcol(0.f, 0.f, 0.f);
            const uint32_t pixel = (uint32_t)y * (uint32_t)f->width + (uint32_t)x;
            for (int s = 0; s < f->p.spp; ++s) {
                const uint32_t key = ora_root_key(f->p.seed, pixel, (uint32_t)s);
                const ray3 ray = camera_ray(f, x, y, key);
                tc->cn[ORA_C_PRIMARY] += 1;
                hit_rec h;
                if (accel_intersect(f->accel, &ray, 1, &h, tc->cn)) final = cadd(final, color_hit(f, tc, &h, 0, key));
                else final = cadd(final, background);
            }
            final = cdiv(final, (float)f->p.spp);
            float *px = &f->rgb[((size_t)y * (size_t)f->width + (size_t)x) * 3];
            px[0] = final.r; px[1] = final.g; px[2] = final.b;
        }
    }
}

typedef struct { frame_ctx *f; thread_ctx tc; } worker_arg;

static void *worker(void *argp) {
    worker_arg *w = (worker_arg *)argp;
    frame_ctx *f = w->f;
    const int ntiles = f->tiles_x * f->tiles_y;
    for (;;) {
        const int t = __sync_fetch_and_add(&f->next_tile, 1);
        if (t >= ntiles) break;
        const int tx = (t % f->tiles_x) * f->bucket, ty = (t / f->tiles_x) * f->bucket;
        const int x1 = tx + f->bucket < f->width ? tx + f->bucket : f->width;
        const int y1 = ty + f->bucket < f->height ? ty + f->bucket : f->height;
        render_tile(f, &w->tc, tx, ty, x1, y1);
    }
    return NULL;
}

static int frame_setup(frame_ctx *f, const ora_accel *a, const ora_render_params *p) {
    memset(f, 0, sizeof(*f));
    f->accel = a; f->p = *p;
    f->width = p->width > 0 ? p->width : a->scene->width;
    f->height = p->height > 0 ? p->height : a->scene->height;
    if (f->width <= 0 || f->height <= 0 || p->spp < 1) return -1;
    f->aspect = (float)f->width / (float)f->height;                        /* render.hpp:26 */
    /* render.hpp:55-57: degrees_to_radians(fov_degrees) runs in double (utils/convert.hpp:4-6, fov_degrees is a double
     * constant) and is rounded to float by `const F fov_radians`; std::tan(fov_radians / F(2)) is the float overload. */
    const float fov_radians = (float)(p->fov_degrees * (3.14159265358979323846 / 180.0));
    f->tan_half_fov = tanf(fov_radians / 2.0f);
    return 0;
}

/* The camera rays of sample `sample` of every pixel, [h][w] x {origin xyz, direction xyz} (render.hpp:35-62). */
int ora_camera_rays(const ora_accel *a, const ora_render_params *p, int sample, float *rays) {
    frame_ctx f;
    if (frame_setup(&f, a, p) != 0) return -1;
    for (int y = 0; y < f.height; ++y) {
        for (int x = 0; x < f.width; ++x) {
            const uint32_t pixel = (uint32_t)y * (uint32_t)f.width + (uint32_t)x;
            const ray3 r = camera_ray(&f, x, y, ora_root_key(p->seed, pixel, (uint32_t)sample));
            float *o = rays + ((size_t)y * (size_t)f.width + (size_t)x) * 6;
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z; o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        }
    }
    return 0;
}

int ora_render_frame(const ora_accel *a, const ora_render_params *p, float *rgb, uint64_t *counters) {
    frame_ctx f;
    if (frame_setup(&f, a, p) != 0) return -1;
    f.rgb = rgb;
    f.bucket = a->scene->bucket_size > 0 ? a->scene->bucket_size : 64;
    f.tiles_x = (f.width + f.bucket - 1) / f.bucket;
    f.tiles_y = (f.height + f.bucket - 1) / f.bucket;
    f.next_tile = 0;
    int nt = p->n_threads > 0 ? p->n_threads : (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (nt < 1) nt = 1;
    if (nt > 1024) nt = 1024;
    worker_arg *args = (worker_arg *)calloc((size_t)nt, sizeof(worker_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)nt, sizeof(pthread_t));
    for (int i = 0; i < nt; ++i) { args[i].f = &f; args[i].tc.cn[ORA_C_COUNT] = p->count_work ? 1u : 0u; }
    if (nt == 1) worker(&args[0]);
    else {
        for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, worker, &args[i]);
        for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
    }
    if (counters) {
        memset(counters, 0, sizeof(uint64_t) * ORA_C_COUNT);
        for (int i = 0; i < nt; ++i) for (int k = 0; k < ORA_C_COUNT; ++k) counters[k] += args[i].tc.cn[k];
    }
    free(args); free(th);
    return 0;
}

/* io/image/ppm.hpp:7-25 */
size_t ora_write_ppm(const float *rgb, int width, int height, char *buf, size_t cap) {
    size_t n = 0;
    char tmp[64];
#define EMIT(str, len) do { if (buf && n + (len) <= cap) memcpy(buf + n, (str), (len)); n += (len); } while (0)
    int l = snprintf(tmp, sizeof(tmp), "P3\n%d %d\n255\n", width, height);
    EMIT(tmp, (size_t)l);
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float *px = &rgb[((size_t)y * (size_t)width + (size_t)x) * 3];
            unsigned ch[3];
            for (int c = 0; c < 3; ++c) {
                float v = px[c];
                v = (v < 0.0f) ? 0.0f : ((1.0f < v) ? 1.0f : v);           /* std::clamp */
                ch[c] = (unsigned)(uint8_t)(255.999 * (double)v);
            }
            l = snprintf(tmp, sizeof(tmp), "%u %u %u\t", ch[0], ch[1], ch[2]);
            EMIT(tmp, (size_t)l);
        }
        EMIT("\n", (size_t)1);
    }
#undef EMIT
    return n;
}
