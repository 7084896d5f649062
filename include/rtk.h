/*
 * rtk.h — C-ABI of the MI355X-native kd-tree traversal / ray-triangle intersection engine.
 *
 * This is the drop-in boundary for ONE path of MihailMihov/simd-raytracer: everything a caller
 * reaches through the reference's `accelerator` concept and `render_frame` for
 * `kd_tree_simd_accel`.  Plain pointers and sizes only; no C++/torch types; never throws.
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/include/raytracer/).  The reference-side binding a maintainer would add is
 * in INTEGRATION.md; the header-only C++ adapter modelling the concept is
 * simd-raytracer_amd/hip_accel.hpp.
 *
 * There is NO CPU fallback behind this interface: compute entry points return
 * RTK_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef RTK_H
#define RTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTK_ABI_VERSION 4

/* status codes (reference: intersect is noexcept, miss = nullopt, kd_tree_simd.hpp:188,231;
 * loader throws std::invalid_argument, io/json/loader.hpp:104,127,145,170,190,224) */
enum {
    RTK_OK = 0,
    RTK_ERR_INVALID = 1,      /* bad argument / malformed description */
    RTK_ERR_NO_DEVICE = 2,    /* no usable HIP device: the product has no CPU path */
    RTK_ERR_HIP = 3,          /* a HIP runtime call failed */
    RTK_ERR_IO = 4,           /* file could not be read / written */
    RTK_ERR_PARSE = 5,        /* .crtscene is not valid JSON or misses a required key */
    RTK_ERR_UNSUPPORTED = 6   /* feature outside the accelerated path (bitmap texture files other than baseline JPEG) */
};

/* material kinds: scene/material/material.hpp:12 */
enum { RTK_MAT_DIFFUSE = 0, RTK_MAT_REFLECTIVE = 1, RTK_MAT_REFRACTIVE = 2, RTK_MAT_CONSTANT = 3, RTK_MAT_TEXTURE = 4 };

/* texture kinds: scene/texture/texture.hpp:13.  RTK_TEX_BITMAP = bitmap_texture (scene/texture/bitmap.hpp): the texels
 * travel as the RGB bytes `stbi_load` returns (tex_pixels); rtk_scene_load_crtscene decodes baseline JPEG files itself
 * (csrc/jpeg.cpp, stb_image's arithmetic) */
enum { RTK_TEX_ALBEDO = 0, RTK_TEX_EDGES = 1, RTK_TEX_CHECKER = 2, RTK_TEX_BITMAP = 3 };

/* traversal strategy of the device kernels; all of them give bit-identical results */
enum {
    RTK_TRACE_AUTO = 0,   /* batched intersect: wave-cooperative while the wave's rays agree, per-lane otherwise; batches of
                             2^18 rays and more are probed first (every 16th wave of 64 rays; one stream synchronisation):
                             coherent as they come -> RTK_TRACE_WAVE, in no useful order -> sorted first (RTK_TRACE_REPACK);
                             frames: the GROUP4 megakernel (GROUP8 when the frame has few pixel blocks); scenes whose ray trees
                             fork (refraction, diffuse GI) are timed through RTK_TRACE_STREAM and the megakernel on their first
                             frames and keep the faster */
    RTK_TRACE_LANE = 1,   /* one ray per lane, independent stackless traversal */
    RTK_TRACE_WAVE = 2,   /* one wave walks the tree once for its 64 rays (scalar node/triangle fetch) */
    RTK_TRACE_GROUP4 = 3, /* frames only: 4 waves share 64 rays and split every large leaf 4 ways (merge through LDS) */
    RTK_TRACE_GROUP8 = 4, /* frames only: same with 8 waves */
    RTK_TRACE_GROUP16 = 5, /* frames only: same with 16 waves (one 1024-thread workgroup per 8x8 pixel block) */
    RTK_TRACE_STREAM = 6, /* frames only: the ray tree level by level — per-depth path / shadow / combine kernels over
                             compacted ray queues (stream.hip) */
    RTK_TRACE_TWOPASS = 7, /* frames only, spp == 1: camera-ray pass, then the GROUP4 shading pass over the pixel blocks
                             sorted by estimated cost, most expensive first */
    RTK_TRACE_REPACK = 8  /* batched intersect only: the rays are first sorted by the cell of their origin and direction
                             (repack.hip: Morton key over the dimensions that vary, rocPRIM radix sort), then traced in that
                             order -- wave-cooperatively when the sort makes tight waves, with the per-lane fallback otherwise;
                             hits land in the caller's order, bit-identical to every other mode.  Needs 16 B of workspace
                             per ray (kept by the accel; a later batch on another stream waits, on the device, for the batch
                             that is still walking it).  Never blocks the host.  RTK_TRACE_AUTO does this by itself for large
                             incoherent batches -- after a probe whose verdict costs one stream synchronisation, so AUTO on
                             2^18 rays and more is not stream-capturable; RTK_REPACK=0 in the environment turns the probe and
                             the sort off */
};

typedef struct rtk_scene rtk_scene;   /* replaces scene<F>, scene/scene.hpp:14-22 */
typedef struct rtk_accel rtk_accel;   /* replaces kd_tree_simd_accel<F,eps,...>, render/accel/kd_tree_simd.hpp:63-98 */

/* Flattened scene<F>: what parse_scene_file (io/json/loader.hpp:235-265) produces. */
typedef struct {
    int32_t n_meshes;
    const int32_t *mesh_material;   /* [n_meshes]  mesh_object::material_idx */
    const int32_t *mesh_nverts;     /* [n_meshes] */
    const int32_t *mesh_ntris;      /* [n_meshes] */
    const float *vertices;          /* concatenated [sum nverts][3] */
    const uint32_t *indices;        /* concatenated [sum ntris][3], mesh-local vertex indices */
    int32_t n_materials;
    const int32_t *mat_kind;        /* [n_materials] RTK_MAT_* */
    const float *mat_albedo;        /* [n_materials][3] */
    const float *mat_ior;           /* [n_materials] */
    const int32_t *mat_smooth;      /* [n_materials] smooth_shading */
    const int32_t *mat_texture;     /* [n_materials] texture index for RTK_MAT_TEXTURE, ignored otherwise; may be NULL */
    const float *uvs;               /* concatenated per-vertex (u, v) of the meshes with mesh_has_uvs != 0, in mesh order
                                       (loader.hpp:173-192 keeps the first two of every three numbers); may be NULL */
    const int32_t *mesh_has_uvs;    /* [n_meshes] 1 = the mesh has nverts uv pairs in `uvs`, 0 = all-zero uvs; may be NULL */
    int32_t n_textures;             /* scene::textures (scene/scene.hpp:18), referenced by index instead of by name */
    const int32_t *tex_kind;        /* [n_textures] RTK_TEX_* */
    const float *tex_color_a;       /* [n_textures][3] albedo / edge_color / color_A */
    const float *tex_color_b;       /* [n_textures][3] (unused) / inner_color / color_B */
    const float *tex_param;         /* [n_textures] (unused) / edge_width / square_size */
    int32_t n_lights;
    const float *light_pos;         /* [n_lights][3] */
    const float *light_intensity;   /* [n_lights] */
    float cam_pos[3];               /* camera::position, scene/camera.hpp:10 */
    float cam_mat[9];               /* camera::matrix row-major, scene/camera.hpp:11 */
    float background[3];            /* settings::background_color */
    int32_t width, height;          /* settings::image_width/height */
    int32_t bucket_size;            /* settings::bucket_size (default 64, loader.hpp:48) */
    /* bitmap_texture::texture (scene/texture/bitmap.hpp:11-44) of every RTK_TEX_BITMAP texture: 3 bytes per texel, rows top-down,
     * as stbi_load returns them (bitmap.hpp turns a byte b into F(b) * F(1.0 / 255.0)); all textures concatenated.
     * tex_bitmap[i] = {byte offset of texture i in tex_pixels, width, height}.  Both may be NULL without bitmap textures. */
    const uint8_t *tex_pixels;
    const int32_t *tex_bitmap;      /* [n_textures][3] */
} rtk_scene_desc;

typedef struct {
    int32_t n_meshes, n_materials, n_lights;
    int32_t n_vertices, n_triangles;
    int32_t width, height, bucket_size;
    int32_t n_textures, n_uv_vertices;   /* n_uv_vertices = vertices of the meshes that carry uvs */
    int32_t n_bitmap_bytes;              /* size of the concatenated texel bytes of all bitmap textures */
} rtk_scene_info;

/* template parameters of kd_tree_simd_accel (kd_tree_simd.hpp:63-67) as runtime values */
typedef struct {
    int32_t max_depth;              /* 8  */
    int32_t max_leaf_size;          /* 64 */
    float eps;                      /* 1e-6f  (config.hpp:8) */
    int32_t normalize_hit_normal;   /* 1 = kd_tree_simd.hpp:250 behaviour, 0 = kd_tree.hpp:140 behaviour */
    int32_t device;                 /* HIP device ordinal; -1 = current device */
    int32_t traversal;              /* RTK_TRAVERSAL_* ; 0 = the reference's order (the parity mode, default) */
} rtk_accel_params;

/* Leaf visiting order of the wave-cooperative walks (kd_tree_simd.hpp:207-214 pushes child0 then child1 for every ray: child1
 * is always visited first, there is no near/far ordering; README.md:118-124 lists better traversal as the author's to-do).
 *   RTK_TRAVERSAL_REFERENCE  that order: every result bit-identical to the reference (hit-record ties included).
 *   RTK_TRAVERSAL_FAST       front to back: at every split plane the child on the side the rays come from is visited first
 *                            (chosen per wave from the majority sign of the direction on each axis), so a hit found early
 *                            prunes what lies behind it (`best_t < box.t_min`, :203-205).  The closest distance t of every
 *                            ray is unchanged; which of several triangles with EXACTLY that t wins (shared edges and vertices,
 *                            duplicated references) may differ, and with it u, v, the triangle index and the normal of such hits.
 *                            Occlusion queries give the same answers -- except that on scenes with transmissive materials the
 *                            streaming pipeline answers one with a SINGLE any-hit query against the triangles that are not
 *                            transmissive, on the straight segment to the light, instead of is_occluded's loop of closest hits
 *                            that steps through every transmissive surface (render.hpp:110-131; the README's "any-hit shadow
 *                            rays").  The answers differ where an occluder lies within shadow_bias behind a transmissive surface
 *                            (the loop steps over it), beyond the light by less than the biases the loop has accumulated (the loop
 *                            does not take them off max_t), or where a ray grazes an occluder's edge within rounding; measured on
 *                            hw11/scene8 at 1920x1080: 0 pixels.  RTK_FAST_OCCLUDERS=0 keeps the loop.
 *                            Not the parity mode: off unless asked for here, or by RTK_TRAVERSAL_FAST=1 in the environment when
 *                            the accel is built. */
enum { RTK_TRAVERSAL_REFERENCE = 0, RTK_TRAVERSAL_FAST = 1 };

typedef struct {
    int32_t n_nodes, n_inner, n_leaves;
    int32_t n_leaf_refs;            /* unpadded triangle references over all leaves */
    int32_t max_leaf_refs;
    int32_t n_triangles;
    int32_t tree_depth;
    int32_t reserved;
} rtk_tree_info;

/* ray3<F> without the derived inv_direction (core/math/ray3.hpp:5-15); 24 bytes */
typedef struct { float origin[3]; float direction[3]; } rtk_ray;

/* compact hit<F> (render/hit.hpp:9-21); 32 bytes.  Miss: t = -1, tri = mesh = 0xFFFFFFFF.
 * position = origin + t*direction, w = 1-u-v, face_normal/uvs = triangles[tri] — all derivable by the caller. */
typedef struct {
    float t, u, v;
    uint32_t tri;                   /* global triangle index (position in the concatenated mesh order, kd_tree_simd.hpp:103-111) */
    uint32_t mesh;                  /* hit<F>::mesh_idx */
    float normal[3];                /* hit<F>::hit_normal */
} rtk_hit;

/* config.hpp:6-17 as runtime parameters + multi-GPU tile sharding */
typedef struct {
    int32_t width, height;          /* 0 = take from the scene */
    int32_t spp;                    /* samples_per_pixel */
    int32_t max_ray_depth;          /* max_ray_depth */
    int32_t diffuse_rays;           /* diffuse_reflection_ray_count */
    uint32_t seed;                  /* fixed_rng_seed */
    double fov_degrees;             /* fov_degrees */
    float shadow_bias, reflection_bias, refraction_bias;
    int32_t trace_mode;             /* RTK_TRACE_* */
    int32_t rank, world_size;       /* world_size <= 1: whole frame.  Else this rank's share of the scene's buckets
                                     * (render/tile/bucket.hpp:7-21, row-major): bucket i belongs to rank i % world_size -- or,
                                     * when a row of buckets is a whole number of rounds (tiles_x % world_size == 0, which would give
                                     * every rank the same columns), bucket (bx, by) to rank (bx + by) % world_size */
    int32_t collect_stats;          /* 1 = also count nodes/leaves/triangles per ray, as the reference algorithm visits them
                                     * (slower kernel variant); 2 = count what the production path visits (its occlusion
                                     * queries stop at the first answering hit when no material is transmissive; same frame) */
    /* Progressive accumulation (the spp loop of render/render.hpp:34-72 cut into passes): render samples
     * [sample_begin, sample_begin + sample_count) of every pixel.  sample_count == 0: all `spp` samples in one call.
     * A pass with sample_begin > 0 continues the running per-pixel sum the previous pass left in the output buffer (the
     * sum stays in sample order, so N passes give the bits of one call); the pass that ends at `spp` divides by spp
     * (render.hpp:72) and leaves the finished frame.  Until then the buffer holds sums, not colours. */
    int32_t sample_begin, sample_count;
} rtk_render_params;

typedef struct {
    uint64_t rays;                  /* intersect() invocations (primary + shadow segments + reflection + refraction + GI) */
    uint64_t primary;               /* camera rays */
    uint64_t hits;                  /* valid only with collect_stats */
    uint64_t nodes;                 /* tree nodes popped, per ray, summed           (collect_stats) */
    uint64_t boxpass;               /* nodes whose slab test passed                 (collect_stats) */
    uint64_t leaves;                /* leaves entered                               (collect_stats) */
    uint64_t tris;                  /* unpadded triangles tested                    (collect_stats) */
    uint64_t packets16;             /* = sum ceil(leaf_count/16): W=16 packets the reference would test (collect_stats) */
} rtk_counters;

/* ---- library ---- */
int rtk_abi_version(void);
const char *rtk_last_error(void);               /* thread-local message for the last non-OK status */
int rtk_device_count(int *count);               /* number of usable HIP devices (0 is not an error) */

/* ---- scene: replaces parse_scene_file + scene<F> ---- */
int rtk_scene_create(const rtk_scene_desc *desc, rtk_scene **out);      /* copies everything it needs */
int rtk_scene_load_crtscene(const char *path, rtk_scene **out);         /* io/json/loader.hpp:235-265 */
int rtk_scene_get_info(const rtk_scene *scene, rtk_scene_info *info);
/* copies the flattened arrays back out (sizes from rtk_scene_get_info); any pointer may be NULL */
int rtk_scene_get_arrays(const rtk_scene *scene, int32_t *mesh_material, int32_t *mesh_nverts, int32_t *mesh_ntris,
                         float *vertices, uint32_t *indices, int32_t *mat_kind, float *mat_albedo, float *mat_ior,
                         int32_t *mat_smooth, float *light_pos, float *light_intensity, float *cam_pos,
                         float *cam_mat, float *background);
/* texture side of the scene: [n_materials], [n_meshes], [n_uv_vertices][2], [n_textures], [n_textures][3] x2, [n_textures] */
int rtk_scene_get_textures(const rtk_scene *scene, int32_t *mat_texture, int32_t *mesh_has_uvs, float *uvs,
                           int32_t *tex_kind, float *tex_color_a, float *tex_color_b, float *tex_param);
/* bitmap textures: tex_bitmap [n_textures][3] = {byte offset, width, height}, tex_pixels [n_bitmap_bytes]; either may be NULL */
int rtk_scene_get_bitmaps(const rtk_scene *scene, int32_t *tex_bitmap, uint8_t *tex_pixels);
/* The decoder behind bitmap textures: what `stbi_load(path, &w, &h, &channels, 0)` (scene/texture/bitmap.hpp:15) returns for a
 * baseline JPEG held in memory.  *channels = 1 or 3; writes at most cap bytes (pixels may be NULL to query the size). */
int rtk_decode_jpeg(const uint8_t *data, size_t size, int32_t *width, int32_t *height, int32_t *channels,
                    uint8_t *pixels, size_t cap);
/* mesh_object::vertex_normals (scene/object/mesh.hpp:25-43), [nverts of that mesh][3] */
int rtk_scene_vertex_normals(const rtk_scene *scene, int32_t mesh, float *out);
void rtk_scene_destroy(rtk_scene *scene);

/* ---- accel: replaces kd_tree_simd_accel ctor + build_tree (kd_tree_simd.hpp:100-185) ---- */
int rtk_accel_build(const rtk_scene *scene, const rtk_accel_params *params, rtk_accel **out);
int rtk_accel_tree_info(const rtk_accel *accel, rtk_tree_info *info);
/* Tree in the REFERENCE's node order (creation order).  nodes_box [n][6] = min xyz, max xyz;
 * nodes_link [n][4] = child0, child1, leaf_start (index into leaf_refs, -1 for inner), leaf_count;
 * leaf_refs [n_leaf_refs] global triangle indices in leaf order (the unpadded packet contents). */
int rtk_accel_tree_dump(const rtk_accel *accel, float *nodes_box, int32_t *nodes_link, int32_t *leaf_refs);
void rtk_accel_destroy(rtk_accel *accel);

/* ---- batched closest hit: replaces accel.template intersect<cull>(ray) (render/accel/accel.hpp:8-12,
 *      kd_tree_simd.hpp:187-264); one ray per lane ---- */
int rtk_accel_intersect(rtk_accel *accel, const rtk_ray *rays, size_t n, int cull, int trace_mode,
                        rtk_hit *out);                                   /* host buffers, synchronous */
int rtk_accel_intersect_device(rtk_accel *accel, const rtk_ray *d_rays, size_t n, int cull, int trace_mode,
                               rtk_hit *d_out, void *hip_stream);        /* device buffers, stream-ordered */
/* same launch with per-ray work counters accumulated into *counters (device side, synchronous) */
int rtk_accel_intersect_stats(rtk_accel *accel, const rtk_ray *d_rays, size_t n, int cull, int trace_mode,
                              rtk_hit *d_out, rtk_counters *counters);

/* ---- frame: replaces render_frame<A,F> (render/render.hpp:18-108) with color_hit/is_occluded device-side ---- */
/* number of floats the (rank-local) output of rtk_render_frame_device holds */
int rtk_render_output_floats(const rtk_accel *accel, const rtk_render_params *p, size_t *n_floats);
/* host variant: a pass with sample_begin > 0 uploads `rgb` (the running sums of the passes before it) first */
int rtk_render_frame(rtk_accel *accel, const rtk_render_params *p, float *rgb /* host [h][w][3] */,
                     rtk_counters *counters /* may be NULL */);
/* world_size <= 1: d_out is the frame [h][w][3].  world_size > 1: d_out is this rank's compact bucket
 * buffer [buckets_per_rank][bucket][bucket][3] (equal length on every rank, ready for an all-gather). */
int rtk_render_frame_device(rtk_accel *accel, const rtk_render_params *p, float *d_out, void *hip_stream);
/* counters of the most recent rtk_render_frame_device on this accel; synchronises the stream it ran on */
int rtk_render_last_counters(rtk_accel *accel, rtk_counters *counters);
/* The longest 8x8 pixel block of the most recent megakernel frame on this accel, in milliseconds (device real-time clock):
 * the frame's critical path -- no number of compute units or GPUs makes the frame shorter than this.  0 for frames that
 * went through the streaming pipeline or whose blocks all took less than 10 microseconds.  Synchronises the stream the frame ran on. */
int rtk_render_last_critical_path(rtk_accel *accel, double *ms);
/* after the all-gather: d_gathered = [world][buckets_per_rank][bucket][bucket][3] -> d_rgb [h][w][3] */
int rtk_tiles_assemble_device(const rtk_accel *accel, const rtk_render_params *p, const float *d_gathered,
                              float *d_rgb, void *hip_stream);

/* The camera rays render_frame spawns (render/render.hpp:35-62), sample `sample` of every pixel, [h][w] in row-major order:
 * what `ray3<F> ray(camera.position, direction)` at :62 holds.  (Also the generator of the fixed synthetic workload.) */
int rtk_camera_rays(rtk_accel *accel, const rtk_render_params *p, int32_t sample, rtk_ray *rays /* host [h*w] */);
int rtk_camera_rays_device(rtk_accel *accel, const rtk_render_params *p, int32_t sample, rtk_ray *d_rays, void *hip_stream);

/* ---- image out: replaces write_ppm (io/image/ppm.hpp:7-25) ---- */
/* The quantisation of write_ppm on the device: out[i] = uint8(255.999 * clamp(rgb[i], 0, 1)) (ppm.hpp:17-19, the product in
 * double), n = number of floats.  A finished frame leaves the GPU as 3 bytes per pixel instead of 12. */
int rtk_frame_to_rgb8_device(const float *d_rgb, size_t n, uint8_t *d_out, void *hip_stream);
/* the P3 text of write_ppm from such bytes (same output as rtk_format_ppm on the floats) */
int rtk_format_ppm_rgb8(const uint8_t *rgb8, int32_t width, int32_t height, char *buf, size_t cap, size_t *n);
int rtk_write_ppm(const float *rgb, int32_t width, int32_t height, const char *path);
/* returns the byte count in *n; writes at most cap bytes into buf (buf may be NULL to query) */
int rtk_format_ppm(const float *rgb, int32_t width, int32_t height, char *buf, size_t cap, size_t *n);

#ifdef __cplusplus
}
#endif
#endif /* RTK_H */
